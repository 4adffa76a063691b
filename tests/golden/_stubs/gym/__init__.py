"""Test-only stand-in for `gym` so the read-only reference imports (PokerEnv.py:8 uses gym.spaces only)."""
from . import spaces  # noqa: F401
