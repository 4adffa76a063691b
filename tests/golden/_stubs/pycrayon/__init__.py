"""Test-only stand-in for `pycrayon` (imported unconditionally by the reference's CrayonWrapper, never used here)."""


class CrayonClient:
    def __init__(self, *a, **k):
        raise RuntimeError("pycrayon stub: no crayon server in the golden-vector harness")
