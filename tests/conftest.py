import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The CPU suite needs the shared library for its host-side entry points (tree builder, LUTs, env, scalar
    evaluator) and to check the exported symbols. hipcc cross-compiles without a GPU."""
    from pokerrl_amd import _native
    if not os.path.isfile(_native.LIB_PATH):
        from pokerrl_amd.build import build_native
        build_native()
    yield
