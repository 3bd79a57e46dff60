import os
import sys

import pytest

try:  # torch ships its own HIP runtime: load it before libmrt.so pulls in the system one, whatever tests are selected
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One mrt context on GPU 0 for the whole session (GPU tests only)."""
    from raytracing_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
