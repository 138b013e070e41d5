import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import cpu_oracle
    cpu_oracle.build()


@pytest.fixture(autouse=True)
def _reset_binding_state():
    """Process-wide switches of the binding a test may have flipped go back to their defaults after every test: the opt-in
    geometry reuse between two rasterizer calls and the allocation-poison test hook."""
    yield
    mod = sys.modules.get("diff_gaussian_rasterization._C")
    if mod is not None:
        mod.set_geometry_cache(None)
        mod.set_alloc_poison(None)
