import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def book_scene(oracle):
    """The reference's 484-sphere randomScene as oracle structs."""
    arr, n, draws, end = oracle.random_scene()
    return arr, n


@pytest.fixture(scope="session")
def gpu_renderer():
    """A Renderer on device 0 with the book scene loaded; fails (not skips) without the library."""
    from raytracinginoneweekend_amd import raytracer as R
    r = R.Renderer(0)
    r.setWorld(R.randomScene())
    yield r
    r.close()
