import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once, here, where
    hipcc and gcc exist.  On a GPU box the prebuilt files travel with the tree and nothing happens."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "librtow_hip.so")
    exe = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "rtow_raytracer")
    if (not os.path.exists(lib) or not os.path.exists(exe)) and shutil.which("hipcc"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracinginoneweekend_amd", "csrc"), "-s"])


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
