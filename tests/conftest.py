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


THUMB_BOXES = (100, 150)   # 8 x 8 pixel boxes of the 1200 x 800 frame


def load_png_thumbnail():
    """Box means (8 x 8 pixels, in 8-bit levels) of the reference's own render media/rayTracing.png and the
    masks of boxes that lie on one object each; tests/golden/make_golden.py derives them (the image itself
    is not copied)."""
    import numpy as np
    g = np.load(os.path.join(GOLDEN, "reference_png_thumb.npz"))
    thumb = g["thumb_x64"].astype(np.float64) / 64.0
    masks = {k[5:]: np.unpackbits(g[k])[: THUMB_BOXES[0] * THUMB_BOXES[1]].reshape(THUMB_BOXES).astype(bool)
             for k in g.files if k.startswith("mask_")}
    return thumb, masks


# mean |difference| in 8-bit levels a render may show against the thumbnail, per object, at the two noise levels
# the tests use (a wrong Schlick term, a dropped fuzz or an inverted refraction ratio exceed them several times)
PNG_LIMITS_SPP16 = {"all": 0.9, "glass": 2.0, "smallglass": 2.2, "smallmetal": 1.0, "bigmetal": 0.6, "brown": 0.9,
                    "smalllambert": 0.9, "ground": 1.1, "sky": 0.05}
# the same at spp 128 over the whole frame (about twice what the GPU's render shows)
PNG_LIMITS_SPP128 = {"all": 0.45, "glass": 0.8, "smallglass": 0.95, "smallmetal": 0.55, "bigmetal": 0.3, "brown": 0.45,
                     "smalllambert": 0.45, "ground": 0.6, "sky": 0.05}


def png_box_differences(rgb, thumb, masks, bands=None):
    """rgb: uint8 [rows, 1200, 3] covering whole 8-row bands (all 100, or those listed in `bands`).
    Returns {object: mean |box mean - thumbnail|}, plus "all" and the signed mean difference "bias"."""
    import numpy as np
    nb = rgb.shape[0] // 8
    m = rgb.astype(np.float64).reshape(nb, 8, 150, 8, 3).mean((1, 3))
    sel = slice(None) if bands is None else list(bands)
    d = m - thumb[sel]
    out = {"all": float(np.abs(d).mean()), "bias": float(np.abs(d.mean((0, 1))).max())}
    for k, v in masks.items():
        mk = v[sel]
        out[k] = float(np.abs(d[mk]).mean()) if mk.any() else 0.0
    return out
