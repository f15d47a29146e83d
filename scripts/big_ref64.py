"""The full 1200x800 frame against the FP64 restatement of the reference (same random streams)."""
import math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from raytracinginoneweekend_amd import raytracer as R
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 64
with R.Renderer(0) as r:
    r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(W / H))
    rgb, _, hid = r.renderRows(W, H, spp, 50, want_hit_id=True)
    seg = r.stats()["ray_segments"]
arr, n, _, _ = O.random_scene()
t0 = time.time()
ref = O.render_ref64(arr, n, O.book_camera(W / H), W, H, spp, 50, threads=16)
d = rgb.astype(np.float64) - ref[0].astype(np.float64)
mse = np.mean(d * d)
print(f"spp {spp}: ref64 {time.time() - t0:.1f} s; PSNR {10 * math.log10(255.0 ** 2 / mse):.2f} dB, mean |diff| {np.mean(np.abs(d)):.4f} levels, "
      f"max |diff| {np.max(np.abs(d)):.0f}, pixels differing {np.mean(np.any(d != 0, axis=2)):.5f}, primary hit ids differing "
      f"{int(np.sum(hid != ref[2]))} of {W * H}, segments {seg} vs {ref[3]} ({(seg - ref[3]) / ref[3]:+.2e})", flush=True)
