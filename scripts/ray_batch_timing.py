"""Wall time of the batched per-ray entry points (host arrays in, host arrays out) for n rays."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import raytracer as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1)
with R.Renderer(0) as r:
    r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(1.5))
    s, t = rng.uniform(size=n), rng.uniform(size=n)
    for name, f in (("getRays", lambda: r.getRays(s, t)),):
        f(); t0 = time.perf_counter(); o, d = f(); print(f"{name:12s} n = {n}: {(time.perf_counter() - t0) * 1e3:8.1f} ms")
    for depth in (0, 1, 50):
        r.traceRays(o, d, depth); t0 = time.perf_counter(); c, hid, tt = r.traceRays(o, d, depth)
        print(f"traceRays    n = {n}, depth {depth:2d}: {(time.perf_counter() - t0) * 1e3:8.1f} ms  ({(hid >= 0).mean():.2f} of the rays hit)")
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    args = (-nrm + 0.3 * rng.normal(size=(n, 3)), nrm, np.ones(n, np.uint8), rng.integers(0, 3, n), rng.uniform(size=(n, 3)),
            rng.uniform(size=n), np.full(n, 1.5))
    r.scatterRays(*args); t0 = time.perf_counter(); r.scatterRays(*args)
    print(f"scatterRays  n = {n}: {(time.perf_counter() - t0) * 1e3:8.1f} ms")
