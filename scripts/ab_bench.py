"""A/B timing of several builds of librtow_hip.so in ONE process, interleaved rounds.
usage: ab_bench.py [--spp N] [--rounds R] [--workload C2|C3] lib_a.so lib_b.so ..."""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import _lib, multigpu, raytracer as R
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=125)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--workload", default="C2")
ap.add_argument("--share", type=int, default=1, help="render only rank 0's row tiles of an N-way split")
ap.add_argument("--half-extent", type=int, default=0, help="randomScene(halfExtent=...) instead of a named workload (11 = 484 spheres, 50 = 10 001)")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
scene = {"C2": lambda: R.randomScene(), "C3": lambda: R.randomScene(lambertBelow=0.0, metalBelow=0.5),
         "C5": lambda: R.randomScene(halfExtent=50)}[a.workload]()
if a.half_extent:
    scene = R.randomScene(halfExtent=a.half_extent)
n = len(scene)
scene = scene.flatten()
scene.pop("hittable_kind")   # builds from before rtow_set_hittables take the plain sphere arrays
rows = None if a.share == 1 else multigpu.tile_rows(800, a.share, 0, 4)
rs = []
for path in a.libs:
    L = _lib.load(os.path.abspath(path), strict=False)
    r = R.Renderer(0, lib=L)
    if hasattr(L, "rtow_set_scene_f64"):
        r.setWorld(scene); r.setCamera(R.bookCamera(1.5))
    else:   # a build from before the binary64 scene / camera entry points
        r.setWorld(scene, binary32=True); r.setCamera(R.bookCamera(1.5).to_struct(binary32=True))
    r.renderAsync(1200, 800, 8, 50); r.sync()
    rs.append(r)
times = [[] for _ in rs]; seg = [0] * len(rs)
for _ in range(a.rounds):
    for k, r in enumerate(rs):
        r.renderAsync(1200, 800, a.spp, 50, rows=rows); r.sync(); st = r.stats()
        times[k].append(st["trace_ms"]); seg[k] = st["ray_segments"]
for k, path in enumerate(a.libs):
    med, mn = statistics.median(times[k]), min(times[k])
    flop = seg[k] * (18 * n + 65)
    print(f"{os.path.basename(path):28s} median {med:8.3f} ms  min {mn:8.3f} ms  {seg[k]/med/1e3:8.1f} Mrays/s  "
          f"{flop/med/1e9/157.3*100:5.2f}% roofline  segs {seg[k]}")
