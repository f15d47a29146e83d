"""Kernel time against work size: full frame at several spp, and 1/N row shares at spp 500.
A straight-line fit separates the per-launch fixed cost from the per-path cost."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import multigpu, raytracer as R
r = R.Renderer(0)
r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(1.5))
for _ in range(3):
    r.renderAsync(1200, 800, 125, 50); r.sync()
def run(spp, rows, tag):
    ts = []
    for _ in range(5):
        r.renderAsync(1200, 800, spp, 50, rows=rows); r.sync(); st = r.stats(); ts.append(st["trace_ms"])
    print(f"{tag:24s} trace {statistics.median(ts):8.3f} ms (min {min(ts):8.3f})  segs {st['ray_segments']:11d}  "
          f"{st['ray_segments'] / statistics.median(ts) / 1e3:8.1f} Mrays/s", flush=True)
for spp in (16, 31, 62, 125, 250, 500):
    run(spp, None, f"full spp {spp}")
for n in (2, 4, 8, 16):
    run(500, multigpu.tile_rows(800, n, 0, 4), f"1/{n} share spp 500")
run(500, list(range(400, 500)), "rows 400..499 spp 500")
