"""Frames in flight: wall time per frame with one context (serial) and with two contexts alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import multigpu, raytracer as R
scene = R.randomScene()
rs = []
for _ in range(3):
    r = R.Renderer(0); r.setWorld(scene); r.setCamera(R.bookCamera(1.5)); rs.append(r)
    r.renderAsync(1200, 800, 64, 50); r.sync()
def run(depth, rows, spp, frames=12):
    for r in rs: r.sync()
    t0 = time.perf_counter()
    for i in range(frames):
        r = rs[i % depth]
        r.sync()
        r.renderAsync(1200, 800, spp, 50, rows=rows)
    for r in rs: r.sync()
    return (time.perf_counter() - t0) / frames * 1e3
for tag, rows, spp in (("full spp 500", None, 500), ("1/8 share spp 500", multigpu.tile_rows(800, 8, 0, 4), 500),
                       ("full spp 62", None, 62)):
    for depth in (1, 2, 3, 1, 2):
        print(f"{tag:18s} depth {depth}: {run(depth, rows, spp):8.3f} ms/frame", flush=True)
