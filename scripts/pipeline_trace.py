"""Twelve 1/8-share frames, two in flight, for a rocprofv3 kernel trace (where do launches overlap, where are gaps?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import multigpu, raytracer as R
scene = R.randomScene()
rows = multigpu.tile_rows(800, 8, 0, 4)
rs = []
for _ in range(2):
    r = R.Renderer(0); r.setWorld(scene); r.setCamera(R.bookCamera(1.5)); rs.append(r)
    r.renderAsync(1200, 800, 64, 50, rows=rows); r.sync()
t0 = time.perf_counter()
for i in range(12):
    r = rs[i % 2]
    r.sync()
    r.renderAsync(1200, 800, 500, 50, rows=rows)
for r in rs: r.sync()
print(f"{(time.perf_counter() - t0) / 12 * 1e3:.3f} ms/frame")
