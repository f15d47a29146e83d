"""Kernel time against the chunk sizes of the work counter (RTOW_CHUNK_MAX / RTOW_CHUNK_MIN / RTOW_CHUNK_TAPER), full frame and a 1/8 share."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import multigpu, raytracer as R
wl = sys.argv[1] if len(sys.argv) > 1 else 'C2'
scene = {'C2': lambda: R.randomScene(), 'C3': lambda: R.randomScene(lambertBelow=0.0, metalBelow=0.5), 'C5': lambda: R.randomScene(halfExtent=50)}[wl]()
spp_full, spp_share = (250, 500) if wl != 'C5' else (32, 64)
r = R.Renderer(0); r.setWorld(scene); r.setCamera(R.bookCamera(1.5))
r.renderAsync(1200, 800, 64, 50); r.sync()
share = multigpu.tile_rows(800, 8, 0, 4)
combos = [(256, 256, 4), (512, 256, 4), (512, 256, 16), (512, 256, 32), (512, 256, 64), (512, 512, 4), (384, 256, 16), (512, 192, 32), (1024, 256, 32)]
res = {c: ([], []) for c in combos}
import random
random.seed(1)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 9):
    for cmax, cmin, taper in random.sample(combos, len(combos)):
        os.environ["RTOW_CHUNK_MAX"], os.environ["RTOW_CHUNK_MIN"], os.environ["RTOW_CHUNK_TAPER"] = str(cmax), str(cmin), str(taper)
        r.renderAsync(1200, 800, spp_full, 50); r.sync(); res[(cmax, cmin, taper)][0].append(r.stats()["trace_ms"])
        r.renderAsync(1200, 800, spp_share, 50, rows=share); r.sync(); res[(cmax, cmin, taper)][1].append(r.stats()["trace_ms"])
for c in combos:
    print(f"chunk max {c[0]:5d} min {c[1]:4d} taper {c[2]:3d}: full {statistics.median(res[c][0]):8.3f} ms   1/8 share {statistics.median(res[c][1]):8.3f} ms")
