"""Load balance of the interleaved row tiles: every rank's share of an N-way split rendered on one GPU."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import multigpu, raytracer as R
r = R.Renderer(0); r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(1.5))
r.renderAsync(1200, 800, 125, 50); r.sync()
for n in (2, 4, 8):
    for tile in (4, 1, 16):
        seg, ms = [], []
        for rank in range(n):
            rows = multigpu.tile_rows(800, n, rank, tile)
            t = []
            for _ in range(3):
                r.renderAsync(1200, 800, 500, 50, rows=rows); r.sync(); st = r.stats(); t.append(st["trace_ms"])
            seg.append(st["ray_segments"]); ms.append(statistics.median(t))
        print(f"{n} ranks, tiles of {tile:2d} rows: segments max/mean {max(seg) / statistics.mean(seg):.4f}, "
              f"kernel ms max/mean {max(ms) / statistics.mean(ms):.4f} (mean {statistics.mean(ms):.2f} ms)", flush=True)
