"""Kernel time against resident workgroups per CU (RTOW_BLOCKS_PER_CU), full frame and a 1/8 share."""
import os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from raytracinginoneweekend_amd import multigpu, raytracer as R
    r = R.Renderer(0); r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(1.5))
    for _ in range(2):
        r.renderAsync(1200, 800, 125, 50); r.sync()
    for tag, spp, rows in (("full spp 250", 250, None), ("1/8 share spp 500", 500, multigpu.tile_rows(800, 8, 0, 4))):
        ts = []
        for _ in range(5):
            r.renderAsync(1200, 800, spp, 50, rows=rows); r.sync(); st = r.stats(); ts.append(st["trace_ms"])
        print(f"blocks/CU {os.environ.get('RTOW_BLOCKS_PER_CU', 'max'):>3s} {tag:18s} trace {statistics.median(ts):8.3f} ms  "
              f"{st['ray_segments'] / statistics.median(ts) / 1e3:8.1f} Mrays/s", flush=True)
else:
    for b in ("2", "3", "4", "5", "6"):
        env = dict(os.environ, RTOW_BLOCKS_PER_CU=b)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True, timeout=120)
