"""Segments per path of the book scene at growing sample counts: stays at 2.7132 across 2^31 and 2^32 paths
(it dropped once the path index passed 2^31 before the chunk index was widened correctly)."""
import os, sys
sys.path.insert(0, os.getcwd())
from raytracinginoneweekend_amd import raytracer as R
r = R.Renderer(0); r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(1.5))
for spp in (500, 2000, 2300, 2500, 4400, 4600, 5000):
    r.renderAsync(1200, 800, spp, 50); r.sync(); st = r.stats()
    print(spp, 1200*800*spp, st["ray_segments"], st["ray_segments"] / (1200*800*spp), st["trace_ms"], flush=True)
