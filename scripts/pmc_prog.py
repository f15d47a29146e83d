"""Program run under rocprofv3 --pmc: one warm-up render, then one measured render of C2 at --spp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import _lib, raytracer as R
lib = sys.argv[1]; spp = int(sys.argv[2]); workload = sys.argv[3] if len(sys.argv) > 3 else "C2"
depth = int(os.environ.get("PMC_DEPTH", "50"))
scene = R.randomScene() if workload == "C2" else R.initHittables(0) if workload == "EMPTY" else R.randomScene(lambertBelow=0.0, metalBelow=0.5)
r = R.Renderer(0, lib=_lib.load(os.path.abspath(lib), strict=False))
scene = scene.flatten(); scene.pop("hittable_kind")   # builds from before rtow_set_hittables
r.setWorld(scene); r.setCamera(R.bookCamera(1.5))
r.renderAsync(1200, 800, 2, depth); r.sync()
r.renderAsync(1200, 800, spp, depth); r.sync()
st = r.stats()
print(f"segments {st['ray_segments']} trace_ms {st['trace_ms']:.3f}")
