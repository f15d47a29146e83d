"""Instrumented build (librtow_hip_prof.so): wave-level statistics of the sphere scan."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTOW_LIB_PATH"] = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "librtow_hip_prof.so")
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import raytracer as R, _lib
L = _lib.lib()
wl = sys.argv[2] if len(sys.argv) > 2 else 'C2'
scene = {'C2': lambda: R.randomScene(), 'C3': lambda: R.randomScene(lambertBelow=0.0, metalBelow=0.5), 'C5': lambda: R.randomScene(halfExtent=50)}[wl]()
r = R.Renderer(0); r.setWorld(scene); r.setCamera(R.bookCamera(1.5))
buf = (C.c_ulonglong * 16)()
L.rtow_debug_counters(buf, 1)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r.renderAsync(1200, 800, spp, 50); r.sync(); st = r.stats()
L.rtow_debug_counters(buf, 0)
g, cw, cl, wi, li = buf[0], buf[1], buf[2], buf[3], buf[4]
print(f"segments {st['ray_segments']}  wave iterations {wi}  lane iterations {li}  lane utilisation {li/(64*wi):.4f}")
print(f"per wave iteration: group slow-path entries {g/wi:.2f} of {(len(scene)+7)//8*2}, candidate spheres (wave level) {cw/wi:.2f}, candidate lanes {cl/wi:.2f} (per lane {cl/li:.3f})")
tot = buf[5] + buf[6] + buf[7] + buf[8]
print(f"wave-cycle split: refill {buf[5]/tot:.3f}  normalise+big {buf[6]/tot:.3f}  scan {buf[7]/tot:.3f}  shade {buf[8]/tot:.3f}  (cycles per wave iteration {tot/wi:.0f})")
print(f"per wave iteration: groups with a survivor in any lane {buf[9]/wi:.2f}, drain rounds {buf[10]/wi:.2f}, "
      f"unit-ball loop trips (slowest lane) {buf[11]/wi:.2f}, pool batches {buf[14]/wi:.3f}")
