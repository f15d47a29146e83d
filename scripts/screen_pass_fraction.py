"""Share of a PRIMARY scan's groups of four spheres that go on from the camera rays' first tests (mask test + interval
test, interval test alone, slab test) to the 7-FMA screen, per benchmark workload: instrumented build (librtow_hip_prof.so), book camera, 1200x800 frame
(the 10 001-sphere scene at 8 samples per pixel, the others at 64: one pixel's samples per wavefront, as at the
benchmark's sample counts).  Prints the JSON bench.py reads from profiles/r*_screen_pass_fraction.json."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import _lib, raytracer as R
prof = _lib.load(os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "librtow_hip_prof.so"))
scenes = {"C2": (R.randomScene(), 64), "C3": (R.randomScene(lambertBelow=0.0, metalBelow=0.5), 64), "C5": (R.randomScene(halfExtent=50), 8)}
out = {"source": "scripts/screen_pass_fraction.py (instrumented build): groups of a PRIMARY scan that reach the filter / groups of the "
                 "table; book camera, aperture 0.1, 1200x800; C5 = the 10 001-sphere scene at spp 8 (the full configuration renders "
                 "3840x2160: smaller pixels, so if anything fewer groups pass)", "mask": {}, "interval": {}, "slab": {}}
for name, (scene, spp) in scenes.items():
    r = R.Renderer(0, lib=prof)
    r.setWorld(scene)
    r.setCamera(R.bookCamera(1.5))
    groups = (len(scene) + 7) // 8 * 2
    for screen in ("mask", "interval", "slab"):
        if screen != "mask":
            os.environ["RTOW_CAMERA_SCREEN"] = screen
        b = (C.c_ulonglong * 32)()
        prof.rtow_debug_counters(b, 1)
        r.renderAsync(1200, 800, spp, 50); r.sync()
        prof.rtow_debug_counters(b, 0)
        os.environ.pop("RTOW_CAMERA_SCREEN", None)
        out[screen][name] = round(b[24] / max(1, b[14]) / groups, 4)
        if screen == "interval":
            out.setdefault("interval_first_axis_only", {})[name] = round(b[25] / max(1, b[14]) / groups, 4)
        if screen == "mask":   # groups whose word has the bit of some ray of the wave: these see the interval test
            out.setdefault("mask_word_pass", {})[name] = round(b[26] / max(1, b[14]) / groups, 4)
    r.close()
print(json.dumps(out, indent=1))
