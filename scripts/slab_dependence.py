"""What the cheap per-ray tests in front of the camera rays' screen are worth, and what that depends on.

The default is the mask test (the image tile the ray aims at against the tiles in which a group's spheres can be met)
followed by the interval test (the ray's image coordinates against the ranges in which a sphere can be met at all;
alone with RTOW_CAMERA_SCREEN=interval); RTOW_CAMERA_SCREEN=slab selects the older slab test (distance to the ray's
line seen along a world axis).  Each lets a wave skip the 7-FMA screen for a group of four spheres when none of its 64
camera rays passes for any of them.  Those 64 rays are consecutive (pixel, sample) pairs of a chunk: 64 samples of ONE pixel when
spp >= 64, several neighbouring pixels of a row when spp is smaller, and the lens spreads their origins over the
aperture.  This script renders the same frame with the default tests, with the interval test alone, with the slab test and with
RTOW_NO_SLAB_TEST=1 (read at every launch)
over sample counts, apertures, a camera inside the sphere field, the all metal / glass scene and the
10 001-sphere scene, and reports trace-kernel times (HIP events, median of 3) and - from the instrumented
build, when it is present - how many groups of a PRIMARY scan go on to the screen.
usage: slab_dependence.py [--quick]"""
import ctypes as C, math, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import _lib, raytracer as R
from raytracinginoneweekend_amd.basetypes import point, vec3
from raytracinginoneweekend_amd.camera import initCamera

quick = "--quick" in sys.argv
W, H, DEPTH = 1200, 800, 50
prof_path = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "librtow_hip_prof.so")
prof = _lib.load(prof_path) if os.path.exists(prof_path) else None


def book(aperture):
    return initCamera(point(13, 2, 3), point(0, 0, 0), vec3(0, 1, 0), 20, W / H, aperture=aperture, focusDist=10.0)


inside = initCamera(point(1, 0.5, 1), point(-3, 0.4, -2), vec3(0, 1, 0), 40, W / H, aperture=0.1, focusDist=3.0)
down = initCamera(point(0, 30, 0.01), point(0, 0, 0), vec3(0, 1, 0), 40, W / H, aperture=0.1, focusDist=30.0)
scenes = {"C2 book scene (484)": R.randomScene(), "C3 metal/glass (484)": R.randomScene(lambertBelow=0.0, metalBelow=0.5),
          "10 001 spheres": R.randomScene(halfExtent=50)}
cases = []
for spp in ((1, 8, 32, 64, 500) if not quick else (1, 64)):
    cases.append(("C2 book scene (484)", "book camera, aperture 0.1", book(0.1), spp))
for ap in (0.0, 2.0):
    for spp in ((8, 64, 500) if not quick else (64,)):
        cases.append(("C2 book scene (484)", f"book camera, aperture {ap}", book(ap), spp))
for spp in ((8, 64, 500) if not quick else (64,)):
    cases.append(("C2 book scene (484)", "camera inside the field at (1, 0.5, 1)", inside, spp))
    cases.append(("C2 book scene (484)", "camera 30 units above, looking down", down, spp))
for spp in ((8, 500) if not quick else (64,)):
    cases.append(("C3 metal/glass (484)", "book camera, aperture 0.1", book(0.1), spp))
for spp in ((1, 8, 64) if not quick else (8,)):
    cases.append(("10 001 spheres", "book camera, aperture 0.1", book(0.1), spp))
    cases.append(("10 001 spheres", "camera inside the field at (1, 0.5, 1)", inside, spp))


def timed(r, spp, mode):
    if mode == "none":
        os.environ["RTOW_NO_SLAB_TEST"] = "1"
    elif mode in ("slab", "interval"):
        os.environ["RTOW_CAMERA_SCREEN"] = mode
    try:
        ms = []
        r.renderAsync(W, H, spp, DEPTH); r.sync()     # one untimed launch of this kernel instance first
        for _ in range(5):
            r.renderAsync(W, H, spp, DEPTH); r.sync()
            ms.append(r.stats()["trace_ms"])
        return statistics.median(ms), r.stats()["ray_segments"]
    finally:
        os.environ.pop("RTOW_NO_SLAB_TEST", None)
        os.environ.pop("RTOW_CAMERA_SCREEN", None)


# timing first, all of it with the product build; the instrumented build runs afterwards, so that its slow launches
# sit between no two timed ones
rows = []
current = None
r = None
for scene_name, cam_name, cam, spp in cases:
    if current != scene_name:
        if r is not None:
            r.close()
        r = R.Renderer(0); r.setWorld(scenes[scene_name])
        current = scene_name
    r.setCamera(cam)
    t_with, seg = timed(r, spp, "default")
    t_slab, seg3 = timed(r, spp, "slab")
    t_interval, seg4 = timed(r, spp, "interval")
    t_without, seg2 = timed(r, spp, "none")
    assert seg == seg2 == seg3 == seg4
    rows.append([scene_name, cam_name, cam, spp, t_with, t_without, seg, "", t_slab, t_interval])
    print(f"timed {scene_name} / {cam_name} / spp {spp}", file=sys.stderr, flush=True)
r.close()
if prof is not None:
    current, rp = None, None
    for row in rows:
        scene_name, cam_name, cam, spp = row[:4]
        n = len(scenes[scene_name])
        if n > 1000 and spp > 8:
            continue
        if current != scene_name:
            if rp is not None:
                rp.close()
            rp = R.Renderer(0, lib=prof); rp.setWorld(scenes[scene_name])
            current = scene_name
        b = (C.c_ulonglong * 32)()
        rp.setCamera(cam)
        prof.rtow_debug_counters(b, 1)
        rp.renderAsync(W, H, min(spp, 64), DEPTH); rp.sync()
        prof.rtow_debug_counters(b, 0)
        groups = (n + 7) // 8 * 2
        row[7] = f"{b[26] / max(1, b[14]):6.1f} / {b[24] / max(1, b[14]):5.1f} of {groups} = {b[24] / max(1, b[14]) / groups:.3f}"
print(f"{'scene':22s} {'camera':40s} {'spp':>4s} {'default ms':>11s} {'interval ms':>11s} {'slab ms':>9s} {'without ms':>10s} {'gain':>7s} {'intv gain':>9s} {'slab gain':>9s} {'Mrays/s':>9s} {'frac':>6s} {'no-test frac':>12s} "
      f"{'groups past the masks / on to the screen':>40s}")
for scene_name, cam_name, cam, spp, t_with, t_without, seg, passing, t_slab, t_interval in rows:
    n = len(scenes[scene_name])
    flop = seg * (18 * n + 65)
    print(f"{scene_name:22s} {cam_name:40s} {spp:4d} {t_with:11.3f} {t_interval:11.3f} {t_slab:9.3f} {t_without:10.3f} {100 * (t_without / t_with - 1):6.1f}% {100 * (t_without / t_interval - 1):8.1f}% {100 * (t_without / t_slab - 1):8.1f}% "
          f"{seg / t_with / 1e3:9.1f} {flop / t_with / 1e9 / 157.3:6.3f} {flop / t_without / 1e9 / 157.3:12.3f} {passing:>40s}", flush=True)
