"""Soak of the camera rays' first tests (mask test + interval test, the default) against the same render without any
first test: seeded random worlds and look-at cameras with extreme parameters - fields of view from 1 to 175 degrees,
lenses from closed to wider than the scene, focus distances from a thousandth of to a hundred times the distance to
the target, cameras inside, far outside and on the ground, rolled, frames from 16 x 16 to long strips, spheres from
1e-3 to 50 in radius of either sign, worlds from one sphere to a few thousand.  Hit ids, 64-bit sums, pixels and
segment counts must be equal.  usage: soak_first_tests.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import _lib, raytracer as R
from raytracinginoneweekend_amd.basetypes import point, vec3
from raytracinginoneweekend_amd.camera import initCamera

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
modes = {_lib.SCREEN_MASK: 0, _lib.SCREEN_INTERVAL: 0, _lib.SCREEN_SLAB: 0}
bad = 0
pixels = hits = 0
with R.Renderer(0) as r:
    for case in range(cases):
        if case % 5 == 0:
            kind = int(rng.integers(0, 4))
            if kind == 0:
                r.setWorld(R.randomScene(halfExtent=int(rng.integers(1, 16))))
                extent = 12.0
            else:
                n = int(rng.choice([1, 3, 40, 400, 3000]))
                extent = float(rng.choice([0.5, 5.0, 80.0]))
                rad = np.exp(rng.uniform(np.log(1e-3), np.log(0.6), n)) * extent * rng.choice([1.0, -1.0], n)
                r.setWorld({"center_x": rng.uniform(-extent, extent, n), "center_y": rng.uniform(-0.3 * extent, 0.6 * extent, n),
                            "center_z": rng.uniform(-extent, extent, n), "radius": rad, "kind": rng.integers(0, 3, n),
                            "albedo": rng.uniform(0, 1, (n, 3)), "fuzz": rng.uniform(0, 0.5, n), "ir": np.full(n, 1.5)})
        look_from = rng.normal(0, 1, 3) * extent * float(rng.choice([0.05, 0.5, 1.5, 20.0]))
        if rng.random() < 0.3:
            look_from[1] = 1e-3 * extent
        look_at = rng.normal(0, 1, 3) * extent * float(rng.choice([0.0, 0.3, 1.0]))
        dist = float(np.linalg.norm(look_from - look_at))
        if dist < 1e-6 * extent:
            continue
        vup = (0.0, 1.0, 0.0) if rng.random() < 0.5 else tuple(rng.normal(0, 1, 3))
        if np.linalg.norm(np.cross(np.asarray(vup), look_from - look_at)) < 1e-3 * dist:
            continue
        W, H = [(16, 16), (64, 40), (200, 16), (16, 120), (97, 61)][int(rng.integers(0, 5))]
        cam = initCamera(point(*look_from), point(*look_at), vec3(*vup), vfov=float(rng.choice([1.0, 15.0, 60.0, 120.0, 175.0])),
                         aspectRatio=W / H, aperture=float(rng.choice([0.0, 0.01, 1.0, 30.0])) * extent / 10.0,
                         focusDist=dist * float(rng.choice([1e-3, 0.3, 1.0, 3.0, 100.0])))
        r.setCamera(cam)
        spp = int(rng.choice([1, 5, 64, 200]))
        got = r.renderRows(W, H, spp, 12, want_accum=True, want_hit_id=True)
        st = r.stats()
        modes[st["camera_screen"]] = modes.get(st["camera_screen"], 0) + 1
        os.environ["RTOW_NO_SLAB_TEST"] = "1"
        plain = r.renderRows(W, H, spp, 12, want_accum=True, want_hit_id=True)
        del os.environ["RTOW_NO_SLAB_TEST"]
        same = (got[2] == plain[2]).all() and (got[1] == plain[1]).all() and (got[0] == plain[0]).all() and st["ray_segments"] == r.stats()["ray_segments"]
        pixels += got[2].size; hits += int((got[2] >= 0).sum())
        if not same:
            bad += 1
            print(f"case {case}: DIFFERENT (screen {st['camera_screen']}, {W}x{H}, spp {spp}, hit ids differing {(got[2] != plain[2]).sum()})", flush=True)
        if case % 100 == 99:
            print(f"{case + 1} cases, {bad} different", flush=True)
print(f"cases by first test (rtow_stats.camera_screen): {modes}; {hits} of {pixels} pixels have a sphere behind their first sample")
print("soak ok" if bad == 0 else f"soak FAILED: {bad} cases differ")
sys.exit(0 if bad == 0 else 1)
