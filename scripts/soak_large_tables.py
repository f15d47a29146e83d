"""Soak of the large-table paths (per-wave LDS rings of -k, scalar-fed centres, 32-bit survivor lists): seeded random
worlds of 2 049 .. 7 000 spheres - table lengths that end anywhere inside a ring tile - with random look-at cameras,
bit for bit against the CPU mirror.  usage: soak_large_tables.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from raytracinginoneweekend_amd import raytracer as R
from raytracinginoneweekend_amd.basetypes import point, vec3
from raytracinginoneweekend_amd.camera import initCamera
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, H = 64, 40
with R.Renderer(0) as r:
    for case in range(cases):
        n = int(rng.integers(2049, 7001))
        spread = float(rng.choice([12.0, 30.0]))
        center = np.column_stack([rng.uniform(-spread, spread, n), rng.uniform(0.0, 1.0, n), rng.uniform(-spread, spread, n)])
        radius = rng.uniform(0.05, 0.4, n) * rng.choice([1.0, 1.0, -1.0], n)
        center[0], radius[0] = (0.0, -1000.0, 0.0), 1000.0
        kind = rng.integers(0, 3, n)
        albedo = rng.uniform(0.0, 1.0, (n, 3)); fuzz = rng.uniform(0.0, 0.6, n); ir = rng.uniform(1.1, 2.0, n)
        r.setWorld({"center_x": center[:, 0], "center_y": center[:, 1], "center_z": center[:, 2], "radius": radius,
                    "kind": kind, "albedo": albedo, "fuzz": fuzz, "ir": ir})
        arr, k = O.spheres_from_arrays(center, radius, kind, albedo, fuzz, ir)
        look_from = (float(rng.normal(0, spread)), float(rng.uniform(0.3, 8.0)), float(rng.normal(0, spread)))
        look_at = (float(rng.normal(0, 3)), 0.3, float(rng.normal(0, 3)))
        vfov, ap, focus = float(rng.uniform(15, 90)), float(rng.choice([0.0, 0.1, 1.0])), float(rng.uniform(2.0, 30.0))
        r.setCamera(initCamera(point(*look_from), point(*look_at), vec3(0, 1, 0), vfov=vfov, aspectRatio=W / H, aperture=ap, focusDist=focus))
        cam = O.book_camera(W / H, look_from=look_from, look_at=look_at, vfov=vfov, aperture=ap, focus=focus)
        spp, depth = int(rng.choice([1, 3, 70])), int(rng.choice([1, 50]))
        got = r.renderRows(W, H, spp, depth, want_accum=True, want_hit_id=True)
        ref = O.render_mirror32(arr, k, cam, W, H, spp, depth, threads=16)
        ok = (got[2] == ref[2]).all() and (got[1] == ref[1]).all() and (got[0] == ref[0]).all() and r.stats()["ray_segments"] == ref[3]
        print(f"case {case}: n {n} ({(n + 7) // 8 * 2} groups), spp {spp}, depth {depth}, camera screen {r.stats()['camera_screen']}: {'equal' if ok else 'DIFFERENT'}", flush=True)
        assert ok
print("soak ok")
