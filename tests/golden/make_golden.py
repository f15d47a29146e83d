"""Regenerates the fixtures in this directory.  Run from the repo root:

    python tests/golden/make_golden.py

Inputs: the CPU oracle (oracle/) and, when present, the reference's shipped render
/root/reference/media/rayTracing.png, from which only derived numbers (row means, exact sky
pixels) are stored - the image itself is not copied.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import ctypes as C  # noqa: E402

L = O.lib()

# ---- Nim std/random + randomScene
r = O.nim_default()
raw = [L.orc_nimrand_next(C.byref(r)) for _ in range(4)]
r = O.nim_default()
unit = [L.orc_nimrand_rand(C.byref(r), 1.0) for _ in range(4)]
arr, n, draws, end = O.random_scene()
kinds = [arr[j].kind for j in range(n)]


def sph(j):
    s = arr[j]
    return {"index": j, "kind": s.kind, "center": list(s.center), "radius": s.radius,
            "albedo": list(s.albedo), "fuzz": s.fuzz, "ir": s.ir}


seeds = {}
for jump in (0, 1):
    g = O.NimRand()
    L.orc_nimrand_init(C.byref(g), 0xE7, jump)
    seeds["jump" if jump else "legacy"] = {"a0": hex(g.a0), "a1": hex(g.a1),
                                           "first_rand": L.orc_nimrand_rand(C.byref(g), 1.0)}
json.dump({
    "default_state_raw_u64": [hex(x) for x in raw],
    "default_state_rand_1": unit,
    "random_scene": {"n": n, "draws": draws, "end_a0": hex(end.a0), "end_a1": hex(end.a1),
                     "n_lambertian": kinds.count(0), "n_metal": kinds.count(1),
                     "n_dielectric": kinds.count(2), "first_dielectric": kinds.index(2),
                     "spheres": [sph(j) for j in (0, 1, 2, 41, 480, 481, 482, 483)]},
    "randomize_0xE7": seeds,
}, open(os.path.join(HERE, "nim_random.json"), "w"), indent=1)

# ---- numbers derived from the reference's own output image
png = "/root/reference/media/rayTracing.png"
if os.path.exists(png):
    from PIL import Image
    im = np.asarray(Image.open(png).convert("RGB"))
    rows = [0, 50, 100, 150, 300, 400, 500, 600, 700, 780]
    json.dump({
        "source": "media/rayTracing.png of the reference (1200x800, spp 500, depth 50)",
        "shape": list(im.shape),
        "mean_rgb": im.reshape(-1, 3).mean(0).tolist(),
        "row0_unique": np.unique(im[0], axis=0).tolist(),
        "pixels": {"50,600": im[50, 600].tolist(), "100,100": im[100, 100].tolist(),
                   "150,1199": im[150, 1199].tolist()},
        "rows": rows,
        "row_mean_rgb": [im[y].mean(0).tolist() for y in rows],
        # 8-pixel box means of the green channel: the sphere layout along the row
        "row_profile_g_box8": [im[y, :, 1].reshape(150, 8).mean(1).round(2).tolist() for y in rows],
    }, open(os.path.join(HERE, "reference_png_stats.json"), "w"), indent=1)

    # 8 x 8 box means of the whole image (150 x 100 boxes): each mean is a multiple of 1/64, stored x 64
    thumb = im.astype(np.uint32).reshape(100, 8, 150, 8, 3).sum((1, 3)).astype(np.uint16)
    # which boxes lie (>= 95 % of their pixels) on which object, from the oracle's primary hit ids: the three big
    # spheres of raytracer.nim:104-111, the small spheres by material, the ground, the sky
    _, _, hid, _ = O.render_ref64(arr, n, O.book_camera(1.5), 1200, 800, 1, 1)
    ids = hid.reshape(100, 8, 150, 8)
    small = lambda kind: [j for j in range(1, n - 3) if arr[j].kind == kind]  # noqa: E731
    sets = {"glass": [n - 3], "brown": [n - 2], "bigmetal": [n - 1], "smalllambert": small(0), "smallmetal": small(1),
            "smallglass": small(2), "ground": [0], "sky": [-1]}
    masks = {"mask_" + k: np.packbits(np.isin(ids, v).mean((1, 3)) >= 0.95) for k, v in sets.items()}
    np.savez_compressed(os.path.join(HERE, "reference_png_thumb.npz"), thumb_x64=thumb, **masks)

# ---- FP32 mirror fixtures the GPU must reproduce bit for bit
cam = O.book_camera(64 / 48)
rgb, acc, hid, seg = O.render_mirror32(arr, n, cam, 64, 48, 1, 1)
np.savez_compressed(os.path.join(HERE, "mirror32_64x48_spp1_d1.npz"), rgb=rgb, hit_id=hid, accum=acc,
                    segments=np.uint64(seg))
cam = O.book_camera(96 / 64)
rgb, acc, hid, seg = O.render_mirror32(arr, n, cam, 96, 64, 8, 50)
np.savez_compressed(os.path.join(HERE, "mirror32_96x64_spp8_d50.npz"), rgb=rgb, hit_id=hid, accum=acc,
                    segments=np.uint64(seg))
m_arr, m_n = O.mix_of_spheres()
mcam = O.book_camera(80 / 60, look_from=(3, 3, 2), look_at=(0, 0, -1), vfov=20.0, aperture=2.0,
                     focus=(27.0 ** 0.5))
rgb, acc, hid, seg = O.render_mirror32(m_arr, m_n, mcam, 80, 60, 16, 50)
np.savez_compressed(os.path.join(HERE, "mirror32_mix_80x60_spp16_d50.npz"), rgb=rgb, hit_id=hid, accum=acc,
                    segments=np.uint64(seg))

# ---- FP64 faithful fixture for tolerance tests (100x67, spp 64)
cam = O.book_camera(100 / 67)
rgb, acc, hid, seg = O.render_ref64(arr, n, cam, 100, 67, 64, 50)
np.savez_compressed(os.path.join(HERE, "ref64_100x67_spp64_d50.npz"), rgb=rgb, hit_id=hid,
                    accum=acc.astype(np.float32), segments=np.uint64(seg))
print("golden fixtures written to", HERE)
