"""Unit tests of the FP64 restatement against hand-derived answers and against the host-side
Python restatement of basetypes.nim (an independent second implementation)."""
import ctypes as C
import math

import numpy as np
import pytest

from raytracinginoneweekend_amd import basetypes as B


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return math.inf if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def _sphere(oracle, center, radius, kind=0, albedo=(0.5, 0.5, 0.5), fuzz=0.0, ir=1.5):
    arr, _ = oracle.spheres_from_arrays([center], [radius], [kind], [albedo], [fuzz], [ir])
    return arr


def _hit(oracle, s, o, d, tmin=0.001, tmax=math.inf):
    t = C.c_double()
    p = (C.c_double * 3)()
    nrm = (C.c_double * 3)()
    ff = C.c_int()
    ok = oracle.lib().orc_ref64_sphere_hit(s, oracle.vec3(*o), oracle.vec3(*d), tmin, tmax, C.byref(t), p, nrm,
                                           C.byref(ff))
    return ok, t.value, list(p), list(nrm), ff.value


def test_sphere_hit_front_and_inside(oracle):
    s = _sphere(oracle, (0, 0, -5), 1.0)
    ok, t, p, n, ff = _hit(oracle, s, (0, 0, 0), (0, 0, -2))  # |dir| = 2: t is in units of |dir|
    assert ok and t == 2.0 and p == [0, 0, -4] and n == [0, 0, 1] and ff == 1
    ok, t, p, n, ff = _hit(oracle, s, (0, 0, -5), (0, 0, -1))  # from the centre: far root, back face
    assert ok and t == 1.0 and n == [0, 0, 1] and ff == 0
    assert not _hit(oracle, s, (0, 3, 0), (0, 0, -1))[0]         # discriminant < 0
    assert not _hit(oracle, s, (0, 0, 0), (0, 0, 1))[0]          # both roots negative
    assert not _hit(oracle, s, (0, 0, 0), (0, 0, -1), tmax=3.9)[0]  # t_max < root on both


def test_sphere_hit_negative_radius_flips_normal(oracle):
    s = _sphere(oracle, (0, 0, -5), -1.0)
    ok, t, p, n, ff = _hit(oracle, s, (0, 0, 0), (0, 0, -1))
    assert ok and t == 4.0 and n == [0, 0, 1] and ff == 0  # outward normal points inwards


def test_sphere_hit_tmin_boundary_is_inclusive(oracle):
    s = _sphere(oracle, (0, 0, -5), 1.0)
    assert _hit(oracle, s, (0, 0, 0), (0, 0, -1), tmin=4.0)[1] == 4.0   # root == t_min accepted
    assert _hit(oracle, s, (0, 0, 0), (0, 0, -1), tmin=4.0000001)[1] == 6.0
    assert _hit(oracle, s, (0, 0, 0), (0, 0, -1), tmax=4.0)[1] == 4.0   # root == t_max accepted


def test_world_hit_ties_go_to_later_index(oracle):
    arr, n = oracle.spheres_from_arrays([(0, 0, -5)] * 3, [1.0] * 3, [0] * 3, [(1, 1, 1)] * 3, [0] * 3, [1.5] * 3)
    t = C.c_double()
    idx = oracle.lib().orc_ref64_world_hit(arr, n, oracle.vec3(0, 0, 0), oracle.vec3(0, 0, -1), 0.001, math.inf,
                                           C.byref(t))
    assert idx == 2 and t.value == 4.0
    arr, n = oracle.spheres_from_arrays([(0, 0, -5), (0, 0, -3), (0, 0, -9)], [1.0] * 3, [0] * 3, [(1, 1, 1)] * 3,
                                        [0] * 3, [1.5] * 3)
    idx = oracle.lib().orc_ref64_world_hit(arr, n, oracle.vec3(0, 0, 0), oracle.vec3(0, 0, -1), 0.001, math.inf,
                                           C.byref(t))
    assert idx == 1 and t.value == 2.0
    idx = oracle.lib().orc_ref64_world_hit(arr, 0, oracle.vec3(0, 0, 0), oracle.vec3(0, 0, -1), 0.001, math.inf,
                                           C.byref(t))
    assert idx == -1


def _draws(oracle, key, k0, count):
    return [oracle.lib().orc_draw_unit(key, k0 + i) for i in range(count)]


def _scatter(oracle, s, in_o, in_d, p, n, front, key, k):
    kk = C.c_uint32(k)
    att = (C.c_double * 3)()
    oo = (C.c_double * 3)()
    od = (C.c_double * 3)()
    ok = oracle.lib().orc_ref64_scatter(s, oracle.vec3(*in_o), oracle.vec3(*in_d), oracle.vec3(*p), oracle.vec3(*n),
                                        int(front), key, C.byref(kk), att, oo, od)
    return ok, list(att), list(oo), list(od), kk.value


def _unit_ball_from(draws):
    """randomInUnitSphere on an explicit draw sequence, via the Python restatement."""
    i = 0
    while True:
        v = B.Vec3(*(d * 2.0 + -1.0 for d in draws[i:i + 3]))
        i += 3
        if B.length_squared(v) >= 1:
            continue
        return v, i


@pytest.mark.parametrize("sample", [0, 1, 7])
def test_scatter_lambertian_and_metal_follow_basetypes(oracle, sample):
    key = (C.c_uint32 * 2)()
    oracle.lib().orc_path_key(0xE7, 1234, sample, key)
    draws = _draws(oracle, key, 5, 64)
    n = B.unitVector(B.Vec3(0.3, 0.9, -0.2))
    p = (1.0, 2.0, 3.0)
    v, used = _unit_ball_from(draws)
    s = _sphere(oracle, (0, 0, 0), 1.0, kind=oracle.LAMBERTIAN, albedo=(0.1, 0.2, 0.3))
    ok, att, oo, od, k = _scatter(oracle, s, (0, 0, 0), (0, -1, 0), p, n.arr, True, key, 5)
    exp = n + B.unitVector(v)
    assert ok == 1 and att == [0.1, 0.2, 0.3] and oo == list(p) and od == exp.arr and k == 5 + used
    s = _sphere(oracle, (0, 0, 0), 1.0, kind=oracle.METAL, albedo=(0.7, 0.6, 0.5), fuzz=0.25)
    d_in = B.Vec3(0.5, -2.0, 0.1)
    ok, att, oo, od, k = _scatter(oracle, s, (0, 0, 0), d_in.arr, p, n.arr, True, key, 5)
    exp = B.reflect(B.unitVector(d_in), n) + 0.25 * v
    assert od == exp.arr and att == [0.7, 0.6, 0.5] and k == 5 + used
    assert ok == int(B.dot(exp, n) > 0)


def test_scatter_dielectric_branches(oracle):
    key = (C.c_uint32 * 2)()
    oracle.lib().orc_path_key(0xE7, 99, 3, key)
    s = _sphere(oracle, (0, 0, 0), 1.0, kind=oracle.DIELECTRIC, ir=1.5)
    n = B.Vec3(0, 1, 0)
    # total internal reflection from inside at a grazing angle: no draw is consumed
    d_in = B.Vec3(1.0, -0.2, 0.0)
    ok, att, oo, od, k = _scatter(oracle, s, (0, 0, 0), d_in.arr, (0, 0, 0), n.arr, False, key, 0)
    assert ok == 1 and att == [1.0, 1.0, 1.0] and k == 0
    assert od == B.reflect(B.unitVector(d_in), n).arr
    # front face, near-normal incidence: one draw decides between reflect and refract
    d_in = B.Vec3(0.1, -1.0, 0.0)
    u = B.unitVector(d_in)
    cos_t = min(B.dot(-u, n), 1.0)
    r0 = ((1 - 1 / 1.5) / (1 + 1 / 1.5)) ** 2
    refl = r0 + (1 - r0) * math.pow(1 - cos_t, 5)
    seen = set()
    for k0 in range(40):
        xi = oracle.lib().orc_draw_unit(key, k0)
        ok, att, oo, od, k = _scatter(oracle, s, (0, 0, 0), d_in.arr, (0, 0, 0), n.arr, True, key, k0)
        assert k == k0 + 1
        exp = B.reflect(u, n) if refl > xi else B.refract(u, n, 1.0 / 1.5)
        seen.add(refl > xi)
        assert od == exp.arr
    assert False in seen  # refraction is the common case here


def test_get_ray_follows_camera_nim(oracle):
    cam = oracle.book_camera(1.5)
    key = (C.c_uint32 * 2)()
    oracle.lib().orc_path_key(1, 2, 3, key)
    k = C.c_uint32(0)
    o = (C.c_double * 3)()
    d = (C.c_double * 3)()
    oracle.lib().orc_ref64_get_ray(C.byref(cam), 0.25, 0.75, key, C.byref(k), o, d)
    draws = _draws(oracle, key, 0, 32)
    i = 0
    while True:
        x, y = draws[i] * 2.0 + -1.0, draws[i + 1] * 2.0 + -1.0
        i += 2
        if x * x + y * y + 0.0 < 1.0:
            break
    assert k.value == i
    rd = cam.lens_radius * B.Vec3(x, y, 0)
    off = B.Vec3(*cam.u) * rd.x + B.Vec3(*cam.v) * rd.y
    assert list(o) == (B.Vec3(*cam.origin) + off).arr
    exp = B.Vec3(*cam.lower_left) + 0.25 * B.Vec3(*cam.horizontal) + 0.75 * B.Vec3(*cam.vertical) \
        - B.Vec3(*cam.origin) - off
    assert list(d) == exp.arr


def test_quantise_matches_write_color(oracle):
    q = oracle.lib().orc_ref64_quantise
    assert q(0.0, 10) == 0 and q(10.0, 10) == 255 and q(1e9, 10) == 255
    assert q(2.5, 10) == int(256 * math.sqrt(0.25)) == 128
    assert q(500 * 0.7437, 500) == int(256 * min(math.sqrt(500 * 0.7437 * (1.0 / 500)), 0.999))


def test_counter_rng_statistics(oracle):
    """Uniformity and independence of the per-path streams."""
    L = oracle.lib()
    key = (C.c_uint32 * 2)()
    vals = np.empty((256, 64))
    for s in range(256):
        L.orc_path_key(0xE7, 4242, s, key)
        vals[s] = [L.orc_draw_unit(key, k) for k in range(64)]
    assert abs(vals.mean() - 0.5) < 0.01 and abs(vals.var() - 1 / 12) < 0.004
    hist = np.histogram(vals, bins=16, range=(0, 1))[0]
    chi2 = ((hist - vals.size / 16) ** 2 / (vals.size / 16)).sum()
    assert chi2 < 45  # 15 dof, p ~ 1e-4
    # successive draws of a stream, same draw of successive samples, same draw of adjacent pixels
    assert abs(np.corrcoef(vals[:, :-1].ravel(), vals[:, 1:].ravel())[0, 1]) < 0.02
    assert abs(np.corrcoef(vals[:-1].ravel(), vals[1:].ravel())[0, 1]) < 0.02
    pix = np.empty((256, 16))
    for p in range(256):
        L.orc_path_key(0xE7, 1000 + p, 0, key)
        pix[p] = [L.orc_draw_unit(key, k) for k in range(16)]
    assert abs(np.corrcoef(pix[:-1].ravel(), pix[1:].ravel())[0, 1]) < 0.03
    # rejection rates of the disk and ball samplers come out at pi/4 and pi/6
    pts = 2 * vals.reshape(-1, 2) - 1
    assert abs((np.sum(pts ** 2, 1) < 1).mean() - math.pi / 4) < 0.02
    pts = 2 * vals[:, :63].reshape(-1, 3) - 1
    assert abs((np.sum(pts ** 2, 1) < 1).mean() - math.pi / 6) < 0.025
    # a draw is a multiple of 2^-24 in [0, 1)
    assert ((vals * 2 ** 24) % 1 == 0).all() and vals.min() >= 0 and vals.max() < 1
    # different seeds give different streams
    L.orc_path_key(0xE8, 4242, 0, key)
    assert L.orc_draw_unit(key, 0) != vals[0, 0]


def test_mirror32_tracks_ref64(oracle, book_scene):
    """FP32 device arithmetic against the FP64 restatement on the same random streams."""
    arr, n = book_scene
    W, H, spp = 120, 80, 48
    cam = oracle.book_camera(W / H)
    r64, a64, h64, s64 = oracle.render_ref64(arr, n, cam, W, H, spp, 50)
    r32, a32, h32, s32 = oracle.render_mirror32(arr, n, cam, W, H, spp, 50)
    mse = np.mean((r64.astype(np.float64) - r32.astype(np.float64)) ** 2)
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    mae = np.mean(np.abs(r64.astype(np.int32) - r32.astype(np.int32)))
    assert psnr >= 40.0, psnr          # the stated bar
    assert mae <= 0.25, mae            # mean absolute error per channel, in 8-bit levels
    assert np.mean(h64 != h32) < 1e-3  # primary hits differ only on silhouette pixels
    assert abs(s64 - s32) / s64 < 2e-3
    lin64 = a64 / spp
    lin32 = a32.astype(np.float64) / (spp * 2.0 ** 40)
    assert abs(lin64.mean() - lin32.mean()) < 2e-4


def test_mirror32_small_scenes_track_ref64(oracle):
    arr, n = oracle.mix_of_spheres()
    cam = oracle.book_camera(4 / 3, look_from=(3, 3, 2), look_at=(0, 0, -1), vfov=20.0, aperture=2.0,
                             focus=math.sqrt(27.0))
    r64 = oracle.render_ref64(arr, n, cam, 80, 60, 64, 50)
    r32 = oracle.render_mirror32(arr, n, cam, 80, 60, 64, 50)
    assert _psnr(r64[0], r32[0]) >= 40.0
    arr, n = oracle.red_blue()
    cam = oracle.book_camera(2.0, look_from=(0, 0, 0), look_at=(0, 0, -1), vfov=90.0, aperture=0.0, focus=1.0)
    r64 = oracle.render_ref64(arr, n, cam, 64, 32, 16, 50)
    r32 = oracle.render_mirror32(arr, n, cam, 64, 32, 16, 50)
    assert (r64[2] == r32[2]).mean() > 0.995
    assert _psnr(r64[0], r32[0]) >= 40.0


def test_mirror32_tracks_ref64_on_a_wide_scene(oracle):
    """A 41 x 41 grid (1.6k spheres, centres out to |c| = 29): the filter margin grows with the
    extent of the scene; hits must still agree with the FP64 restatement."""
    arr, n, _, _ = oracle.random_scene(half_extent=20)
    W, H, spp = 90, 60, 8
    cam = oracle.book_camera(W / H)
    r64 = oracle.render_ref64(arr, n, cam, W, H, spp, 50)
    r32 = oracle.render_mirror32(arr, n, cam, W, H, spp, 50)
    assert _psnr(r64[0], r32[0]) >= 40.0
    assert np.mean(r64[2] != r32[2]) < 2e-3
    assert abs(r64[3] - r32[3]) / r64[3] < 5e-3


def test_sphere_screen_never_drops_a_true_hit(oracle, book_scene):
    """The packed screen (n^2 + (2o).c - k >= |o|^2 - margin) that the GPU and its CPU mirror put in
    front of Sphere.hit (hitrecord.nim:56-63) only ever adds work: with the screen disabled the mirror
    produces the same sums bit for bit, no sphere with an exact discriminant >= 0 falls below the
    threshold, and the closest call still has most of the margin left."""
    cases = []
    arr, n = book_scene
    cases.append(("book", arr, n, oracle.book_camera(1.5), 120, 80, 6))
    a3, n3, _, _ = oracle.random_scene(lambert_below=0.0, metal_below=0.5)
    cases.append(("metal+glass", a3, n3, oracle.book_camera(1.5), 90, 60, 6))
    a5, n5, _, _ = oracle.random_scene(half_extent=50)        # 10k spheres, centres out to |c| = 71
    cases.append(("10k", a5, n5, oracle.book_camera(1.5), 48, 32, 2))
    am, nm = oracle.mix_of_spheres()                            # hollow glass (negative radius), r = 100 ground
    cases.append(("mix", am, nm, oracle.book_camera(4 / 3, look_from=(3, 3, 2), look_at=(0, 0, -1), vfov=20.0, aperture=2.0,
                                                   focus=math.sqrt(27.0)), 80, 60, 16))
    # camera far from the origin: |o| dominates the margin
    cases.append(("far camera", arr, n, oracle.book_camera(1.5, look_from=(130, 20, 30), vfov=2.0, focus=133.0), 90, 60, 4))
    for name, a, k, cam, W, H, spp in cases:
        plain = oracle.render_mirror32(a, k, cam, W, H, spp, 50)
        audited, hits, dropped, slack = oracle.render_mirror32_audited(a, k, cam, W, H, spp, 50)
        assert hits > 1000, name
        assert dropped == 0, f"{name}: the screen dropped {dropped} of {hits} true hits"
        assert slack >= 0.5, f"{name}: only {slack:.3f} of the margin left"
        for x, y in zip(plain[:3], audited[:3]):
            assert (x == y).all(), name
        assert plain[3] == audited[3], name


def test_mirror_scatter_tracks_the_fp64_scatter(oracle):
    """The binary32 restatement of scatter (hitrecord.nim:82-132) against the binary64 one on the same draws:
    same decisions except within rounding of a threshold, directions within binary32 rounding."""
    import ctypes as C
    import numpy as np
    L = oracle.lib()
    rng = np.random.default_rng(3)
    n = 1500
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    key = (C.c_uint32 * 2)()
    zero = (C.c_double * 3)(0.0, 0.0, 0.0)
    other = 0
    kinds = [0, 0, 0]
    for i in range(n):
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        d = rng.normal(size=3)
        if d @ nrm > 0:
            d = -d
        d *= np.exp(rng.uniform(-2, 2))
        kind = int(rng.integers(0, 3))
        arr, _ = oracle.spheres_from_arrays([(0, 0, 0)], [1.0], [kind], [rng.uniform(0, 1, 3)], [rng.uniform(0, 1)], [rng.uniform(1.1, 2.4)])
        front = int(rng.integers(0, 2))
        a32, o32 = (C.c_double * 3)(), (C.c_double * 3)()
        ok32 = L.orc_mirror32_scatter(arr, dp(d), dp(nrm), front, 0xE7, i, a32, o32)
        L.orc_path_key(0xE7, i, 0, key)
        k = C.c_uint32(0)
        a64, oo, o64 = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
        ok64 = L.orc_ref64_scatter(arr, zero, dp(d), zero, dp(nrm), front, key, C.byref(k), a64, oo, o64)
        if ok32 != ok64 or not np.allclose(list(o32), list(o64), rtol=0, atol=2e-5 * max(1.0, np.abs(list(o64)).max())):
            other += 1
            continue
        kinds[kind] += 1
        assert np.allclose(list(a32), list(a64), rtol=1e-6, atol=0)
    assert other <= 3 and min(kinds) > 300


def test_slab_test_of_camera_rays_is_conservative():
    """The renderer puts a slab test in front of the sphere screen for camera rays: component k of (c - o) x u,
    evaluated in binary32 on rounded operands with two FMAs per (ray, sphere) pair, and the smallest |value| of a
    group of four spheres against ONE radius per group, the largest |r| of the four (raised to 2^-8 of the scene's
    extent).  The slack for rounding, delta = 2^-14 (|o|_1 + cmax), enters as a scale of the ray's operands,
    s = 1 / (1 + delta / rmin) with rmin the smallest group radius.  Restated here in numpy (binary32 operations, fused
    where the kernel fuses) and held against the exact geometry in binary64: no sphere whose line distance is within
    its radius may fail, along any axis, for near and far cameras and small and large scenes, equal and mixed radii
    within a group; the worst case must leave most of the slack unused."""
    import numpy as np
    f32 = np.float32

    def fma32(a, b, c):   # a * b is exact in binary64; one rounding to binary64 before the one to binary32 is harmless here
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

    rng = np.random.default_rng(9)
    worst = 0.0
    for spread, cam_dist, rad, mixed in ((11.0, 14.0, 0.2, False), (11.0, 300.0, 0.2, True), (200.0, 50.0, 3.0, True), (1.0, 2.0, 0.05, False),
                                         (30.0, 5000.0, 20.0, True), (11.0, 14.0, 0.0005, True)):
        n = 400
        c64 = np.column_stack([rng.uniform(-spread, spread, n), rng.uniform(0, 0.3 * spread, n), rng.uniform(-spread, spread, n)])
        r64 = rad * (np.exp(rng.uniform(-1, 1, n)) if mixed else np.ones(n)) * rng.choice([1.0, -1.0], n)
        cmax = f32((np.linalg.norm(c64.astype(f32).astype(np.float64), axis=1) + np.abs(r64)).max())
        m = 3000
        o64 = rng.normal(0, 1, (m, 3)); o64 *= cam_dist / np.linalg.norm(o64, axis=1, keepdims=True)
        target = c64[rng.integers(0, n, m)] + rng.normal(0, 1, (m, 3)) * np.abs(r64).mean()   # aim near spheres: many grazing rays
        d64 = (target - o64) * np.exp(rng.uniform(-2, 2, (m, 1)))
        # exact geometry: distance from the centre to the ray's line
        u64 = d64 / np.linalg.norm(d64, axis=1, keepdims=True)
        oc = c64[None, :, :] - o64[:, None, :]
        dist = np.linalg.norm(np.cross(oc, u64[:, None, :]), axis=2)
        hit = dist <= np.abs(r64)[None, :]
        # the kernel's operands: rounded origin, direction normalised in binary32, rounded centres and radii
        o = o64.astype(f32); d = d64.astype(f32)
        inv = (f32(1.0) / np.sqrt(fma32(d[:, 2], d[:, 2], fma32(d[:, 1], d[:, 1], d[:, 0] * d[:, 0])))).astype(f32)
        u = (d * inv[:, None]).astype(f32)
        c = c64.astype(f32); r = r64.astype(f32)
        # the host's table: one radius per group of four, floored, and the reciprocal of the smallest
        rg = np.maximum(np.abs(r).reshape(-1, 4).max(axis=1), cmax * f32(2.0 ** -8)).astype(f32)
        inv_rmin = f32(1.0) / rg.min()
        delta = ((np.abs(o[:, 0]) + np.abs(o[:, 1]) + np.abs(o[:, 2]) + cmax) * f32(2.0 ** -14)).astype(f32)
        # hardware reciprocal: within one ulp of the exact one; the kernel backs off by 2^-20
        s_scale = ((f32(1.0) / fma32(delta, np.full(m, inv_rmin, f32), np.full(m, 1.0, f32))).astype(f32) * f32(1.0 - 2.0 ** -20)).astype(f32)
        for axis in range(3):
            i, j = (axis + 1) % 3, (axis + 2) % 3
            sui = (s_scale * u[:, i]).astype(f32); suj = (s_scale * u[:, j]).astype(f32)
            mm = (s_scale * fma32(o[:, j], u[:, i], -(o[:, i] * u[:, j]).astype(f32))).astype(f32)
            t = fma32(np.broadcast_to(c[None, :, j], (m, n)), np.broadcast_to(-sui[:, None], (m, n)), np.broadcast_to(mm[:, None], (m, n)))
            v = np.abs(fma32(np.broadcast_to(c[None, :, i], (m, n)), np.broadcast_to(suj[:, None], (m, n)), t))
            group_min = v.reshape(m, -1, 4).min(axis=2)
            passes = np.repeat(group_min <= rg[None, :], 4, axis=1)   # the whole group goes on to the screen or none of it
            assert passes[hit].all(), (spread, cam_dist, axis, int((~passes[hit]).sum()))
            # how much of the slack rounding used: the value the test saw against the exact distance (which bounds the
            # component) scaled the same way
            over = (v[hit] - (np.broadcast_to(s_scale[:, None].astype(np.float64), (m, n)) * dist)[hit])
            room = (np.broadcast_to((s_scale.astype(np.float64) * delta.astype(np.float64))[:, None], (m, n)))[hit]
            worst = max(worst, float((over / room).max()))
        assert hit.sum() > 1000
    assert worst < 0.1, worst   # rounding takes less than a tenth of the slack the scale provides


def test_interval_test_of_camera_rays_is_conservative():
    """The renderer's default screen for camera rays: per sphere a range [s_lo, s_hi] of the image coordinate s (the
    first argument of getRay, camera.nim:34) outside which no camera ray can meet the sphere, worked out by the host
    from scene and camera.  Seen along the camera's up axis the lens is a segment |x| <= R on one line, the focus plane
    a parallel line a distance f away, the sphere a circle (X, Z, rho); the lines from a lens point through the focus
    line that touch the circle are the two tangents, and over the lens the extremes sit at its ends.  Restated here in
    numpy and held against the exact geometry of camera.nim:34-39 rays in binary64: for random look-at cameras (near,
    far, inside the field, wide and closed lenses) and random spheres, every ray whose LINE passes within |r| of a
    centre has its s inside that sphere's range (widened as the kernel's binary32 subtraction needs) and, by the same
    construction along the image's other axis, its t inside the sphere's range of t; spheres wholly
    behind the lens are met by no forward ray, and a sphere that reaches across the lens plane gets the whole axis."""
    import numpy as np
    rng = np.random.default_rng(17)
    f32 = np.float32
    checked = inside = 0
    for trial in range(40):
        spread = rng.choice([3.0, 11.0, 60.0])
        n = 300
        c = np.column_stack([rng.uniform(-spread, spread, n), rng.uniform(-0.2 * spread, 0.4 * spread, n), rng.uniform(-spread, spread, n)])
        r = rng.choice([0.05, 0.2, 1.0, 4.0], n) * np.exp(rng.uniform(-0.5, 0.5, n)) * rng.choice([1.0, -1.0], n)
        look_from = rng.normal(0, 1, 3) * rng.choice([0.3, 1.0, 3.0]) * spread
        look_at = rng.normal(0, 0.3, 3) * spread
        vup = np.array([0.0, 1.0, 0.0]) if trial % 3 else rng.normal(0, 1, 3)
        vfov, aspect = rng.uniform(10, 100), rng.choice([1.0, 1.5, 16 / 9])
        focus = rng.uniform(0.3, 2.0) * np.linalg.norm(look_from - look_at)
        R = rng.choice([0.0, 0.05, 0.5, 2.0])
        # initCamera (camera.nim:12-31)
        h = np.tan(np.radians(vfov) / 2)
        w = (look_from - look_at) / np.linalg.norm(look_from - look_at)
        u = np.cross(vup, w); u /= np.linalg.norm(u)
        v = np.cross(w, u)
        hor, ver = focus * aspect * 2 * h * u, focus * 2 * h * v
        llc = look_from - hor / 2 - ver / 2 - focus * w
        # the host's table, for s (across the image: drop v) and for t (up the image: drop u)
        hu, hv = np.linalg.norm(hor), np.linalg.norm(ver)
        tables = []
        for across, extent in ((u, hu), (v, hv)):
            X, Z = (c - look_from) @ across, -((c - look_from) @ w)
            rho = np.abs(r) * (1 + 1e-9) + 4e-9 * (hu + hv + focus + R + np.abs(X) + np.abs(Z)) * (1 + np.abs(Z) / focus)
            never = Z < -rho
            always = ~never & ~(Z > rho * (1 + 1e-6))
            k = Z * Z - rho * rho
            with np.errstate(invalid="ignore", divide="ignore"):
                lo, hi = np.full(n, np.inf), np.full(n, -np.inf)
                for xl in (-R * (1 + 1e-9), R * (1 + 1e-9)):
                    A = X - xl
                    root = rho * np.sqrt(A * A + k)
                    lo = np.minimum(lo, (xl + focus * (A * Z - root) / k) / extent + 0.5)
                    hi = np.maximum(hi, (xl + focus * (A * Z + root) / k) / extent + 0.5)
            pad = 2.0 ** -19 * (np.abs(lo) + np.abs(hi) + 2)
            mid = (0.5 * (lo + hi)).astype(f32)
            half = (np.maximum(hi - mid.astype(np.float64), mid.astype(np.float64) - lo) + pad).astype(f32)
            tables.append((never, always, mid, np.nextafter(half, f32(np.inf))))
        # rays of getRay (camera.nim:34-39): lens point, focus-plane point; s and t a little beyond [0, 1] as the jitter allows
        m = 4000
        s, t = rng.uniform(-0.01, 1.01, m), rng.uniform(-0.01, 1.01, m)
        ang, rad = rng.uniform(0, 2 * np.pi, m), np.sqrt(rng.uniform(0, 1, m))
        rd = R * np.column_stack([rad * np.cos(ang), rad * np.sin(ang)])
        off = rd[:, :1] * u[None, :] + rd[:, 1:] * v[None, :]
        o = look_from[None, :] + off
        d = llc[None, :] + s[:, None] * hor[None, :] + t[:, None] * ver[None, :] - look_from[None, :] - off
        dn = d / np.linalg.norm(d, axis=1, keepdims=True)
        oc = c[None, :, :] - o[:, None, :]
        along = np.einsum("mnk,mk->mn", oc, dn)
        dist = np.linalg.norm(oc - along[:, :, None] * dn[:, None, :], axis=2)
        meets = dist <= np.abs(r)[None, :]                      # the ray's LINE meets the sphere
        with np.errstate(invalid="ignore"):                     # ... at a point in front of the ray's origin (the far root)
            forward = meets & (along + np.sqrt(np.maximum(np.abs(r)[None, :] ** 2 - dist ** 2, 0.0)) > 0)
        # the kernel's tests in binary32: both must let every such pair through
        for (never, always, mid, half), coord in zip(tables, (s, t)):
            passes = np.abs(mid[None, :] - coord.astype(f32)[:, None]) <= half[None, :]
            passes = np.where(always[None, :], True, np.where(never[None, :], False, passes))
            assert passes[forward].all(), (trial, int((~passes[forward]).sum()))
            assert not forward[:, never].any()
            inside += int(always.sum())
        checked += int(forward.sum())
    assert checked > 20000 and inside > 100
