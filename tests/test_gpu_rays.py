"""The reference's per-ray procs on the GPU - hit (hittables.nim:79, hitrecord.nim:55), rayColor
(raytracer.nim:5), getRay (camera.nim:34), scatter (hitrecord.nim:127) - through rtow_trace_rays /
rtow_get_rays / rtow_scatter_rays, against the oracle's
single-ray probes.  The first world.hit of a ray is evaluated in binary64 with the reference's operations, so
hit index and t must EQUAL the FP64 restatement's (oracle/ref64.c); colours must equal the renderer's CPU
mirror bit for bit, and the FP64 restatement's exactly at depth 1 and closely at depth 50."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _R():
    from raytracinginoneweekend_amd import raytracer
    return raytracer


def _rays(rng, n, scale=14.0):
    """Rays that start around the scene and point roughly at it (most hit something), with directions of any length."""
    o = rng.normal(0.0, scale / 2, (n, 3))
    o[:, 1] = np.abs(o[:, 1]) + 0.05
    target = rng.normal(0.0, 4.0, (n, 3)) * np.array([1.0, 0.05, 1.0]) + np.array([0.0, 0.3, 0.0])
    d = (target - o) * np.exp(rng.uniform(-3.0, 3.0, (n, 1)))
    return o, d


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_hit_equals_the_fp64_restatement(gpu_renderer, oracle, book_scene):
    arr, n = book_scene
    R = _R()
    gpu_renderer.setWorld(R.randomScene())
    L = oracle.lib()
    rng = np.random.default_rng(11)
    o, d = _rays(rng, 3000)
    for t_min, t_max in ((0.001, math.inf), (0.25, 0.9), (1e-9, 1e9)):
        _, hid, t = gpu_renderer.traceRays(o, d, 0, t_min=t_min, t_max=t_max, want_color=False)
        hits = 0
        for i in range(len(o)):
            tt = C.c_double(0.0)
            want = L.orc_ref64_world_hit(arr, n, _dp(o[i]), _dp(d[i]), t_min, t_max, C.byref(tt))
            assert hid[i] == want, (i, hid[i], want)
            if want >= 0:
                assert t[i] == tt.value, (i, t[i], tt.value)   # same bits: t in units of |dir|
                hits += 1
            else:
                assert t[i] == math.inf
        assert hits > (1000 if t_max == math.inf else 50)
    # the hollow glass shell (negative radius), the radius-100 ground and two coincident spheres (later index wins)
    arr2, n2 = oracle.spheres_from_arrays([(0, -100.5, -1), (0, 0, -1), (-1, 0, -1), (-1, 0, -1), (1, 0, -1), (1, 0, -1)],
                                          [100, 0.5, 0.5, -0.4, 0.5, 0.5], [0, 0, 2, 2, 1, 0],
                                          [(0.8, 0.8, 0.0), (0.1, 0.2, 0.5), (1, 1, 1), (1, 1, 1), (0.8, 0.6, 0.2), (0.3, 0.3, 0.3)],
                                          [0, 0, 0, 0, 1.0, 0], [1.0, 1.0, 1.5, 1.5, 1.0, 1.0])
    with R.Renderer(0) as r:
        r.setWorld({"center_x": [0, 0, -1, -1, 1, 1], "center_y": [-100.5, 0, 0, 0, 0, 0], "center_z": [-1] * 6,
                    "radius": [100, 0.5, 0.5, -0.4, 0.5, 0.5], "kind": [0, 0, 2, 2, 1, 0],
                    "albedo": [(0.8, 0.8, 0.0), (0.1, 0.2, 0.5), (1, 1, 1), (1, 1, 1), (0.8, 0.6, 0.2), (0.3, 0.3, 0.3)],
                    "fuzz": [0, 0, 0, 0, 1.0, 0], "ir": [1.0, 1.0, 1.5, 1.5, 1.0, 1.0]})
        o2 = rng.normal(0.0, 1.5, (2000, 3)) + np.array([0.0, 0.6, 1.0])
        d2 = np.array([0.0, 0.0, -1.0]) + rng.normal(0.0, 0.6, (2000, 3))
        _, hid, t = r.traceRays(o2, d2, 0, want_color=False)
        for i in range(len(o2)):
            tt = C.c_double(0.0)
            want = L.orc_ref64_world_hit(arr2, n2, _dp(o2[i]), _dp(d2[i]), 0.001, math.inf, C.byref(tt))
            assert hid[i] == want and (want < 0 or t[i] == tt.value), (i, hid[i], want)
        assert 5 in hid and 4 not in hid and 2 in hid and 0 in hid    # of the coincident pair only the later entry is ever hit


def test_ray_color_equals_mirror_and_tracks_fp64(gpu_renderer, oracle, book_scene):
    arr, n = book_scene
    R = _R()
    gpu_renderer.setWorld(R.randomScene())
    L = oracle.lib()
    rng = np.random.default_rng(12)
    o, d = _rays(rng, 600)
    for depth in (1, 2, 50):
        col, hid, t = gpu_renderer.traceRays(o, d, depth, seed=0xE7)
        close = 0
        for i in range(len(o)):
            c = (C.c_double * 3)()
            tt = C.c_double(0.0)
            want = L.orc_mirror32_trace_ray(arr, n, _dp(o[i]), _dp(d[i]), 0.001, math.inf, depth, 0xE7, i, c, C.byref(tt))
            assert hid[i] == want and list(col[i]) == list(c), (depth, i, list(col[i]), list(c))
            f = (C.c_double * 3)()
            L.orc_ref64_ray_color(arr, n, _dp(o[i]), _dp(d[i]), depth, 0xE7, i, f)
            if depth == 1:
                assert list(col[i]) == list(f), (i, list(col[i]), list(f))   # black, or the binary64 sky colour
            close += max(abs(col[i][k] - f[k]) for k in range(3)) < 1e-4
        assert close >= 0.97 * len(o), (depth, close)   # same streams: the same path almost always, then ~1e-6 apart
    assert (col.max(axis=1) > 0).sum() > 300


def test_get_rays_equal_the_fp64_restatement(gpu_renderer, oracle):
    R = _R()
    L = oracle.lib()
    rng = np.random.default_rng(13)
    for aspect, kw in ((1.5, {}), (16 / 9, {"look_from": (3, 3, 2), "look_at": (0, 0, -1), "vfov": 20.0, "aperture": 2.0,
                                            "focus": math.sqrt(27.0)})):
        cam = oracle.book_camera(aspect, **kw)
        from raytracinginoneweekend_amd.basetypes import point, vec3
        from raytracinginoneweekend_amd.camera import initCamera
        hc = R.bookCamera(aspect) if not kw else initCamera(point(3, 3, 2), point(0, 0, -1), vec3(0, 1, 0), 20.0, aspect,
                                                              aperture=2.0, focusDist=math.sqrt(27.0))
        gpu_renderer.setCamera(hc)
        s, t = rng.uniform(-0.1, 1.1, 500), rng.uniform(-0.1, 1.1, 500)
        og, dg = gpu_renderer.getRays(s, t, seed=0xE7)
        for i in range(len(s)):
            key = (C.c_uint32 * 2)()
            L.orc_path_key(0xE7, i, 0, key)
            k = C.c_uint32(0)
            oo, dd = (C.c_double * 3)(), (C.c_double * 3)()
            L.orc_ref64_get_ray(C.byref(cam), s[i], t[i], key, C.byref(k), oo, dd)
            assert list(og[i]) == list(oo) and list(dg[i]) == list(dd), i
            assert k.value >= 2


def test_reference_signatures(oracle, book_scene):
    """rayColor(r, world, depth), hit(world, r, t_min, t_max, rec), getRay(camera, s, t) as the reference spells them."""
    arr, n = book_scene
    R = _R()
    from raytracinginoneweekend_amd import camera as cam_mod, hittables as HT
    from raytracinginoneweekend_amd.basetypes import Color, initRay, point, vec3
    from raytracinginoneweekend_amd.hitrecord import HitRecord
    L = oracle.lib()
    world = R.randomScene()
    cam = R.bookCamera(1.5)
    r = cam_mod.getRay(cam, 0.5, 0.45)
    rec = HitRecord()
    assert HT.hit(world, r, 0.001, math.inf, rec) is True
    o, d = np.array(list(r.orig)), np.array(list(r.dir))
    tt, pp, nn, ff = C.c_double(0), (C.c_double * 3)(), (C.c_double * 3)(), C.c_int(0)
    assert L.orc_ref64_sphere_hit(C.byref(arr[rec.index]), _dp(o), _dp(d), 0.001, math.inf, C.byref(tt), pp, nn, C.byref(ff)) == 1
    assert rec.t == tt.value and list(rec.p) == list(pp) and list(rec.normal) == list(nn) and rec.frontFace == bool(ff.value)
    assert rec.mat is world[rec.index].hSphere.mat
    up = initRay(point(0, 5, 0), vec3(0, 1, 0))
    assert HT.hit(world, up, 0.001, math.inf, HitRecord()) is False
    c = R.rayColor(up, world, 50)
    assert isinstance(c, Color) and list(c) == [0.5, 0.7, 1.0]            # straight up: t = 1 in raytracer.nim:18-20
    assert list(R.rayColor(up, world, 0)) == [0.0, 0.0, 0.0]              # depth <= 0 (raytracer.nim:8-9)
    down = initRay(point(0, 5, 0), vec3(0.3, -1, 0.2))
    f = (C.c_double * 3)()
    L.orc_ref64_ray_color(arr, n, _dp(np.array(list(down.orig))), _dp(np.array(list(down.dir))), 50, 0xE7, 3, f)
    got = R.rayColor(down, world, 50, index=3)
    assert max(abs(got[k] - f[k]) for k in range(3)) < 1e-4


def test_hit_overloads_for_one_object(oracle):
    """hit*(s: Sphere, ...) (hitrecord.nim:55-79) and hit*(h: Hittable, ...) (hittables.nim:75-77): the record of a
    one-object query equals the FP64 restatement's Sphere.hit - t, p, normal, frontFace - for rays from outside,
    from inside, with a negative radius, and for a restricted [t_min, t_max]."""
    from raytracinginoneweekend_amd import hittables as HT
    from raytracinginoneweekend_amd.basetypes import color, initRay, point, vec3
    from raytracinginoneweekend_amd.hitrecord import HitRecord, Sphere, initDielectric, initLambertian, initMaterial
    L = oracle.lib()
    rng = np.random.default_rng(21)
    cases = [(Sphere(point(0.3, 0.1, -2.0), 0.75, initMaterial(initLambertian(color(0.2, 0.3, 0.4)))), (0.001, math.inf)),
             (Sphere(point(0.0, 0.0, 0.0), 3.0, initMaterial(initDielectric(1.5))), (0.001, math.inf)),       # rays start inside
             (Sphere(point(-0.2, 0.0, -1.5), -0.6, initMaterial(initDielectric(1.5))), (0.001, math.inf)),    # normal flipped
             (Sphere(point(0.0, 0.0, -2.0), 0.9, initMaterial(initLambertian(color(0.5, 0.5, 0.5)))), (1.5, 4.0))]
    n_hits = 0
    for sphere, (t_min, t_max) in cases:
        arr, _ = oracle.spheres_from_arrays([list(sphere.center)], [sphere.radius], [sphere.mat.flat()[0]],
                                            [sphere.mat.flat()[1]], [sphere.mat.flat()[2]], [sphere.mat.flat()[3]])
        for i in range(40):
            o = rng.normal(0.0, 0.4, 3)
            d = (np.array([0.0, 0.0, -1.0]) + rng.normal(0.0, 0.5, 3)) * math.exp(rng.uniform(-2.0, 2.0))
            ray = initRay(point(*o), vec3(*d))
            tt, pp, nn, ff = C.c_double(0), (C.c_double * 3)(), (C.c_double * 3)(), C.c_int(0)
            want = L.orc_ref64_sphere_hit(arr, _dp(o), _dp(d), t_min, t_max, C.byref(tt), pp, nn, C.byref(ff))
            for obj in (sphere, HT.Hittable(HT.htSphere, sphere)):
                rec = HitRecord()
                assert HT.hit(obj, ray, t_min, t_max, rec) == bool(want), (i, want)
                if want:
                    assert rec.t == tt.value and list(rec.p) == list(pp) and list(rec.normal) == list(nn)
                    assert rec.frontFace == bool(ff.value) and rec.mat is sphere.mat
            n_hits += want
    assert n_hits > 60


def test_list_with_unknown_hittable_kind_is_refused(gpu_renderer, oracle, book_scene):
    """rtow_set_hittables takes the list as the reference holds it - Hittable.kind per entry (hittables.nim:4-8) -
    and has, like hit*(h: Hittable) (hittables.nim:75-77), one branch: htSphere.  Another kind is an error that
    names the entry, and the context keeps the world it had."""
    from raytracinginoneweekend_amd import _lib, hittables as HT
    R = _R()
    arr, n = book_scene
    world = R.randomScene()
    gpu_renderer.setWorld(world)        # through rtow_set_hittables: the flattened list carries the kinds
    gpu_renderer.setCamera(R.bookCamera(1.5))
    before = gpu_renderer.renderRows(60, 40, 2, 50, want_accum=True)
    ref = oracle.render_mirror32(arr, n, oracle.book_camera(1.5), 60, 40, 2, 50)
    assert (before[1] == ref[1]).all()
    broken = R.randomScene()
    broken[17] = HT.Hittable(1, broken[17].hSphere)
    with pytest.raises(_lib.RtowError) as e:
        gpu_renderer.setWorld(broken)
    assert e.value.status == -1 and "entry 17" in str(e.value) and "htSphere" in str(e.value)
    after = gpu_renderer.renderRows(60, 40, 2, 50, want_accum=True)
    assert (after[1] == before[1]).all()
    with pytest.raises(ValueError):      # the host-side dispatch refuses it as well
        HT.hit(broken[17], R.getRay(R.bookCamera(1.5), 0.5, 0.5), 0.001, math.inf, R.HitRecord())


def test_world_edited_in_place_is_seen_by_the_per_ray_procs():
    """The reference's procs take `var HittablesList` and see its current contents.  A sphere replaced, or edited
    in place, at the same list length must reach the device before the next query."""
    from raytracinginoneweekend_amd.basetypes import color, initRay, point, vec3
    from raytracinginoneweekend_amd.hitrecord import HitRecord, Sphere, initLambertian, initMaterial
    from raytracinginoneweekend_amd.hittables import initHittables
    R = _R()
    world = initHittables(0)
    world.add(Sphere(point(0, 0, -3), 0.5, initMaterial(initLambertian(color(0.5, 0.5, 0.5)))))
    world.add(Sphere(point(0, 0, -6), 0.5, initMaterial(initLambertian(color(0.5, 0.5, 0.5)))))
    ray = initRay(point(0, 0, 0), vec3(0, 0, -1))
    rec = HitRecord()
    assert R.hit(world, ray, 0.001, math.inf, rec) and rec.index == 0 and rec.t == 2.5
    world[0].hSphere.center = point(5, 0, -3)                    # edited in place: out of the ray's way
    assert R.hit(world, ray, 0.001, math.inf, rec) and rec.index == 1 and rec.t == 5.5
    world[1] = type(world[1])(0, Sphere(point(0, 0, -8), 1.0))   # replaced at the same length
    assert R.hit(world, ray, 0.001, math.inf, rec) and rec.index == 1 and rec.t == 7.0
    world[1].hSphere.radius = 2.0
    assert R.hit(world, ray, 0.001, math.inf, rec) and rec.t == 6.0
    assert list(R.rayColor(initRay(point(0, 0, 0), vec3(0, 1, 0)), world, 5)) == [0.5, 0.7, 1.0]


def test_directions_far_outside_binary32_range(gpu_renderer, oracle, book_scene):
    """hit* takes un-normalised binary64 directions (camera.nim:38-39 never normalises).  A direction whose
    length under- or overflows binary32 must still find the reference's hit: index equal, t equal (t is in units of
    |direction|, so it scales inversely)."""
    arr, n = book_scene
    R = _R()
    gpu_renderer.setWorld(R.randomScene())
    L = oracle.lib()
    rng = np.random.default_rng(31)
    o, d = _rays(rng, 400)
    _, hid0, _ = gpu_renderer.traceRays(o, d, 0, want_color=False)
    for scale in (1e-25, 1e25, 2.0 ** -70, 2.0 ** 90, 1e-150, 1e140):
        ds = d * scale
        _, hid, t = gpu_renderer.traceRays(o, ds, 0, t_min=0.001 / scale, want_color=False)
        hits = 0
        for i in range(len(o)):
            tt = C.c_double(0.0)
            want = L.orc_ref64_world_hit(arr, n, _dp(o[i]), _dp(np.ascontiguousarray(ds[i])), 0.001 / scale, math.inf, C.byref(tt))
            assert hid[i] == want, (scale, i, hid[i], want)
            if want >= 0:
                assert t[i] == tt.value, (scale, i)
                hits += 1
        assert hits > 150
        if scale in (2.0 ** -70, 2.0 ** 90):   # powers of two change no rounding anywhere: the same hits as unscaled
            assert (hid == hid0).all()
    # colours: the sky gradient of a miss uses the binary64 unit direction, whatever the length
    up = np.array([[0.0, 5.0, 0.0]])
    for scale in (1e-30, 1.0, 1e30):
        col, _, _ = gpu_renderer.traceRays(up, np.array([[0.0, scale, 0.0]]), 50)
        assert list(col[0]) == [0.5, 0.7, 1.0]


def test_cpp_host_per_ray_procs(oracle, book_scene):
    """getRay, hit, rayColor and scatter of the compiled host (raytracinginoneweekend_amd/host) through its CLI."""
    import os
    import subprocess
    from conftest import ROOT
    arr, n = book_scene
    L = oracle.lib()
    exe = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "rtow_raytracer")
    cam = oracle.book_camera(1.5)
    for s_, t_ in ((0.5, 0.45), (0.2, 0.9), (0.9, 0.2)):
        v = subprocess.check_output([exe, "--probe", repr(s_), repr(t_), "--depth", "50"], text=True).split()
        o, d = np.array([float(x) for x in v[0:3]]), np.array([float(x) for x in v[3:6]])
        key = (C.c_uint32 * 2)()
        L.orc_path_key(0xE7, 0, 0, key)
        k = C.c_uint32(0)
        oo, dd = (C.c_double * 3)(), (C.c_double * 3)()
        L.orc_ref64_get_ray(C.byref(cam), s_, t_, key, C.byref(k), oo, dd)
        assert list(o) == list(oo) and list(d) == list(dd)
        tt = C.c_double(0.0)
        want = L.orc_ref64_world_hit(arr, n, _dp(o), _dp(d), 0.001, math.inf, C.byref(tt))
        assert int(v[6]) == (want >= 0) and int(v[7]) == want
        if want >= 0:
            pp, nn, ff = (C.c_double * 3)(), (C.c_double * 3)(), C.c_int(0)
            assert L.orc_ref64_sphere_hit(C.byref(arr[want]), _dp(o), _dp(d), 0.001, math.inf, C.byref(tt), pp, nn, C.byref(ff)) == 1
            assert float(v[8]) == tt.value and [float(x) for x in v[9:12]] == list(pp)
            assert [float(x) for x in v[12:15]] == list(nn) and int(v[15]) == ff.value
        c = (C.c_double * 3)()
        L.orc_mirror32_trace_ray(arr, n, _dp(o), _dp(d), 0.001, math.inf, 50, 0xE7, 0, c, C.byref(tt))
        assert [float(x) for x in v[16:19]] == list(c)
        if want >= 0:   # scatter(rec.mat, r, rec) on that hit, stream of path (pixel 0, sample 0)
            a32, o32 = (C.c_double * 3)(), (C.c_double * 3)()
            nrm = np.array([float(x) for x in v[12:15]])
            ok32 = L.orc_mirror32_scatter(C.byref(arr[want]), _dp(d), _dp(nrm), int(v[15]), 0xE7, 0, a32, o32)
            assert int(v[19]) == ok32 and [float(x) for x in v[20:23]] == list(a32) and [float(x) for x in v[23:26]] == list(o32)


def test_cpp_host_one_object_hits_and_unknown_kind(oracle, book_scene):
    """The compiled host's hit(Sphere) (hitrecord.nim:55) and hit(Hittable) (hittables.nim:75-77) overloads against
    the FP64 restatement's Sphere.hit, and its upload of a list whose entry has a kind the reference lacks."""
    import os
    import subprocess
    from conftest import ROOT
    arr, n = book_scene
    L = oracle.lib()
    exe = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "rtow_raytracer")
    seen = 0
    # (s, t) of a camera ray and a list entry: the sphere it hits in the whole list, the ground it would hit behind
    # that sphere, the big metal sphere, the ground from a pixel that sees it, and two entries the ray misses
    for (s_, t_), j in (((0.5, 0.45), 298), ((0.5, 0.45), 0), ((0.52, 0.5), 483), ((0.5, 0.2), 0), ((0.2, 0.9), 0),
                        ((0.52, 0.5), 481)):
        lines = subprocess.check_output([exe, "--probe", repr(s_), repr(t_), "--probe-one", str(j), "--depth", "2"],
                                        text=True).strip().splitlines()
        assert len(lines) == 3
        v = lines[0].split()
        o, d = np.array([float(x) for x in v[0:3]]), np.array([float(x) for x in v[3:6]])
        tt, pp, nn, ff = C.c_double(0), (C.c_double * 3)(), (C.c_double * 3)(), C.c_int(0)
        want = L.orc_ref64_sphere_hit(C.byref(arr[j]), _dp(o), _dp(d), 0.001, math.inf, C.byref(tt), pp, nn, C.byref(ff))
        for line in lines[1:]:
            w = line.split()
            assert int(w[0]) == want, (j, line)
            if want:
                assert float(w[1]) == tt.value and [float(x) for x in w[2:5]] == list(pp)
                assert [float(x) for x in w[5:8]] == list(nn) and int(w[8]) == ff.value
        seen += want
    assert seen == 4
    bad = subprocess.run([exe, "--width", "60", "--spp", "1", "--corrupt-kind", "5", "--out", os.devnull], capture_output=True, text=True)
    assert bad.returncode == 1 and "entry 5" in bad.stderr and "htSphere" in bad.stderr


def test_ray_entry_point_errors(gpu_renderer):
    from raytracinginoneweekend_amd import _lib
    R = _R()
    one = np.zeros((1, 3))
    with pytest.raises(_lib.RtowError) as e:
        gpu_renderer.traceRays(one, one + 1, -1)
    assert e.value.status == -1
    with pytest.raises(_lib.RtowError):
        gpu_renderer.traceRays(one, one + 1, 1, t_min=2.0, t_max=1.0)
    col, hid, t = gpu_renderer.traceRays(np.zeros((0, 3)), np.zeros((0, 3)), 5)
    assert col.shape == (0, 3) and hid.shape == (0,)
    with R.Renderer(0) as fresh:
        with pytest.raises(_lib.RtowError) as e:
            fresh.traceRays(one, one + 1, 1)
        assert e.value.status == -3
        with pytest.raises(_lib.RtowError) as e:
            fresh.getRays([0.5], [0.5])
        assert e.value.status == -3


def _scatter_cases(rng, n):
    """Random (material, incoming ray, hit record) triples: unit normals turned against the rays, directions of
    any length, all three material kinds, grazing and head-on incidence."""
    nrm = rng.normal(0.0, 1.0, (n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    d = rng.normal(0.0, 1.0, (n, 3))
    flip = np.einsum("ij,ij->i", d, nrm) > 0
    d[flip] = -d[flip]                                   # rec.normal always faces the ray (hitrecord.nim:51-53)
    d *= np.exp(rng.uniform(-3.0, 3.0, (n, 1)))
    kind = rng.integers(0, 3, n).astype(np.uint8)
    albedo = rng.uniform(0.0, 1.0, (n, 3))
    fuzz = np.where(rng.uniform(size=n) < 0.3, 0.0, rng.uniform(0.0, 1.0, n))
    ir = np.where(rng.uniform(size=n) < 0.5, 1.5, rng.uniform(1.05, 2.4, n))
    front = rng.integers(0, 2, n).astype(np.uint8)
    return d, nrm, front, kind, albedo, fuzz, ir


def test_scatter_against_mirror_and_fp64_restatement(gpu_renderer, oracle):
    """scatter (hitrecord.nim:127-132) through rtow_scatter_rays: attenuation, scattered direction and result
    equal the renderer's CPU mirror bit for bit, and the FP64 restatement of the reference closely (same draws:
    a decision that differs needs a value within rounding of its threshold)."""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    n = 4000
    d, nrm, front, kind, albedo, fuzz, ir = _scatter_cases(rng, n)
    seed = 0xE7
    att, out, ok = gpu_renderer.scatterRays(d, nrm, front, kind, albedo, fuzz, ir, seed=seed)
    assert att.shape == out.shape == (n, 3) and ok.shape == (n,)
    key = (C.c_uint32 * 2)()
    zero = (C.c_double * 3)(0.0, 0.0, 0.0)
    differing_decisions = 0
    absorbed = 0
    for i in range(n):
        arr, _ = oracle.spheres_from_arrays([(0, 0, 0)], [1.0], [kind[i]], [albedo[i]], [fuzz[i]], [ir[i]])
        a32, o32 = (C.c_double * 3)(), (C.c_double * 3)()
        ok32 = L.orc_mirror32_scatter(arr, _dp(d[i]), _dp(nrm[i]), int(front[i]), seed, i, a32, o32)
        assert bool(ok32) == bool(ok[i]), i
        assert list(a32) == list(att[i]) and list(o32) == list(out[i]), (i, kind[i], list(o32), list(out[i]))
        absorbed += not ok32
        L.orc_path_key(seed, i, 0, key)
        k = C.c_uint32(0)
        a64, oo, o64 = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
        ok64 = L.orc_ref64_scatter(arr, zero, _dp(d[i]), zero, _dp(nrm[i]), int(front[i]), key, C.byref(k), a64, oo, o64)
        want = np.array(list(o64))
        if bool(ok64) != bool(ok[i]) or not np.allclose(out[i], want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max())):
            differing_decisions += 1                      # reflect / refract, or a near-zero test, decided the other way
            continue
        if kind[i] == 2:
            assert list(a64) == [1.0, 1.0, 1.0] == list(att[i])
        else:
            assert np.allclose(att[i], list(a64), rtol=1e-6, atol=0)
    assert absorbed > 20                                   # fuzzed metal reflections below the surface occur
    assert differing_decisions <= n // 500
    assert gpu_renderer.scatterRays(d[:0], nrm[:0], front[:0], kind[:0], albedo[:0], fuzz[:0], ir[:0])[2].shape == (0,)


def test_scatter_wrapper_and_errors(gpu_renderer, oracle):
    """The host proc with the reference's parameter list, chained after hit: scatter(rec.mat, r, rec)."""
    R = _R()
    from raytracinginoneweekend_amd import _lib, basetypes as B, hitrecord as HR
    world = R.randomScene()
    r = B.initRay(B.point(13, 2, 3), B.vec3(-13, -1.5, -3))
    rec = HR.HitRecord()
    assert R.hit(world, r, 0.001, math.inf, rec)
    ok, att, sc = R.scatter(rec.mat, r, rec, index=3)
    arr, _ = oracle.spheres_from_arrays([(0, 0, 0)], [1.0], *[[x] for x in rec.mat.flat()])
    a32, o32 = (C.c_double * 3)(), (C.c_double * 3)()
    d = np.asarray(list(r.dir), np.float64)
    nn = np.asarray(list(rec.normal), np.float64)
    ok32 = oracle.lib().orc_mirror32_scatter(arr, _dp(d), _dp(nn), int(rec.frontFace), R.RENDER_SEED, 3, a32, o32)
    assert ok == bool(ok32) and list(att) == list(a32) and list(sc.dir) == list(o32) and list(sc.orig) == list(rec.p)
    with pytest.raises(_lib.RtowError) as e:
        gpu_renderer.scatterRays([[0, 0, -1]], [[0, 0, 1]], [1], [7], [[1, 1, 1]], [0.0], [1.5])
    assert e.value.status == -1 and "kind" in str(e.value)
    with pytest.raises(_lib.RtowError):
        gpu_renderer.scatterRays([[0, 0, -1]], [[0, 0, 1]], [1], [2], [[1, 1, 1]], [0.0], [0.0])
