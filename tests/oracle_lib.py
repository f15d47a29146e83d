"""ctypes binding of the CPU oracle (oracle/_build/librtow_oracle.so).

Test infrastructure only: the renderer package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "librtow_oracle.so")

LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
RNG_COUNTER, RNG_NIM_SEQUENTIAL = 0, 1


class Sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("kind", C.c_int32),
                ("pad_", C.c_int32), ("albedo", C.c_double * 3), ("fuzz", C.c_double),
                ("ir", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left", C.c_double * 3),
                ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3),
                ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
                ("lens_radius", C.c_double)]


class NimRand(C.Structure):
    _fields_ = [("a0", C.c_uint64), ("a1", C.c_uint64)]


def build(force=False):
    """Compile the oracle with gcc if the library is missing or stale."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("rng.c", "ref64.c", "mirror32.c", "rtow_oracle.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.exists(s))
    if force or stale:
        if not all(os.path.exists(s) for s in srcs):
            raise RuntimeError("oracle sources missing and no prebuilt library")
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    u32p = C.POINTER(C.c_uint32)
    dp = C.POINTER(C.c_double)
    L.orc_nimrand_default.argtypes = [C.POINTER(NimRand)]
    L.orc_nimrand_init.argtypes = [C.POINTER(NimRand), C.c_int64, C.c_int]
    L.orc_nimrand_next.argtypes = [C.POINTER(NimRand)]
    L.orc_nimrand_next.restype = C.c_uint64
    L.orc_nimrand_rand.argtypes = [C.POINTER(NimRand), C.c_double]
    L.orc_nimrand_rand.restype = C.c_double
    L.orc_path_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, u32p]
    L.orc_draw_u32.argtypes = [u32p, C.c_uint32]
    L.orc_draw_u32.restype = C.c_uint32
    L.orc_draw_unit.argtypes = [u32p, C.c_uint32]
    L.orc_draw_unit.restype = C.c_double
    L.orc_scene_red_blue.argtypes = [C.POINTER(Sphere), C.c_int]
    L.orc_scene_mix_of_spheres.argtypes = [C.POINTER(Sphere), C.c_int]
    L.orc_scene_random.argtypes = [C.POINTER(NimRand), C.c_int, C.c_double, C.c_double,
                                   C.POINTER(Sphere), C.c_int, C.POINTER(C.c_uint64)]
    L.orc_init_camera.argtypes = [dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.POINTER(Camera)]
    L.orc_init_camera.restype = None
    L.orc_ref64_render.argtypes = [C.POINTER(Sphere), C.c_int, C.POINTER(Camera), C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(NimRand),
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_uint64)]
    L.orc_mirror32_render.argtypes = [C.POINTER(Sphere), C.c_int, C.POINTER(Camera), C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.orc_mirror32_filter_audit.argtypes = [C.c_int]
    L.orc_mirror32_filter_audit.restype = None
    L.orc_mirror32_filter_audit_result.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.orc_mirror32_filter_audit_result.restype = None
    L.orc_ref64_sphere_hit.argtypes = [C.POINTER(Sphere), dp, dp, C.c_double, C.c_double, dp, dp, dp,
                                       C.POINTER(C.c_int)]
    L.orc_ref64_world_hit.argtypes = [C.POINTER(Sphere), C.c_int, dp, dp, C.c_double, C.c_double, dp]
    L.orc_ref64_scatter.argtypes = [C.POINTER(Sphere), dp, dp, dp, dp, C.c_int, u32p, u32p, dp, dp, dp]
    L.orc_ref64_get_ray.argtypes = [C.POINTER(Camera), C.c_double, C.c_double, u32p, u32p, dp, dp]
    L.orc_ref64_get_ray.restype = None
    L.orc_ref64_quantise.argtypes = [C.c_double, C.c_int]
    L.orc_ref64_ray_color.argtypes = [C.POINTER(Sphere), C.c_int, dp, dp, C.c_int, C.c_uint64, C.c_uint32, dp]
    L.orc_ref64_ray_color.restype = None
    L.orc_mirror32_trace_ray.argtypes = [C.POINTER(Sphere), C.c_int, dp, dp, C.c_double, C.c_double, C.c_int, C.c_uint64,
                                         C.c_uint32, dp, dp]
    L.orc_mirror32_scatter.argtypes = [C.POINTER(Sphere), dp, dp, C.c_int, C.c_uint64, C.c_uint32, dp, dp]
    L.orc_ref64_set_variant.argtypes = [C.c_int]
    L.orc_ref64_set_variant.restype = None
    _lib = L
    return L


def vec3(x, y, z):
    return (C.c_double * 3)(x, y, z)


def nim_default():
    r = NimRand()
    lib().orc_nimrand_default(C.byref(r))
    return r


def random_scene(half_extent=11, lambert_below=0.8, metal_below=0.95, rng=None):
    """randomScene() of the reference from the un-seeded Nim generator state."""
    rng = rng or nim_default()
    cap = 1 + (2 * half_extent) ** 2 + 3
    arr = (Sphere * cap)()
    draws = C.c_uint64(0)
    n = lib().orc_scene_random(C.byref(rng), half_extent, lambert_below, metal_below, arr, cap,
                               C.byref(draws))
    return arr, n, draws.value, rng


def mix_of_spheres():
    arr = (Sphere * 8)()
    return arr, lib().orc_scene_mix_of_spheres(arr, 8)


def red_blue():
    arr = (Sphere * 8)()
    return arr, lib().orc_scene_red_blue(arr, 8)


def spheres_from_arrays(center, radius, kind, albedo, fuzz, ir):
    n = len(radius)
    arr = (Sphere * max(n, 1))()
    for j in range(n):
        for c in range(3):
            arr[j].center[c] = float(center[j][c])
            arr[j].albedo[c] = float(albedo[j][c])
        arr[j].radius = float(radius[j])
        arr[j].kind = int(kind[j])
        arr[j].fuzz = float(fuzz[j])
        arr[j].ir = float(ir[j])
    return arr, n


def book_camera(aspect, look_from=(13, 2, 3), look_at=(0, 0, 0), vup=(0, 1, 0), vfov=20.0,
                aperture=0.1, focus=10.0):
    cam = Camera()
    lib().orc_init_camera(vec3(*look_from), vec3(*look_at), vec3(*vup), vfov, aspect, aperture, focus,
                          C.byref(cam))
    return cam


def _rows(rows, H):
    if rows is None:
        return None, H, None
    r = np.ascontiguousarray(rows, dtype=np.int32)
    return r, len(r), r.ctypes.data


def render_ref64(spheres, n, cam, W, H, spp, depth, seed=0xE7, rows=None, threads=8,
                 rng_mode=RNG_COUNTER, nim_state=None):
    keep, n_rows, rows_p = _rows(rows, H)
    rgb = np.zeros((n_rows, W, 3), np.uint8)
    acc = np.zeros((n_rows, W, 3), np.float64)
    hid = np.zeros((n_rows, W), np.int32)
    seg = C.c_uint64(0)
    rc = lib().orc_ref64_render(spheres, n, C.byref(cam), W, H, spp, depth, seed, rng_mode,
                                C.byref(nim_state) if nim_state is not None else None,
                                rows_p, n_rows, threads, rgb.ctypes.data, acc.ctypes.data,
                                hid.ctypes.data, C.byref(seg))
    assert rc == 0
    return rgb, acc, hid, seg.value


def render_mirror32(spheres, n, cam, W, H, spp, depth, seed=0xE7, rows=None, threads=8):
    keep, n_rows, rows_p = _rows(rows, H)
    rgb = np.zeros((n_rows, W, 3), np.uint8)
    acc = np.zeros((n_rows, W, 3), np.uint64)
    hid = np.zeros((n_rows, W), np.int32)
    seg = C.c_uint64(0)
    rc = lib().orc_mirror32_render(spheres, n, C.byref(cam), W, H, spp, depth, seed, rows_p, n_rows,
                                   threads, rgb.ctypes.data, acc.ctypes.data, hid.ctypes.data,
                                   C.byref(seg))
    assert rc == 0
    return rgb, acc, hid, seg.value


def render_mirror32_audited(*args, **kw):
    """render_mirror32 with the sphere screen switched to audit mode (it rejects nothing and is
    checked against the exact discriminant instead).  Returns (render result, spheres with
    disc >= 0, how many of them the screen would have dropped, smallest slack in margins)."""
    lib().orc_mirror32_filter_audit(1)
    try:
        out = render_mirror32(*args, **kw)
        hits, dropped, slack = C.c_uint64(0), C.c_uint64(0), C.c_double(0.0)
        lib().orc_mirror32_filter_audit_result(C.byref(hits), C.byref(dropped), C.byref(slack))
    finally:
        lib().orc_mirror32_filter_audit(0)
    return out, hits.value, dropped.value, slack.value
