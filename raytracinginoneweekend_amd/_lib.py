"""ctypes binding of librtow_hip.so (the C ABI of include/rtow.h).

Loading fails loudly when the library has not been built: nothing in this package can render
without it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTOW_LIB_PATH selects an alternative build of the same ABI (instrumented builds for profiling)
LIB_PATH = os.environ.get("RTOW_LIB_PATH") or os.path.join(_HERE, "lib", "librtow_hip.so")

LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
HT_SPHERE = 0                                   # HittableKind (hittables.nim:4-5)
GATHER_NONE, GATHER_HOST, GATHER_RCCL, GATHER_HOST_AFTER_RCCL = 0, 1, 2, 3   # rtow_stats.gather_path
SCREEN_NONE, SCREEN_SLAB, SCREEN_INTERVAL, SCREEN_BUNDLE, SCREEN_MASK = 0, 1, 2, 3, 4          # rtow_stats.camera_screen


class RtowError(RuntimeError):
    """A call into librtow_hip.so returned a negative status."""

    def __init__(self, status, message):
        super().__init__(f"rtow status {status}: {message}")
        self.status = status


class CameraStruct(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float)]


class CameraStruct64(C.Structure):
    """rtow_camera_f64: the fields as initCamera computes them, unrounded."""
    _fields_ = [("origin", C.c_double * 3), ("lower_left", C.c_double * 3),
                ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3),
                ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
                ("lens_radius", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("ray_segments", C.c_uint64), ("paths", C.c_uint64), ("trace_ms", C.c_float),
                ("total_ms", C.c_float), ("n_spheres", C.c_int32), ("n_big", C.c_int32),
                ("shader_clock_mhz", C.c_float), ("gather_path", C.c_int32), ("camera_screen", C.c_int32),
                ("reserved_", C.c_int32)]


# every symbol include/rtow.h declares
SYMBOLS = ["rtow_last_error", "rtow_version", "rtow_device_count", "rtow_create", "rtow_destroy",
           "rtow_set_scene", "rtow_set_scene_f64", "rtow_set_hittables", "rtow_set_camera", "rtow_set_camera_f64", "rtow_render", "rtow_render_async", "rtow_sync",
           "rtow_render_multi", "rtow_multi_plan", "rtow_interval_plan", "rtow_mask_plan", "rtow_trace_rays", "rtow_get_rays", "rtow_scatter_rays", "rtow_render_pass", "rtow_render_noalloc", "rtow_read_image", "rtow_format_ppm", "rtow_selftest_arithmetic", "rtow_get_stats", "rtow_get_stream"]

_lib = None


def _preload_hip_runtime():
    """Make this process use ONE HIP runtime.

    PyTorch wheels carry their own libamdhip64.so (same SONAME as /opt/rocm's).  If librtow_hip.so
    pulled in the system copy first and torch then loaded its own, the second runtime would find
    no GPU.  Loading torch's copy up front (when torch is installed) lets both resolve to it,
    whichever is imported first; without torch the system runtime is used.
    """
    if os.environ.get("RTOW_SYSTEM_HIP_RUNTIME") == "1":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    # the same for RCCL, which rtow_render_multi opens on first use (never at import): name torch's copy
    rccl = os.path.join(libdir, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("RTOW_RCCL_LIB", rccl)
    path = os.path.join(libdir, "libamdhip64.so")
    if not os.path.exists(path):
        return None
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


class _Tolerant:
    """Prototype declarations for builds that predate some entry points (A/B timing of old builds):
    attribute writes to symbols the build lacks are dropped."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        try:
            return getattr(self._cdll, name)
        except AttributeError:
            class _Missing:
                pass
            return _Missing()


def load(path, strict=True):
    """dlopen one build of the library and declare its prototypes.  strict=False tolerates a build
    that lacks newer entry points (only for timing old builds against new ones)."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `make -C raytracinginoneweekend_amd/csrc` "
            "(or __graft_entry__.build()); this package has no CPU fallback")
    _preload_hip_runtime()
    cdll = C.CDLL(path)
    L = cdll if strict else _Tolerant(cdll)
    vp = C.c_void_p
    L.rtow_last_error.restype = C.c_char_p
    L.rtow_last_error.argtypes = []
    L.rtow_version.argtypes = []
    L.rtow_device_count.argtypes = [C.POINTER(C.c_int)]
    L.rtow_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rtow_destroy.argtypes = [vp]
    L.rtow_set_scene.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rtow_set_scene_f64.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rtow_set_hittables.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rtow_multi_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.POINTER(C.c_uint64)]
    if hasattr(L, "rtow_interval_plan"):
        L.rtow_interval_plan.argtypes = [C.POINTER(CameraStruct64), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int)]
    if hasattr(L, "rtow_mask_plan"):
        L.rtow_mask_plan.argtypes = [C.POINTER(CameraStruct64), C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rtow_set_camera.argtypes = [vp, C.POINTER(CameraStruct)]
    L.rtow_set_camera_f64.argtypes = [vp, C.POINTER(CameraStruct64)]
    L.rtow_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, vp, C.c_int, vp, vp, vp]
    L.rtow_render_async.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, vp, C.c_int, vp]
    L.rtow_sync.argtypes = [vp]
    L.rtow_render_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, vp,
                                    C.POINTER(C.c_uint64)]
    L.rtow_render_pass.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, vp, C.c_int, vp]
    L.rtow_render_noalloc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, vp, C.c_int, vp]
    L.rtow_trace_rays.argtypes = [vp, C.c_int, vp, vp, C.c_double, C.c_double, C.c_int, C.c_uint64, vp, vp, vp]
    L.rtow_get_rays.argtypes = [vp, C.c_int, vp, vp, C.c_uint64, vp, vp]
    L.rtow_scatter_rays.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]
    L.rtow_read_image.argtypes = [vp, vp, vp]
    L.rtow_format_ppm.argtypes = [vp, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rtow_selftest_arithmetic.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rtow_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rtow_get_stream.argtypes = [vp, C.POINTER(vp)]
    return cdll


def lib():
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


def check(status, library=None):
    """Raise RtowError for a negative status.  `library` is the build whose call returned it: the
    message is thread-local state of THAT shared object (a Renderer may run on an alternative build
    loaded with load(path))."""
    if status != 0:
        L = library if library is not None else lib()
        raise RtowError(status, L.rtow_last_error().decode("utf-8", "replace"))
