"""Driver with the reference's interface (raytracer.nim): render, writeColor, the three scene
builders and main.  `render` keeps the reference's signature; its triple loop
(raytracer.nim:36-45) is one call into librtow_hip.so.
"""
import ctypes as C
import math
import sys

import numpy as np

from . import _lib, nimrandom
from .basetypes import Color, Image, Point, Ray, Vec3, color, length, point, vec3
from .camera import Camera, initCamera
from .hitrecord import (HitRecord, Sphere, initDielectric, initLambertian, initMaterial, initMetal)  # noqa: F401
from .hittables import HittablesList, initHittables

RENDER_SEED = 0xE7  # raytracer.nim:135


class Renderer:
    """One rtow_ctx: a GPU, its stream and its workspace.

    `setWorld` / `setCamera` upload once; `render*` calls can then be repeated without touching
    host memory again (the benchmark and the multi-GPU driver rely on that).
    """

    def __init__(self, device=0, lib=None):
        self._lib = lib if lib is not None else _lib.lib()
        h = C.c_void_p()
        self._check(self._lib.rtow_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._last_shape = None   # (n_rows, width) of the last render enqueued through this object

    def _check(self, status):
        _lib.check(status, self._lib)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rtow_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def setWorld(self, world, binary32=False):
        """Upload a HittablesList (or its flattened dict of arrays).  The values go to the library in
        binary64 (rtow_set_scene_f64); binary32=True rounds them first and uses rtow_set_scene."""
        if isinstance(world, HittablesList):
            world = world.flatten()
        ft = np.float32 if binary32 else np.float64
        fa = lambda a: np.ascontiguousarray(a, dtype=ft)  # noqa: E731
        cx, cy, cz, rad = fa(world["center_x"]), fa(world["center_y"]), fa(world["center_z"]), fa(world["radius"])
        kind = np.ascontiguousarray(world["kind"], dtype=np.uint8)
        albedo, fuzz, ir = fa(world["albedo"]), fa(world["fuzz"]), fa(world["ir"])
        n = len(rad)
        if not (len(cx) == len(cy) == len(cz) == len(kind) == len(fuzz) == len(ir) == n and albedo.shape == (n, 3)):
            raise ValueError("setWorld: arrays of unequal length")
        p = lambda a: a.ctypes.data if n else None  # noqa: E731
        hk = world.get("hittable_kind")
        if hk is not None and not binary32:
            # the list as the reference holds it, Hittable.kind included (hittables.nim:4-8): the library refuses any
            # kind but htSphere
            hk = np.ascontiguousarray(hk, dtype=np.uint8)
            if len(hk) != n:
                raise ValueError("setWorld: arrays of unequal length")
            self._check(self._lib.rtow_set_hittables(self._h, n, p(hk), p(cx), p(cy), p(cz), p(rad), p(kind), p(albedo),
                                                     p(fuzz), p(ir)))
        else:
            call = self._lib.rtow_set_scene if binary32 else self._lib.rtow_set_scene_f64
            self._check(call(self._h, n, p(cx), p(cy), p(cz), p(rad), p(kind), p(albedo), p(fuzz), p(ir)))
        self.n_spheres = n

    def setCamera(self, camera):
        """Upload a Camera (binary64 fields) or a ready _lib.CameraStruct / CameraStruct64."""
        s = camera.to_struct() if isinstance(camera, Camera) else camera
        if isinstance(s, _lib.CameraStruct):
            self._check(self._lib.rtow_set_camera(self._h, C.byref(s)))
        else:
            self._check(self._lib.rtow_set_camera_f64(self._h, C.byref(s)))

    @staticmethod
    def _rows(rows, height):
        if rows is None:
            return None, height, None
        r = np.ascontiguousarray(rows, dtype=np.int32)
        if len(r) == 0:   # an empty row list is legal; the ABI wants a non-NULL pointer to tell it from "all rows"
            r = np.zeros(1, np.int32)
            return r, 0, r.ctypes.data
        return r, len(r), r.ctypes.data

    def renderRows(self, width, height, samplesPerPixel, maxDepth, seed=RENDER_SEED, rows=None,
                   want_accum=False, want_hit_id=False):
        """Blocking render into host arrays: (rgb uint8 [n_rows,w,3], accum|None, hit_id|None)."""
        keep, n_rows, rows_p = self._rows(rows, height)
        rgb = np.empty((n_rows, width, 3), np.uint8)
        acc = np.empty((n_rows, width, 3), np.uint64) if want_accum else None
        hid = np.empty((n_rows, width), np.int32) if want_hit_id else None
        self._check(self._lib.rtow_render(
            self._h, width, height, samplesPerPixel, maxDepth, seed, rows_p, n_rows,
            rgb.ctypes.data if rgb.size else None,
            acc.ctypes.data if acc is not None and acc.size else None,
            hid.ctypes.data if hid is not None and hid.size else None))
        self._last_shape = (n_rows, width)
        return rgb, acc, hid

    def renderAsync(self, width, height, samplesPerPixel, maxDepth, seed=RENDER_SEED, rows=None, rgb_dev=None):
        """Enqueue a render whose 8-bit image stays on the device (rgb_dev: device address or None)."""
        keep, n_rows, rows_p = self._rows(rows, height)
        self._check(self._lib.rtow_render_async(self._h, width, height, samplesPerPixel, maxDepth, seed,
                                               rows_p, n_rows, rgb_dev))
        self._last_shape = (n_rows, width)

    def traceRays(self, origins, directions, maxDepth, t_min=0.001, t_max=math.inf, seed=RENDER_SEED,
                  want_color=True, want_hit=True):
        """rayColor / hit for a batch of rays (rtow_trace_rays): origins, directions float64 [n, 3].
        Returns (color float64 [n, 3] | None, hit_id int32 [n] | None, t float64 [n] | None); ray i scatters
        with the stream of path (pixel i, sample 0).  maxDepth = 0 answers the hit query only."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        if o.shape != d.shape:
            raise ValueError("traceRays: origins and directions differ in shape")
        n = o.shape[0]
        col = np.zeros((n, 3), np.float64) if want_color else None
        hid = np.full(n, -1, np.int32) if want_hit else None
        t = np.full(n, np.inf, np.float64) if want_hit else None
        ptr = lambda a: a.ctypes.data if a is not None and a.size else None  # noqa: E731
        self._check(self._lib.rtow_trace_rays(self._h, n, ptr(o), ptr(d), float(t_min), float(t_max), int(maxDepth), seed,
                                              ptr(col), ptr(hid), ptr(t)))
        return col, hid, t

    def getRays(self, s, t, seed=RENDER_SEED):
        """getRay for a batch of (s, t) (rtow_get_rays): returns (origins, directions) float64 [n, 3]."""
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(-1)
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
        if s.shape != t.shape:
            raise ValueError("getRays: s and t differ in length")
        n = s.shape[0]
        o, d = np.zeros((n, 3), np.float64), np.zeros((n, 3), np.float64)
        ptr = lambda a: a.ctypes.data if a.size else None  # noqa: E731
        self._check(self._lib.rtow_get_rays(self._h, n, ptr(s), ptr(t), seed, ptr(o), ptr(d)))
        return o, d

    def scatterRays(self, directions, normals, front_face, kind, albedo, fuzz, ir, seed=RENDER_SEED):
        """scatter for a batch of (material, incoming ray, hit record) triples (rtow_scatter_rays):
        directions, normals, albedo float64 [n, 3]; front_face, kind [n]; fuzz, ir float64 [n].  Returns
        (attenuation [n, 3], scattered direction [n, 3], scattered bool [n]); triple i draws from the stream
        of path (pixel i, sample 0)."""
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        nn = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        al = np.ascontiguousarray(albedo, dtype=np.float64).reshape(-1, 3)
        ff = np.ascontiguousarray(front_face, dtype=np.uint8).reshape(-1)
        kk = np.ascontiguousarray(kind, dtype=np.uint8).reshape(-1)
        fz = np.ascontiguousarray(fuzz, dtype=np.float64).reshape(-1)
        ii = np.ascontiguousarray(ir, dtype=np.float64).reshape(-1)
        n = d.shape[0]
        if not (nn.shape[0] == al.shape[0] == ff.shape[0] == kk.shape[0] == fz.shape[0] == ii.shape[0] == n):
            raise ValueError("scatterRays: the arrays differ in length")
        att, out = np.zeros((n, 3), np.float64), np.zeros((n, 3), np.float64)
        ok = np.zeros(n, np.uint8)
        ptr = lambda a: a.ctypes.data if a.size else None  # noqa: E731
        self._check(self._lib.rtow_scatter_rays(self._h, n, ptr(d), ptr(nn), ptr(ff), ptr(kk), ptr(al), ptr(fz), ptr(ii),
                                                seed, ptr(att), ptr(out), ptr(ok)))
        return att, out, ok.astype(bool)

    def renderNoAlloc(self, width, height, samplesPerPixel, maxDepth, rgb_dev, seed=RENDER_SEED, rows_dev=None,
                      n_rows=None):
        """Enqueue a render that allocates nothing (rtow_render_noalloc): rgb_dev and rows_dev are
        device addresses; at most 1 048 576 pixels; the sums live in per-GPU memory shared by all
        contexts on that GPU, so such renders must not overlap on one GPU."""
        self._check(self._lib.rtow_render_noalloc(self._h, width, height, samplesPerPixel, maxDepth, seed, rows_dev,
                                                  height if n_rows is None else n_rows, rgb_dev))

    def renderPass(self, width, height, firstSample, nSamples, maxDepth, seed=RENDER_SEED, rows=None, rgb_dev=None):
        """Progressive refinement: add samples firstSample .. firstSample+nSamples-1 of every pixel to the
        image this context holds (firstSample = 0 starts one).  Asynchronous."""
        keep, n_rows, rows_p = self._rows(rows, height)
        self._check(self._lib.rtow_render_pass(self._h, width, height, firstSample, nSamples, maxDepth, seed,
                                              rows_p, n_rows, rgb_dev))
        self._last_shape = (n_rows, width)

    def readImage(self, want_accum=False):
        """(rgb uint8 [n_rows, w, 3], accum uint64 | None) of the last render enqueued on this context
        whose image stayed in it (renderRows, renderAsync / renderPass without rgb_dev)."""
        if self._last_shape is None:
            raise RuntimeError("readImage: nothing has been rendered through this Renderer")
        n_rows, width = self._last_shape
        rgb = np.empty((n_rows, width, 3), np.uint8)
        acc = np.empty((n_rows, width, 3), np.uint64) if want_accum else None
        self._check(self._lib.rtow_read_image(self._h, rgb.ctypes.data if rgb.size else None,
                                             acc.ctypes.data if acc is not None and acc.size else None))
        return rgb, acc

    def sync(self):
        self._check(self._lib.rtow_sync(self._h))

    def ppm(self, binary=False):
        """The PPM file of the last render as bytes: P3 text exactly as the reference prints it
        (raytracer.nim:28,35), composed on the device; binary=True gives P6."""
        n = C.c_uint64(0)
        self._check(self._lib.rtow_format_ppm(self._h, int(binary), None, 0, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        self._check(self._lib.rtow_format_ppm(self._h, int(binary), buf.ctypes.data, n.value, C.byref(n)))
        return buf.tobytes()

    def stats(self):
        s = _lib.Stats()
        self._check(self._lib.rtow_get_stats(self._h, C.byref(s)))
        return {"ray_segments": s.ray_segments, "paths": s.paths, "trace_ms": s.trace_ms,
                "total_ms": s.total_ms, "n_spheres": s.n_spheres, "n_big": s.n_big,
                "shader_clock_mhz": s.shader_clock_mhz, "gather_path": s.gather_path,
                "camera_screen": s.camera_screen}

    def stream(self):
        p = C.c_void_p()
        self._check(self._lib.rtow_get_stream(self._h, C.byref(p)))
        return p.value


def intervalPlan(camera, world, lib=None):
    """rtow_interval_plan: per sphere of `world` the ranges of getRay's arguments s and t in which a camera ray can meet
    it, as (mid_s, half_s, mid_t, half_t) float32 arrays, or None when the camera is not one the construction covers.
    Plain host arithmetic: needs no GPU."""
    L = lib or _lib.lib()
    flat = world.flatten() if hasattr(world, "flatten") else world
    cx, cy, cz, rad = (np.ascontiguousarray(flat[k], np.float64) for k in ("center_x", "center_y", "center_z", "radius"))
    n = len(rad)
    out = [np.zeros(n, np.float32) for _ in range(4)]
    covered = C.c_int(0)
    cam = camera.to_struct() if isinstance(camera, Camera) else camera
    _lib.check(L.rtow_interval_plan(C.byref(cam), n, cx.ctypes.data, cy.ctypes.data, cz.ctypes.data, rad.ctypes.data,
                                    out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data, C.byref(covered)))
    return tuple(out) if covered.value else None


def maskPlan(camera, world, lib=None):
    """rtow_mask_plan: the tile grid and, per group of four spheres of `world`, the word of tiles in which a camera ray can
    meet one of them, as (group_mask uint32 [(n + 3) // 4], grid float32 [6], columns), or None when the camera is not
    one the construction covers.  tileOf(grid, columns, s, t) gives the bit of a ray.  Plain host arithmetic: needs no GPU."""
    L = lib or _lib.lib()
    flat = world.flatten() if hasattr(world, "flatten") else world
    cx, cy, cz, rad = (np.ascontiguousarray(flat[k], np.float64) for k in ("center_x", "center_y", "center_z", "radius"))
    n = len(rad)
    masks = np.zeros((n + 3) // 4, np.uint32)
    grid = np.zeros(6, np.float32)
    columns, covered = C.c_int(0), C.c_int(0)
    cam = camera.to_struct() if isinstance(camera, Camera) else camera
    _lib.check(L.rtow_mask_plan(C.byref(cam), n, cx.ctypes.data, cy.ctypes.data, cz.ctypes.data, rad.ctypes.data,
                                masks.ctypes.data, grid.ctypes.data, C.byref(columns), C.byref(covered)))
    return (masks, grid, columns.value) if covered.value else None


def tileOf(grid, columns, s, t):
    """Bit index of the image tile a ray with image coordinates (s, t) aims at (arrays allowed), in the renderer's
    binary32 arithmetic: one fused multiply-add, a clamp to [0, last], a truncation, per axis.  The fused
    multiply-add is formed in binary64 and rounded to binary32 (the product of two binary32 values is exact there, the
    sum is rounded twice), which can differ from the renderer's single rounding only for a coordinate within about
    2^-29 of a tile edge: good for tests and tools, not a bit-for-bit statement."""
    def index(v, k, c, last):
        v = np.asarray(v, np.float32)
        x = (v.astype(np.float64) * np.float64(k) + np.float64(c))
        x32 = x.astype(np.float32)
        x32 = np.where(x32 > 0, np.minimum(x32, np.float32(last)), np.float32(0))
        return x32.astype(np.int64)
    return index(s, grid[0], grid[1], grid[4]) + columns * index(t, grid[2], grid[3], grid[5])


def multiPlan(width, height, n_ctx, tile=4, lib=None):
    """The dealing of rows behind renderMulti (rtow_multi_plan; no GPU is touched): returns
    (ctx_of_row int32 [height], index_in_ctx int32 [height], rows_per_ctx int32 [n_ctx], slot_bytes)."""
    L = lib if lib is not None else _lib.lib()
    ctx_of_row = np.full(max(height, 0), -1, np.int32)
    index_in_ctx = np.full(max(height, 0), -1, np.int32)
    rows_per_ctx = np.full(max(n_ctx, 0), -1, np.int32)
    slot = C.c_uint64(0)
    _lib.check(L.rtow_multi_plan(width, height, n_ctx, tile, ctx_of_row.ctypes.data if height > 0 else None,
                                 index_in_ctx.ctypes.data if height > 0 else None,
                                 rows_per_ctx.ctypes.data if n_ctx > 0 else None, C.byref(slot)), L)
    return ctx_of_row, index_in_ctx, rows_per_ctx, slot.value


def renderMulti(renderers, width, height, samplesPerPixel, maxDepth, seed=RENDER_SEED, tile=4):
    """One frame on several Renderers of this process at once (normally one per GPU, each holding the
    same world and camera): rows dealt in interleaved `tile`-row tiles, image assembled in host
    memory.  Returns (rgb uint8 [height, width, 3], ray segments)."""
    if not renderers:
        raise ValueError("renderMulti: no renderers")
    L = renderers[0]._lib
    handles = (C.c_void_p * len(renderers))(*[r._h for r in renderers])
    rgb = np.empty((height, width, 3), np.uint8)
    seg = C.c_uint64(0)
    _lib.check(L.rtow_render_multi(handles, len(renderers), width, height, samplesPerPixel, maxDepth, seed, tile,
                                   rgb.ctypes.data, C.byref(seg)))
    return rgb, seg.value


# ---- the reference's per-ray procs, one ray at a time, on a context this module keeps ----------------------
_shared = {"renderer": None, "world": None, "camera": None}


def _shared_renderer(world=None, camera=None):
    """The module's own Renderer (device 0), holding `world` / `camera`.  The reference's procs take the list as
    `var HittablesList` and always see its current contents, so the list is flattened on every call and uploaded
    again whenever the flattened values differ from what the device holds - a sphere replaced or edited in place
    at the same length counts (these wrappers are one-ray slow paths; flattening 484 spheres costs about a
    millisecond).  The camera is compared field by field likewise."""
    if _shared["renderer"] is None:
        _shared["renderer"] = Renderer(0)
    r = _shared["renderer"]
    if world is not None:
        flat = world.flatten() if isinstance(world, HittablesList) else world
        key = tuple(np.ascontiguousarray(flat[k]).tobytes() for k in sorted(flat))
        if _shared["world"] != key:
            _shared["world"] = None      # a refused upload leaves no claim about what the device holds
            r.setWorld(flat)
            _shared["world"] = key
    if camera is not None:
        s = camera.to_struct() if isinstance(camera, Camera) else camera
        key = bytes(s)
        if _shared["camera"] != key:
            _shared["camera"] = None
            r.setCamera(s)
            _shared["camera"] = key
    return r


def rayColor(r, world, depth, seed=RENDER_SEED, index=0):
    """rayColor*(r: Ray, world: var HittablesList, depth: int): Color (raytracer.nim:5-20).  Scattering
    draws from the stream of path (pixel `index`, sample 0) of `seed` instead of the global generator."""
    dev = _shared_renderer(world=world)
    n = index + 1   # ray i of a batch uses stream i: place the ray at position `index`
    o = np.tile(np.asarray(list(r.orig), np.float64), (n, 1))
    d = np.tile(np.asarray(list(r.dir), np.float64), (n, 1))
    col, _, _ = dev.traceRays(o, d, depth, seed=seed, want_hit=False)
    return Color(*col[index])


def hit(world, r, t_min, t_max, rec):
    """hit*(h: var HittablesList, r: Ray, t_min, t_max: float, rec: var HitRecord): bool
    (hittables.nim:79-88).  The closest-hit search runs on the GPU; the record is completed here as
    Sphere.hit does (hitrecord.nim:73-77).  A Hittable (hittables.nim:75-77) or a Sphere (hitrecord.nim:55)
    in place of the list is answered as a one-entry list (hittables.hit does the dispatch)."""
    if not isinstance(world, HittablesList):
        from . import hittables
        return hittables.hit(world, r, t_min, t_max, rec)
    dev = _shared_renderer(world=world)
    _, hid, t = dev.traceRays([list(r.orig)], [list(r.dir)], 0, t_min=t_min, t_max=t_max, want_color=False)
    if hid[0] < 0:
        return False
    s = world[int(hid[0])].hSphere
    rec.t = float(t[0])
    rec.p = r.at(rec.t)
    rec.setFaceNormal(r, (Vec3(*rec.p) - Vec3(*s.center)) / s.radius)
    rec.mat = s.mat
    rec.index = int(hid[0])
    return True


def scatter(m, r_in, rec, seed=RENDER_SEED, index=0):
    """scatter*(m: Material, r_in: Ray, rec: HitRecord, attenuation: var Color, scattered: var Ray): bool
    (hitrecord.nim:127-132).  Python has no var parameters: returns (result, attenuation, scattered).  Draws
    from the stream of path (pixel `index`, sample 0) of `seed` instead of the global generator."""
    dev = _shared_renderer()
    n = index + 1
    kind, albedo, fuzz, ir = m.flat()
    att, out, ok = dev.scatterRays(np.tile(np.asarray(list(r_in.dir), np.float64), (n, 1)),
                                   np.tile(np.asarray(list(rec.normal), np.float64), (n, 1)),
                                   [1 if rec.frontFace else 0] * n, [kind] * n, np.tile(np.asarray(albedo, np.float64), (n, 1)),
                                   [fuzz] * n, [ir] * n, seed=seed)
    return bool(ok[index]), Color(*att[index]), Ray(Point(*rec.p), Vec3(*out[index]))


def getRay(c, s, t, seed=RENDER_SEED, index=0):
    """getRay*(c: Camera, s, t: float): Ray (camera.nim:34-39); the lens sample comes from the stream of
    path (pixel `index`, sample 0)."""
    dev = _shared_renderer(camera=c)
    n = index + 1
    o, d = dev.getRays([s] * n, [t] * n, seed=seed)
    return Ray(Point(*o[index]), Vec3(*d[index]))


def writeColor(f, color, samplesPerPixel):
    """raytracer.nim:22-28 for one pixel's colour sum (host side, binary64)."""
    scale = 1.0 / float(samplesPerPixel)
    r = math.sqrt(color.r * scale)
    g = math.sqrt(color.g * scale)
    b = math.sqrt(color.b * scale)
    clamp = lambda x: min(max(x, 0.0), 0.999)  # noqa: E731
    f.write(f"{int(256 * clamp(r))} {int(256 * clamp(g))} {int(256 * clamp(b))}\n")


def writePPM(f, rgb):
    """The P3 container of raytracer.nim:35,28: header, then one `r g b` line per pixel,
    top row first."""
    h, w, _ = rgb.shape
    f.write(f"P3\n{w} {h}\n255\n")
    flat = rgb.reshape(-1, 3)
    f.write("\n".join(f"{p[0]} {p[1]} {p[2]}" for p in flat.tolist()))
    f.write("\n")


def render(img, f, world, camera, samplesPerPixel, maxDepth, seed=RENDER_SEED, device=0):
    """render*(img, f, world, camera, samplesPerPixel, maxDepth) (raytracer.nim:30-46).

    Writes the PPM file `f`; the pixel values come from the GPU.  Returns the image array too.
    """
    with Renderer(device) as r:
        r.setWorld(world)
        r.setCamera(camera)
        rgb, _, _ = r.renderRows(img.width, img.height, samplesPerPixel, maxDepth, seed)
        text = r.ppm()   # the file's bytes come from the device too; writePPM is the host-side equivalent
    with open(f, "wb") as out:
        out.write(text)
    return rgb


def sceneRedBlue():
    """raytracer.nim:48-59"""
    result = initHittables(0)
    R = math.cos(math.pi / 4.0)
    matLeft = initMaterial(initLambertian(color(0, 0, 1)))
    matRight = initMaterial(initLambertian(color(1, 0, 0)))
    result.add(Sphere(point(-R, 0, -1), R, matLeft))
    result.add(Sphere(point(R, 0, -1), R, matRight))
    return result


def mixOfSpheres():
    """raytracer.nim:61-73; the fourth sphere has a negative radius (a hollow glass shell)."""
    result = initHittables(0)
    matGround = initMaterial(initLambertian(color(0.8, 0.8, 0.0)))
    matCenter = initMaterial(initLambertian(color(0.1, 0.2, 0.5)))
    matLeft = initMaterial(initDielectric(1.5))
    matRight = initMaterial(initMetal(color(0.8, 0.6, 0.2), 1.0))
    result.add(Sphere(point(0, -100.5, -1), 100, matGround))
    result.add(Sphere(point(0, 0, -1), 0.5, matCenter))
    result.add(Sphere(point(-1.0, 0, -1), 0.5, matLeft))
    result.add(Sphere(point(-1.0, 0, -1), -0.4, matLeft))
    result.add(Sphere(point(1.0, 0, -1), 0.5, matRight))
    return result


def randomScene(halfExtent=11, lambertBelow=0.8, metalBelow=0.95, rng=None):
    """raytracer.nim:75-111.  The defaults give the reference's 484-sphere scene; halfExtent and
    the two material thresholds generalise it (a 101 x 101 grid, or an all metal/glass variant).
    Draws come from the Nim generator in its un-seeded state unless `rng` is given, because the
    reference builds the scene before it calls randomize (raytracer.nim:122 vs :135)."""
    # A generator of this call's own: the module-level one (what the reference's rand() acts on) is neither touched
    # nor shared, so scenes can be built from several host threads at once.
    g = rng if rng is not None else nimrandom.Rand()
    rand = g.rand

    def randomVec(lo=0.0, hi=1.0):   # basetypes.nim:110-112 on this generator: x, y, z in this order
        return Vec3(rand(lo, hi), rand(lo, hi), rand(lo, hi))

    result = initHittables(0)
    groundMaterial = initMaterial(initLambertian(color(0.5, 0.5, 0.5)))
    result.add(Sphere(point(0, -1000, 0), 1000, groundMaterial))
    for a in range(-halfExtent, halfExtent):
        for b in range(-halfExtent, halfExtent):
            chooseMat = rand(1.0)
            cx = float(a) + 0.9 * rand(1.0)
            cz = float(b) + 0.9 * rand(1.0)
            center = point(cx, 0.2, cz)
            if length(center - point(4, 0.2, 0)) > 0.9:
                if chooseMat < lambertBelow:
                    first = randomVec()
                    second = randomVec()
                    albedo = Color(*(first * second))
                    sphereMaterial = initMaterial(initLambertian(albedo))
                elif chooseMat < metalBelow:
                    albedo = Color(*randomVec(0.5, 1.0))
                    fuzz = rand(0.0, 0.5)
                    sphereMaterial = initMaterial(initMetal(albedo, fuzz))
                else:
                    sphereMaterial = initMaterial(initDielectric(1.5))
                result.add(Sphere(center, 0.2, sphereMaterial))
    result.add(Sphere(point(0, 1, 0), 1.0, initMaterial(initDielectric(1.5))))
    result.add(Sphere(point(-4, 1, 0), 1.0, initMaterial(initLambertian(color(0.4, 0.2, 0.1)))))
    result.add(Sphere(point(4, 1, 0), 1.0, initMaterial(initMetal(color(0.7, 0.6, 0.5), 0.0))))
    result.draws = g.draws
    return result


def bookCamera(aspectRatio):
    """The camera of main (raytracer.nim:125-132)."""
    return initCamera(point(13, 2, 3), point(0, 0, 0), vec3(0, 1, 0), vfov=20, aspectRatio=aspectRatio,
                      aperture=0.1, focusDist=10.0)


def main(width=1200, ratio=3.0 / 2.0, samplesPerPixel=500, maxDepth=50, out="/tmp/test_small.ppm"):
    """raytracer.nim:113-138 with its constants as defaults."""
    img = Image(width, int(width / ratio))
    world = randomScene()
    cam = bookCamera(ratio)
    render(img, out, world, cam, samplesPerPixel, maxDepth)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    if args:
        main(width=int(args[0]), samplesPerPixel=int(args[1]) if len(args) > 1 else 500,
             out=args[2] if len(args) > 2 else "/tmp/test_small.ppm")
    else:
        main()
