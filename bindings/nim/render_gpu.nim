## Drop-in bodies for the procs of the hot path, same names and parameter lists as the reference:
##   render*    raytracer.nim:30-32   (the triple loop replaced by one call; `seed` added, defaulted)
##   rayColor*  raytracer.nim:5       hit*  hittables.nim:79       getRay*  camera.nim:34       scatter*  hitrecord.nim:127
## `flatten` (Camera -> RtowCameraF64) must live in camera.nim because Camera's fields are private
## (camera.nim:5-10): it copies the eight fields, in declaration order.  Import this module INSTEAD of the reference's own definitions
## of these five procs.  Unverified in the build image (no Nim toolchain there).
import basetypes, hittables, camera, hitrecord, rtow_abi

proc check(status: cint, what: string) =
  if status != 0: raise newException(IOError, what & ": " & $rtow_last_error())

type Flat = object
  cx, cy, cz, rad, albedo, fuzz, ir: seq[cdouble]
  kind: seq[uint8]                        # MaterialKind of each sphere
  hkind: seq[uint8]                       # Hittable.kind of each entry (hittables.nim:6)

proc flatten(world: var HittablesList): Flat =
  for obj in world:                       # list order is kept: it decides ties (hittables.nim:84-88)
    result.hkind.add obj.kind.ord.uint8
    let s = obj.hSphere
    result.cx.add s.center.x; result.cy.add s.center.y; result.cz.add s.center.z
    result.rad.add s.radius
    result.kind.add s.mat.kind.ord.uint8
    case s.mat.kind
    of mkLambertian:
      for i in 0 ..< 3: result.albedo.add s.mat.mLambertian.albedo[i]
      result.fuzz.add 0; result.ir.add 1
    of mkMetal:
      for i in 0 ..< 3: result.albedo.add s.mat.mMetal.albedo[i]
      result.fuzz.add s.mat.mMetal.fuzz; result.ir.add 1
    of mkDielectric:
      for i in 0 ..< 3: result.albedo.add 1
      result.fuzz.add 0; result.ir.add s.mat.mDielectric.ir

template p(s: seq[cdouble]): ptr cdouble = (if s.len > 0: unsafeAddr s[0] else: nil)

proc upload(ctx: RtowCtx, f: Flat) =
  let kinds = if f.kind.len > 0: unsafeAddr f.kind[0] else: nil   # an empty world is legal: every ray sees the sky
  let hkinds = if f.hkind.len > 0: unsafeAddr f.hkind[0] else: nil
  # the list with its Hittable kinds: the library, like hit*(h: Hittable) (hittables.nim:75-77), knows htSphere only
  check rtow_set_hittables(ctx, f.rad.len.cint, hkinds, p(f.cx), p(f.cy), p(f.cz), p(f.rad), kinds, p(f.albedo), p(f.fuzz),
                           p(f.ir)), "rtow_set_hittables"

proc upload(ctx: RtowCtx, world: var HittablesList) = upload(ctx, flatten(world))

proc render*(img: Image, f: string, world: var HittablesList, camera: Camera,
             samplesPerPixel, maxDepth: int, seed: uint64 = 0xE7) =
  var ctx: RtowCtx
  check rtow_create(0, addr ctx), "rtow_create"
  defer: discard rtow_destroy(ctx)
  upload(ctx, world)
  var cam = camera.flatten()              # proc flatten*(c: Camera): RtowCameraF64, added to camera.nim
  check rtow_set_camera_f64(ctx, addr cam), "rtow_set_camera_f64"
  # the image stays in the context; the P3 text of raytracer.nim:28,35 is composed on the device
  check rtow_render(ctx, img.width.cint, img.height.cint, samplesPerPixel.cint, maxDepth.cint, seed,
                    nil, img.height.cint, nil, nil, nil), "rtow_render"
  var length: uint64
  check rtow_format_ppm(ctx, 0, nil, 0, addr length), "rtow_format_ppm (length)"
  var text = newString(length.int)
  check rtow_format_ppm(ctx, 0, text.cstring, length, addr length), "rtow_format_ppm"
  writeFile(f, text)

# ---- per-ray procs: a context is kept per thread.  The reference's procs see the CURRENT contents of their
# `var HittablesList`, so the list is flattened on every call and uploaded again whenever the values differ from
# what the device holds (a sphere edited in place at the same length counts); one-ray slow paths anyway.
var gCtx {.threadvar.}: RtowCtx
var gWorld {.threadvar.}: Flat
var gWorldValid {.threadvar.}: bool

proc device(): RtowCtx =
  if gCtx.pointer == nil: check rtow_create(0, addr gCtx), "rtow_create"
  gCtx

proc deviceWith(world: var HittablesList): RtowCtx =
  result = device()
  let f = flatten(world)
  if not gWorldValid or f != gWorld:
    gWorldValid = false
    upload(result, f)
    gWorld = f; gWorldValid = true

proc rayColor*(r: Ray, world: var HittablesList, depth: int): Color =
  let ctx = deviceWith(world)
  var o = [r.orig.x, r.orig.y, r.orig.z]
  var d = [r.dir.x, r.dir.y, r.dir.z]
  var c: array[3, cdouble]
  check rtow_trace_rays(ctx, 1, addr o[0], addr d[0], 0.001, Inf, depth.cint, 0xE7'u64, addr c[0], nil, nil), "rtow_trace_rays"
  color(c[0], c[1], c[2])

proc hit*(h: var HittablesList, r: Ray, t_min, t_max: float, rec: var HitRecord): bool =
  let ctx = deviceWith(h)
  var o = [r.orig.x, r.orig.y, r.orig.z]
  var d = [r.dir.x, r.dir.y, r.dir.z]
  var id: int32
  var t: cdouble
  check rtow_trace_rays(ctx, 1, addr o[0], addr d[0], t_min, t_max, 0, 0xE7'u64, nil, addr id, addr t), "rtow_trace_rays"
  if id < 0: return false
  let s = h[id.int].hSphere
  rec.t = t
  rec.p = r.at(rec.t)
  rec.setFaceNormal(r, (rec.p - s.center) / s.radius)      # hitrecord.nim:73-77
  rec.mat = s.mat
  true

proc hit*(s: Sphere, r: Ray, t_min, t_max: float, rec: var HitRecord): bool =
  ## hitrecord.nim:55-79: one sphere is a one-entry list
  var one = initHittables(0)
  one.add s
  hit(one, r, t_min, t_max, rec)

proc hit*(h: Hittable, r: Ray, t_min, t_max: float, rec: var HitRecord): bool =
  ## hittables.nim:75-77
  case h.kind
  of htSphere: result = h.hSphere.hit(r, t_min, t_max, rec)

proc scatter*(m: Material, r_in: Ray, rec: HitRecord, attenuation: var Color, scattered: var Ray): bool =
  ## hitrecord.nim:127-132; draws from the stream of path (pixel 0, sample 0) instead of the global generator
  let ctx = device()
  var d = [r_in.dir.x, r_in.dir.y, r_in.dir.z]
  var n = [rec.normal.x, rec.normal.y, rec.normal.z]
  var front = (if rec.frontFace: 1'u8 else: 0'u8)
  var kind = m.kind.ord.uint8
  var albedo = [1.0, 1.0, 1.0]
  var fuzz = 0.0
  var ir = 1.0
  case m.kind
  of mkLambertian: albedo = [m.mLambertian.albedo.r, m.mLambertian.albedo.g, m.mLambertian.albedo.b]
  of mkMetal:
    albedo = [m.mMetal.albedo.r, m.mMetal.albedo.g, m.mMetal.albedo.b]
    fuzz = m.mMetal.fuzz
  of mkDielectric: ir = m.mDielectric.ir
  var att, sd: array[3, cdouble]
  var ok: uint8
  check rtow_scatter_rays(ctx, 1, addr d[0], addr n[0], addr front, addr kind, addr albedo[0], addr fuzz, addr ir,
                          0xE7'u64, addr att[0], addr sd[0], addr ok), "rtow_scatter_rays"
  attenuation = color(att[0], att[1], att[2])
  scattered = initRay(rec.p, vec3(sd[0], sd[1], sd[2]))
  ok != 0

proc getRay*(c: Camera, s, t: float): Ray =
  let ctx = device()
  var cam = c.flatten()
  check rtow_set_camera_f64(ctx, addr cam), "rtow_set_camera_f64"
  var ss = s.cdouble
  var tt = t.cdouble
  var o, d: array[3, cdouble]
  check rtow_get_rays(ctx, 1, addr ss, addr tt, 0xE7'u64, addr o[0], addr d[0]), "rtow_get_rays"
  initRay(point(o[0], o[1], o[2]), vec3(d[0], d[1], d[2]))
