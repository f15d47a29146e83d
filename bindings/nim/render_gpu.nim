## Drop-in body for `render*` of raytracer.nim: same signature, the triple loop replaced by one
## call.  `flatten` must live in camera.nim / hittables.nim because Camera's fields are private.
import strformat
import basetypes, hittables, camera, hitrecord, rtow_abi

proc check(status: cint, what: string) =
  if status != 0: raise newException(IOError, what & ": " & $rtow_last_error())

proc renderGpu*(img: Image, f: string, world: var HittablesList, camera: Camera,
                samplesPerPixel, maxDepth: int, seed: uint64 = 0xE7) =
  var cx, cy, cz, rad, albedo, fuzz, ir: seq[cfloat]
  var kind: seq[uint8]
  for obj in world:                       # list order is kept: it decides ties
    let s = obj.hSphere
    cx.add s.center.x.cfloat; cy.add s.center.y.cfloat; cz.add s.center.z.cfloat
    rad.add s.radius.cfloat
    kind.add s.mat.kind.ord.uint8
    case s.mat.kind
    of mkLambertian:
      for i in 0 ..< 3: albedo.add s.mat.mLambertian.albedo[i].cfloat
      fuzz.add 0; ir.add 1
    of mkMetal:
      for i in 0 ..< 3: albedo.add s.mat.mMetal.albedo[i].cfloat
      fuzz.add s.mat.mMetal.fuzz.cfloat; ir.add 1
    of mkDielectric:
      for i in 0 ..< 3: albedo.add 1
      fuzz.add 0; ir.add s.mat.mDielectric.ir.cfloat
  var ctx: RtowCtx
  check rtow_create(0, addr ctx), "rtow_create"
  defer: discard rtow_destroy(ctx)
  let n = rad.len.cint
  check rtow_set_scene(ctx, n, addr cx[0], addr cy[0], addr cz[0], addr rad[0], addr kind[0],
                       addr albedo[0], addr fuzz[0], addr ir[0]), "rtow_set_scene"
  var cam = camera.flatten()              # proc flatten*(c: Camera): RtowCamera, added to camera.nim
  check rtow_set_camera(ctx, addr cam), "rtow_set_camera"
  # the image stays in the context; the P3 text of raytracer.nim:28,35 is composed on the device
  check rtow_render(ctx, img.width.cint, img.height.cint, samplesPerPixel.cint, maxDepth.cint, seed,
                    nil, img.height.cint, nil, nil, nil), "rtow_render"
  var length: uint64
  check rtow_format_ppm(ctx, 0, nil, 0, addr length), "rtow_format_ppm (length)"
  var text = newString(length.int)
  check rtow_format_ppm(ctx, 0, text.cstring, length, addr length), "rtow_format_ppm"
  writeFile(f, text)
