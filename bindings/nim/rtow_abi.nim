## Nim declarations of the C ABI in include/rtow.h (librtow_hip.so).
## Unverified in the build image (no Nim toolchain there); kept thin on purpose.
const rtowLib* = "librtow_hip.so"

type
  RtowCtx* = distinct pointer
  RtowCameraF64* {.bycopy.} = object      ## rtow_camera_f64: Camera's fields as initCamera computes them
    origin*, lowerLeft*, horizontal*, vertical*, u*, v*, w*: array[3, cdouble]
    lensRadius*: cdouble
  RtowStats* {.bycopy.} = object
    raySegments*, paths*: uint64
    traceMs*, totalMs*: cfloat
    nSpheres*, nBig*: int32
    shaderClockMhz*: cfloat
    gatherPath*: int32                    ## RTOW_GATHER_*: how rtow_render_multi brought the rows home
    cameraScreen*: int32                  ## RTOW_SCREEN_*: what the camera rays saw in front of the sphere screen
    reserved: int32

proc rtow_last_error*(): cstring {.importc, dynlib: rtowLib.}
proc rtow_create*(device: cint, ctx: ptr RtowCtx): cint {.importc, dynlib: rtowLib.}
proc rtow_destroy*(ctx: RtowCtx): cint {.importc, dynlib: rtowLib.}
proc rtow_set_scene_f64*(ctx: RtowCtx, n: cint, cx, cy, cz, radius: ptr cdouble, kind: ptr uint8,
                         albedoRgb, fuzz, ir: ptr cdouble): cint {.importc, dynlib: rtowLib.}
proc rtow_set_hittables*(ctx: RtowCtx, n: cint, hittableKind: ptr uint8, cx, cy, cz, radius: ptr cdouble,
                         materialKind: ptr uint8, albedoRgb, fuzz, ir: ptr cdouble): cint {.importc, dynlib: rtowLib.}
proc rtow_multi_plan*(w, h, nCtx, tileRows: cint, ctxOfRow, indexInCtx, rowsPerCtx: ptr int32,
                      slotBytes: ptr uint64): cint {.importc, dynlib: rtowLib.}
proc rtow_set_camera_f64*(ctx: RtowCtx, cam: ptr RtowCameraF64): cint {.importc, dynlib: rtowLib.}
proc rtow_interval_plan*(cam: ptr RtowCameraF64, n: cint, cx, cy, cz, radius: ptr cdouble, midS, halfS, midT, halfT: ptr cfloat,
                         covered: ptr cint): cint {.importc, dynlib: rtowLib.}
proc rtow_mask_plan*(cam: ptr RtowCameraF64, n: cint, cx, cy, cz, radius: ptr cdouble, groupMask: ptr uint32, grid: ptr cfloat,
                     columns, covered: ptr cint): cint {.importc, dynlib: rtowLib.}
proc rtow_render*(ctx: RtowCtx, w, h, spp, maxDepth: cint, seed: uint64, rows: ptr int32, nRows: cint,
                  rgbOut: ptr uint8, accumOut: ptr uint64, hitIdOut: ptr int32): cint {.importc, dynlib: rtowLib.}
proc rtow_render_multi*(ctxs: ptr RtowCtx, nCtx: cint, w, h, spp, maxDepth: cint, seed: uint64, tileRows: cint,
                        rgb: ptr uint8, raySegments: ptr uint64): cint {.importc, dynlib: rtowLib.}
proc rtow_render_pass*(ctx: RtowCtx, w, h, firstSample, nSamples, maxDepth: cint, seed: uint64, rows: ptr int32,
                       nRows: cint, rgbDev: pointer): cint {.importc, dynlib: rtowLib.}
proc rtow_render_noalloc*(ctx: RtowCtx, w, h, spp, maxDepth: cint, seed: uint64, rowsDev: pointer, nRows: cint,
                          rgbDev: pointer): cint {.importc, dynlib: rtowLib.}
proc rtow_trace_rays*(ctx: RtowCtx, n: cint, origins, directions: ptr cdouble, tMin, tMax: cdouble, maxDepth: cint,
                      seed: uint64, colorOut: ptr cdouble, hitIdOut: ptr int32, tOut: ptr cdouble): cint {.importc, dynlib: rtowLib.}
proc rtow_get_rays*(ctx: RtowCtx, n: cint, s, t: ptr cdouble, seed: uint64,
                    originsOut, directionsOut: ptr cdouble): cint {.importc, dynlib: rtowLib.}
proc rtow_scatter_rays*(ctx: RtowCtx, n: cint, directions, normals: ptr cdouble, frontFace, kind: ptr uint8,
                        albedoRgb, fuzz, ir: ptr cdouble, seed: uint64, attenuationOut, scatteredDirOut: ptr cdouble,
                        scatteredOut: ptr uint8): cint {.importc, dynlib: rtowLib.}
proc rtow_read_image*(ctx: RtowCtx, rgb: ptr uint8, accum: ptr uint64): cint {.importc, dynlib: rtowLib.}
proc rtow_format_ppm*(ctx: RtowCtx, binary: cint, dst: cstring, capacity: uint64, length: ptr uint64): cint {.importc, dynlib: rtowLib.}
proc rtow_selftest_arithmetic*(ctx: RtowCtx, mismatches: ptr uint64): cint {.importc, dynlib: rtowLib.}
proc rtow_get_stats*(ctx: RtowCtx, stats: ptr RtowStats): cint {.importc, dynlib: rtowLib.}
proc rtow_sync*(ctx: RtowCtx): cint {.importc, dynlib: rtowLib.}
