"""Instrumented build (librtow_hip_prof.so, -DRTOW_PROFILE_COUNTERS): wave-level statistics of the two kinds
of iteration.  usage: prof_counters.py [spp] [C2|C3|C5]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTOW_LIB_PATH"] = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "librtow_hip_prof.so")
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import raytracer as R, _lib
L = _lib.lib()
wl = sys.argv[2] if len(sys.argv) > 2 else 'C2'
scene = {'C2': lambda: R.randomScene(), 'C3': lambda: R.randomScene(lambertBelow=0.0, metalBelow=0.5), 'C5': lambda: R.randomScene(halfExtent=50)}[wl]()
r = R.Renderer(0); r.setWorld(scene); r.setCamera(R.bookCamera(1.5))
b = (C.c_ulonglong * 32)()
L.rtow_debug_counters(b, 1)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r.renderAsync(1200, 800, spp, 50); r.sync(); st = r.stats()
L.rtow_debug_counters(b, 0)
it, prim = b[3], b[14]
bounce = it - prim
print(f"segments {st['ray_segments']}  paths {st['paths']}  wave iterations {it}: PRIMARY {prim} ({prim/it:.3f}), BOUNCE {bounce}")
print(f"lanes with a ray: PRIMARY {st['paths']/(64*prim):.4f}, BOUNCE {b[4]/(64*bounce):.4f}, all scans {(st['paths']+b[4])/(64*it):.4f}")
print(f"cycles per iteration: all {b[15]/it:.0f};  hand-out {b[5]/it:.0f}")
print(f"  PRIMARY {(b[1]+b[2]+b[13])/prim:.0f} = set-up {b[1]/prim:.0f} + scan {b[2]/prim:.0f} + shading {b[13]/prim:.0f}")
print(f"  BOUNCE  {(b[6]+b[7]+b[8])/bounce:.0f} = set-up {b[6]/bounce:.0f} + scan {b[7]/bounce:.0f} + shading {b[8]/bounce:.0f}")
print(f"per iteration: groups with a survivor {b[9]/it:.2f}, exact-test rounds binary32 {b[10]/bounce:.2f} per BOUNCE, binary64 {b[12]/prim:.2f} per PRIMARY, "
      f"unit-ball passes {b[11]/it:.2f}")
print(f"pixel-sum cache: {b[0]} slots written to the frame = {b[0]/it:.3f} per iteration, {b[0]/(1200*800):.2f} per pixel, {b[0]*192/1e6:.0f} MB of 64-byte atomics")
# lane utilisation of the phases outside the scan (lanes at work / 64, per iteration of that kind)
print(f"hand-out: {b[23]/(64*bounce):.3f} of lanes take a pooled path per BOUNCE iteration")
print(f"PRIMARY shading: sky {b[21]/(64*prim):.3f}, hit (scatter) {b[22]/(64*prim):.3f} of lanes")
print(f"BOUNCE shading: sky {b[16]/(64*bounce):.3f}, Lambertian {b[17]/(64*bounce):.3f}, metal {b[18]/(64*bounce):.3f}, "
      f"dielectric {b[19]/(64*bounce):.3f} of lanes")
if b[11]:
    print(f"unit-ball search: {b[11]/it:.2f} passes per iteration (both kinds), {b[20]/b[11]:.1f} lanes looking per pass")
print(f"camera-ray screen: {b[24]/prim:.1f} groups of a PRIMARY scan go on to the filter (of {(len(scene)+7)//8*2}); "
      f"{b[25]/prim:.1f} pass the interval test's first axis" + (f"; {b[26]/prim:.1f} pass the mask test" if b[26] else ""))
