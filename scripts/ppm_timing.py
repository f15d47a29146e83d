"""Time the device-side PPM text (rtow_format_ppm) against the host writers at 1200x800 and 3840x2160."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import raytracer as R
r = R.Renderer(0); r.setWorld(R.randomScene())
for W, H in ((1200, 800), (3840, 2160)):
    r.setCamera(R.bookCamera(W / H))
    rgb = r.renderRows(W, H, 1, 50)[0]
    t0 = time.perf_counter(); text = r.ppm(); t1 = time.perf_counter() - t0
    r.renderRows(W, H, 1, 50)
    t0 = time.perf_counter(); text = r.ppm(); t1b = time.perf_counter() - t0
    t0 = time.perf_counter(); p6 = r.ppm(binary=True); t2 = time.perf_counter() - t0
    import ctypes as C, numpy as np
    n = C.c_uint64(0)
    r._lib.rtow_format_ppm(r._h, 0, None, 0, C.byref(n))
    buf = np.zeros(n.value, np.uint8)          # touched once: the copy is timed without page faults
    r.renderRows(W, H, 1, 50)
    t0 = time.perf_counter(); r._lib.rtow_format_ppm(r._h, 0, buf.ctypes.data, n.value, C.byref(n)); t4 = time.perf_counter() - t0
    t0 = time.perf_counter(); r._lib.rtow_format_ppm(r._h, 0, buf.ctypes.data, n.value, C.byref(n)); t5 = time.perf_counter() - t0
    print(f"  into a resident caller buffer: compose + copy {t4*1e3:.1f} ms, copy only (text cached) {t5*1e3:.1f} ms", flush=True)
    t0 = time.perf_counter(); s = io.StringIO(); R.writePPM(s, rgb); t3 = time.perf_counter() - t0
    print(f"{W}x{H}: P3 {len(text)/1e6:.1f} MB device+copy {t1*1e3:.1f} ms (2nd frame {t1b*1e3:.1f} ms), P6 {t2*1e3:.1f} ms, "
          f"Python host writer {t3*1e3:.0f} ms", flush=True)
