"""Instrumented build: when do wavefronts start, run out of work and retire within one launch?"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTOW_LIB_PATH"] = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "librtow_hip_timeline.so")
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import raytracer as R, _lib
L = _lib.lib()
r = R.Renderer(0); r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(1.5))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 62
for _ in range(2):
    r.renderAsync(1200, 800, spp, 50); r.sync()
st = r.stats()
n = 6144
buf = (C.c_ulonglong * (12 * n))()
assert L.rtow_debug_wave_times(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 12).astype(np.float64)
t = t[t[:, 0] > 0]   # waves that ran (the grid is CUs x resident workgroups x 4)
print(f"{len(t)} waves")
live_at_dry, it_dry, it_end = t[:, 3], t[:, 4], t[:, 5]
raw = t.copy()
t = t[:, :3] / 100.0  # microseconds
t0 = t[:, 0].min()
start, dry, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
print(f"spp {spp}: trace_ms {st['trace_ms']:.3f}  span of stamps {end.max()/1e3:.3f} ms")
q = [0, 1, 10, 50, 90, 99, 100]
print("wave start  (us) percentiles", q, np.percentile(start, q).round(1))
print("counter dry (us) percentiles", q, np.percentile(dry, q).round(1))
print("wave end    (us) percentiles", q, np.percentile(end, q).round(1))
print("end - dry   (us) percentiles", q, np.percentile(end - dry, q).round(1))
alive = [(end > x).sum() for x in np.linspace(np.percentile(dry, 50), end.max(), 12)]
print("waves still running at 12 instants from median-dry to end:", alive)
print("live lanes at dry        percentiles", q, np.percentile(live_at_dry, q))
print("iterations before dry    percentiles", q, np.percentile(it_dry, q))
print("iterations after dry     percentiles", q, np.percentile(it_end - it_dry, q))
late = dry > np.percentile(dry, 50) + 200
print("waves seeing dry >200us after the median:", late.sum(), " their iterations before dry", np.percentile(it_dry[late], q) if late.any() else None)
print("us per iteration after dry, percentiles", q, np.percentile((end - dry) / np.maximum(1, it_end - it_dry), q).round(1))
longw = (it_end - it_dry) >= 30
print("waves with >= 30 iterations after dry:", longw.sum(), " us per iteration percentiles", q,
      np.percentile(((end - dry) / np.maximum(1, it_end - it_dry))[longw], q).round(1) if longw.any() else None)
print("overall us per iteration before dry, percentiles", q, np.percentile((dry - start) / np.maximum(1, it_dry), q).round(1))
one = (it_end - it_dry) == 1
clk = raw[one, 6:10]   # core-clock stamps of the last iteration: after refill, after big, after scan, after shade
print("waves with exactly one iteration after dry:", one.sum())
print("  last iteration, core cycles: big", np.percentile(clk[:, 1] - clk[:, 0], [10, 50, 90]).round(0),
      " scan", np.percentile(clk[:, 2] - clk[:, 1], [10, 50, 90]).round(0), " shade", np.percentile(clk[:, 3] - clk[:, 2], [10, 50, 90]).round(0))
rt_last_iter_end = raw[one, 11] / 100.0 - t0
rt_loop_exit = raw[one, 10] / 100.0 - t0
print("  dry -> end of last iteration (us)", np.percentile(rt_last_iter_end - dry[one], [10, 50, 90]).round(1),
      " -> loop exit", np.percentile(rt_loop_exit - rt_last_iter_end, [10, 50, 90]).round(1),
      " -> wave end (final flush)", np.percentile(end[one] - rt_loop_exit, [10, 50, 90]).round(1))
