"""Ad-hoc GPU check: GPU vs FP32 mirror (bit-exact) and vs FP64 oracle (tolerance)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from raytracinginoneweekend_amd import raytracer as R

def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(255 ** 2 / mse) if mse > 0 else float("inf")

world = R.randomScene(); cam = R.bookCamera(1.5)
arr, n, _, _ = O.random_scene(); ocam = O.book_camera(1.5)
r = R.Renderer(0); r.setWorld(world); r.setCamera(cam)
for (W, H, spp, depth) in [(64, 48, 1, 1), (120, 80, 1, 1), (120, 80, 4, 50), (300, 200, 16, 50)]:
    ocam = O.book_camera(W / H); r.setCamera(R.bookCamera(W / H))
    t = time.time(); rgb, acc, hid = r.renderRows(W, H, spp, depth, want_accum=True, want_hit_id=True); tg = time.time() - t
    st = r.stats()
    m_rgb, m_acc, m_hid, m_seg = O.render_mirror32(arr, n, ocam, W, H, spp, depth)
    print(f"{W}x{H} spp{spp} d{depth}: gpu {tg*1e3:.1f} ms trace {st['trace_ms']:.2f} ms seg gpu={st['ray_segments']} mirror={m_seg} | "
          f"rgb mismatch {np.sum(rgb != m_rgb)} acc mismatch {np.sum(acc != m_acc)} hit mismatch {np.sum(hid != m_hid)}")
    if spp >= 4:
        f_rgb, f_acc, f_hid, f_seg = O.render_ref64(arr, n, ocam, W, H, spp, depth)
        print(f"     vs FP64: PSNR {psnr(rgb, f_rgb):.2f} dB  MAE {np.mean(np.abs(rgb.astype(int)-f_rgb.astype(int))):.4f} hit mismatch {np.sum(hid != f_hid)} seg64={f_seg}")
# throughput probe
for (W, H, spp) in [(1200, 800, 8), (1200, 800, 32), (1200, 800, 128)]:
    r.setCamera(R.bookCamera(W / H))
    r.renderAsync(W, H, spp, 50); r.sync(); st = r.stats()
    flop = st['ray_segments'] * (18 * 484 + 65)
    print(f"{W}x{H} spp{spp}: trace {st['trace_ms']:.2f} ms total {st['total_ms']:.2f} ms segs {st['ray_segments']} "
          f"-> {st['ray_segments']/st['trace_ms']/1e3:.1f} Mrays/s  {flop/st['trace_ms']/1e9:.2f} TFLOP/s  {flop/st['trace_ms']/1e9/157.3*100:.1f}% roofline")
