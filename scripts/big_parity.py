"""One large bit-exact comparison against the CPU mirror (all waves busy, chunk sizes shrinking, pools
cycling): W H spp from the command line (default 600 400 16), depth 50 on the book scene and on the metal/glass variant."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from raytracinginoneweekend_amd import raytracer as R
W, H, spp = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else ("600", "400", "16")))
for kw, okw in (({}, {}), ({"lambertBelow": 0.0, "metalBelow": 0.5}, {"lambert_below": 0.0, "metal_below": 0.5})):
    with R.Renderer(0) as r:
        r.setWorld(R.randomScene(**kw)); r.setCamera(R.bookCamera(W / H))
        rgb, acc, hid = r.renderRows(W, H, spp, 50, want_accum=True, want_hit_id=True)
        seg = r.stats()["ray_segments"]
    arr, n, _, _ = O.random_scene(**okw)
    t0 = time.time()
    m = O.render_mirror32(arr, n, O.book_camera(W / H), W, H, spp, 50, threads=16)
    print(f"{okw or 'book'}: mirror {time.time() - t0:.1f} s; hit ids equal {(hid == m[2]).all()}, sums equal {(acc == m[1]).all()}, "
          f"pixels equal {(rgb == m[0]).all()}, segments {seg} vs {m[3]}", flush=True)
    assert (hid == m[2]).all() and (acc == m[1]).all() and (rgb == m[0]).all() and seg == m[3]
