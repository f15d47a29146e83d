"""Four host threads, one context each on the same GPU, render at the same time; every result is compared with a lone
render.  usage: thread_race_probe.py lib.so [rounds]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracinginoneweekend_amd import _lib, raytracer as R
L = _lib.load(os.path.abspath(sys.argv[1]), strict=False)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 160, 100
jobs = [(3, 50), (8, 50), (17, 6), (40, 50)]
scene = R.randomScene().flatten(); scene.pop("hittable_kind")
base = R.Renderer(0, lib=L); base.setWorld(scene); base.setCamera(R.bookCamera(W / H))
alone = [base.renderRows(W, H, spp, depth, want_accum=True)[1] for spp, depth in jobs]
again = [base.renderRows(W, H, spp, depth, want_accum=True)[1] for spp, depth in jobs]
print("lone renders repeatable:", all((a == b).all() for a, b in zip(alone, again)))
lone_bad = 0
for rep in range(int(os.environ.get("LONE_REPEATS", "0"))):
    base.setWorld(scene); base.setCamera(R.bookCamera(W / H))
    for k, (spp, depth) in enumerate(jobs):
        g = base.renderRows(W, H, spp, depth, want_accum=True)[1]
        if not (g == alone[k]).all():
            lone_bad += 1
            d = np.argwhere((g != alone[k]).any(axis=2))
            print(f"lone repeat {rep} job {k}: {len(d)} pixels differ, first {d[:4].tolist()}")
print("lone repeats differing:", lone_bad)
bad = 0
for rnd in range(rounds):
    got = [[] for _ in jobs]
    def work(k):
        with R.Renderer(0, lib=L) as r:
            r.setWorld(scene); r.setCamera(R.bookCamera(W / H))
            for _ in range(5):
                got[k].append(r.renderRows(W, H, jobs[k][0], jobs[k][1], want_accum=True)[1])
    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in ts]; [t.join() for t in ts]
    for k in range(len(jobs)):
        for i, g in enumerate(got[k]):
            if not (g == alone[k]).all():
                bad += 1
                d = np.argwhere((g != alone[k]).any(axis=2))
                print(f"round {rnd} job {k} render {i}: {len(d)} pixels differ, first {d[:4].tolist()}, "
                      f"got {g[tuple(d[0])].tolist()} want {alone[k][tuple(d[0])].tolist()}")
print(f"{sys.argv[1]}: {bad} differing renders of {rounds * 5 * len(jobs)}")
