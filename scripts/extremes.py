"""Extreme shapes through size-independent properties: millions of samples on a 2x2 frame, frames 100 000
pixels wide with one sample, rows rendered alone against the same rows of the whole frame."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from raytracinginoneweekend_amd import raytracer as R
r = R.Renderer(0); r.setWorld(R.randomScene()); r.setCamera(R.bookCamera(1.0))
# many samples on a tiny frame: one launch against two passes
W = H = 2; spp = 3_000_001
whole = r.renderRows(W, H, spp, 50, want_accum=True); seg = r.stats()["ray_segments"]
r.renderPass(W, H, 0, 1_000_000, 50); r.renderPass(W, H, 1_000_000, spp - 1_000_000, 50)
rgb, acc = r.readImage(want_accum=True)
print("2x2 spp 3000001: passes equal", bool((acc == whole[1]).all()), "segments", seg, "mean per path", seg / (4 * spp))
# a very wide frame, one sample: rows subset against the full frame
W, H = 100_000, 2
r.setCamera(R.bookCamera(W / H))
full = r.renderRows(W, H, 1, 50, want_accum=True)
row1 = r.renderRows(W, H, 1, 50, rows=[1], want_accum=True)
print("100000x2 spp 1: row 1 alone equals row 1 of the frame", bool((row1[1][0] == full[1][1]).all()))
# wide and deep at once: 30000 x 3, spp 3, against two passes
W, H = 30_000, 3
r.setCamera(R.bookCamera(W / H))
whole = r.renderRows(W, H, 3, 50, want_accum=True)
r.renderPass(W, H, 0, 1, 50); r.renderPass(W, H, 1, 2, 50)
rgb, acc = r.readImage(want_accum=True)
print("30000x3 spp 3: passes equal", bool((acc == whole[1]).all()))
