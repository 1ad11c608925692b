#!/usr/bin/env python3
"""Random extractor configurations (image size, features, levels, scale factor, thresholds, lapping area), device vs
oracle, bit-exact: keypoints, descriptors, monoIndex.  Exercises both LDS pitches of the FAST kernel (cells wider than
36 px appear for widths just below a multiple of 30) and every quadtree regime."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "3")))
N = int(os.environ.get("FUZZ_N", "40"))
bad = 0
ran = 0
for it in range(N):
    w = int(rng.integers(140, 1500))
    h = int(rng.integers(120, 700))
    nf = int(rng.choice([50, 300, 1000, 2000, 4000]))
    nl = int(rng.integers(3, 9))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5]))
    ini, mn = int(rng.choice([20, 20, 30, 12])), int(rng.choice([7, 7, 5, 12]))
    mn = min(mn, ini)
    lap = (0, int(rng.choice([0, 0, 1000, w // 2])))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        img = synth.make_frame(w, h, step=it)
    elif kind == 1:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((((xx // 9) + (yy // 11)) & 1) * 150 + 40 + rng.integers(0, 8, (h, w))).astype(np.uint8)
    img = np.ascontiguousarray(img)
    try:
        fe = V.FExtractor(nf, sf, nl, ini, mn, w, h)
    except V.VslamError as e:
        print("create refused", (w, h, nf, nl, sf), str(e)[:70])
        continue
    try:
        k, d, mono = fe.compute(img, lap)
        k, d = k.copy(), d.copy()
    finally:
        fe.close()
    rk, rd, rmono = orbo.Extractor(nf, sf, nl, ini, mn).compute(img, lap)
    ok = mono == rmono and len(k) == len(rk) and np.array_equal(k.view(np.uint8), rk.view(np.uint8)) and np.array_equal(d, rd)
    ran += 1
    if not ok:
        bad += 1
        print("MISMATCH", dict(w=w, h=h, nf=nf, nl=nl, sf=sf, ini=ini, mn=mn, lap=lap, kind=kind), len(k), len(rk))
print("fuzz:", ran, "cases run,", bad, "mismatches")
sys.exit(1 if bad else 0)
