#!/usr/bin/env python3
"""Random SearchForInitialization problems on extractor output: geometry, feature count, window, ratio, orientation check
and the length of the sorted prefix drawn per case; two consecutive frames (synthetic motion, or the same noise image
shifted) through the device extractor and the device matcher -- vnMatches12 / nmatches / vbPrevMatched vs the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "5")))
N = int(os.environ.get("FUZZ_N", "30"))
bad = ran = 0
for it in range(N):
    w = int(rng.integers(200, 1400))
    h = int(rng.integers(160, 600))
    nf = int(rng.choice([200, 500, 1000, 2000, 3000]))
    window = int(rng.choice([10, 30, 100, 100, 200]))
    ratio = float(rng.choice([0.6, 0.8, 0.9, 0.9, 1.0]))
    ori = bool(rng.integers(0, 2))
    topm = int(rng.choice([0, 0, 1, 2, 4, 16]))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        a, b = synth.make_frame(w, h, seed=it, step=0), synth.make_frame(w, h, seed=it, step=1)
    elif kind == 1:
        base = rng.integers(0, 256, (h + 8, w + 8), dtype=np.uint8)
        a, b = np.ascontiguousarray(base[:h, :w]), np.ascontiguousarray(base[3:h + 3, 5:w + 5])
    else:
        yy, xx = np.mgrid[0:h, 0:w + 16]
        base = ((((xx // 9) + (yy // 11)) & 1) * 150 + 40 + rng.integers(0, 8, (h, w + 16))).astype(np.uint8)
        a, b = np.ascontiguousarray(base[:, :w]), np.ascontiguousarray(base[:, 9:w + 9])
    try:
        fe = V.FExtractor(nf, 1.2, 8, 20, 7, w, h, max_batch=2, tuning=dict(init_topm=topm) if topm else None)
    except V.VslamError as e:
        print("create refused", (w, h, nf), str(e)[:70])
        continue
    try:
        (k1, d1, _), (k2, d2, _) = [(k.copy(), d.copy(), m) for k, d, m in fe.compute_batch([a, b], (0, 1000))]
        p, c = fe.slot_dev_ptrs(0), fe.slot_dev_ptrs(1)
        m = V.FMatcher(fe, ratio, ori)
        try:
            m.search_init_dev_async([(p[0], p[1], p[2], c[0], c[1], c[2], 0)], window)
            out = m.search_init_dev_wait([len(k1)], want_prev=True)
        except V.VslamError as e:
            print("matcher refused", (w, h, nf, window, ratio), str(e)[:70])
            continue
    finally:
        fe.close()
    wn, wm, wp = orbo.search_for_initialization(k1, d1, k2, d2, w, h, window=window, nnratio=ratio, check_ori=ori)
    ok = out[0][0] == wn and np.array_equal(out[0][1], wm) and np.array_equal(out[0][2], wp)
    ran += 1
    if not ok:
        bad += 1
        print("MISMATCH", dict(w=w, h=h, nf=nf, window=window, ratio=ratio, ori=ori, topm=topm, kind=kind), out[0][0], wn)
print("fuzz init matcher:", ran, "cases run,", bad, "mismatches")
sys.exit(1 if bad else 0)
