#!/usr/bin/env python3
"""Random configurations of the grid FAST detector, device vs oracle (bit-exact), including very coarse levels
(1- and 2-pixel cells), wide borders, both cell sizes, every score and tie rule, fractional thresholds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import orbo
from vi_slam_amd import synth
from vi_slam_amd.fastgrid import FASTGPU

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
N = int(os.environ.get("FUZZ_N", "60"))
bad = 0
for it in range(N):
    L1 = int(rng.integers(1, 7))
    L0 = int(rng.integers(0, L1))
    m = 1 << (L1 - 1)
    w = int(rng.integers(2, 20)) * max(m, 8) + (0 if m >= 8 else int(rng.integers(0, 3)) * m)
    h = int(rng.integers(2, 12)) * max(m, 8)
    w, h = max(w, 32), max(h, 32)
    cw, ch = int(rng.choice([32, 64])), int(rng.choice([32, 64]))
    if ch % m:
        ch = 64 if 64 % m == 0 else 32
        if ch % m:
            continue
    hb, vb = int(rng.choice([0, 0, 3, 7, 16, 40])), int(rng.choice([0, 0, 4, 9, 33]))
    thr = float(rng.choice([0.0, 5.0, 10.0, 10.5, 20.0, 33.25, 80.0]))
    arc, score, tie = int(rng.integers(9, 13)), int(rng.integers(0, 3)), int(rng.integers(0, 2))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = synth.make_frame(max(w, 64), max(h, 64), step=it)[:h, :w]
    elif kind == 1:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 2:
        img = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((((xx // 5) + (yy // 7)) & 1) * 180 + 30).astype(np.uint8)
    img = np.ascontiguousarray(img)
    try:
        d = FASTGPU(w, h, cw, ch, L0, L1, hb, vb, thr, arc, score, tie)
    except Exception as e:
        print("create refused", (w, h, cw, ch, L0, L1), str(e)[:60])
        continue
    try:
        got = d.detect(img)
    finally:
        d.close()
    want = orbo.fg_detect(img, (cw, ch), L0, L1, (hb, vb), thr, arc, score, tie)
    ok = all(np.array_equal(a, b) for a, b in zip(got, want))
    if not ok:
        bad += 1
        print("MISMATCH", dict(w=w, h=h, cell=(cw, ch), levels=(L0, L1), border=(hb, vb), thr=thr, arc=arc, score=score, tie=tie,
                               kind=kind), int((got[1] != want[1]).sum()), int((got[0] != want[0]).any(axis=1).sum()))
print("fuzz:", N, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
