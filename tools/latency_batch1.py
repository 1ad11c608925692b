#!/usr/bin/env python3
"""Per-frame latency at batch 1 (SURVEY.md 8d 'reality check'): host image in -> keypoints/descriptors out."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import vi_slam_amd as V
from vi_slam_amd import synth

W, H = 1241, 376
out = {}
for nf in (1000, 2000):
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, W, H, max_batch=2)
    img = synth.make_frame(W, H)
    for _ in range(20):
        fe.compute(img, (0, 1000))
    t0 = time.perf_counter()
    N = 300
    for _ in range(N):
        fe.compute(img, (0, 1000))
    out["mono_extract_n%d_ms" % nf] = (time.perf_counter() - t0) / N * 1e3
    if nf == 2000:
        L, R = synth.make_stereo_pair(W, H)
        for _ in range(10):
            fe.compute_batch([L, R])
            V.ComputeStereoMatches(fe, 0, fe, 1, 386.1448, 718.856)
        t0 = time.perf_counter()
        for _ in range(N):
            fe.compute_batch([L, R])
            V.ComputeStereoMatches(fe, 0, fe, 1, 386.1448, 718.856)
        out["stereo_frame_n2000_ms"] = (time.perf_counter() - t0) / N * 1e3
    fe.close()
print(out)
