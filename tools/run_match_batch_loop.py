#!/usr/bin/env python3
"""The batched brute-force matcher at a size where it can be ALU-bound (target for rocprofv3 passes):
    run_match_batch_loop.py [P N iters]   P independent N x N problems per launch (default 16 x 2000 x 2000: the stereo
pairs of a step through the knnMatch site, frame.cpp:1167-1174)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vi_slam_amd as V  # noqa: E402

P, N, iters = (int(a) for a in (sys.argv[1:4] + ["16", "2000", "20"][len(sys.argv) - 1:]))
fe = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
m = V.FMatcher(fe)
rng = np.random.default_rng(1)
bufs, probs = [], []
for p in range(P):
    q = torch.from_numpy(rng.integers(0, 256, (N, 32), dtype=np.uint8)).cuda()
    t = torch.from_numpy(rng.integers(0, 256, (N, 32), dtype=np.uint8)).cuda()
    bufs.append((q, t))
    probs.append((q.data_ptr(), N, t.data_ptr(), N))
torch.cuda.synchronize()
m.hamming_top2_batch(probs)  # allocates, one synchronous pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    m.hamming_top2_batch_async(probs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
words = P * N * N * 8.0
print("P=%d N=%d: %.1f us per launch pair (host clock, %d back to back), %.2f T xor+popcount words/s" % (P, N, dt * 1e6, iters, words / dt / 1e12))
fe.close()
