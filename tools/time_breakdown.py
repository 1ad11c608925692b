#!/usr/bin/env python3
"""Where does a bench step spend its wall time? (GPU box)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import vi_slam_amd as V
from vi_slam_amd import synth

W, H, NF, B = 1241, 376, 1000, 16
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
pitch = 1280
dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(synth.make_frame(W, H, step=s)).cuda()
ptrs = [dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
m = V.FMatcher(fe, 0.9, True)
for it in range(3):
    t0 = time.perf_counter()
    res = fe.compute_batch(None, (0, 1000), device_ptrs=ptrs, pitch=pitch)
    t1 = time.perf_counter()
    r2 = fe.compute_batch(None, (0, 1000), device_ptrs=ptrs, pitch=pitch, to_host=False)
    t2 = time.perf_counter()
    bufs = [fe.slot_buffers(s) for s in range(B)]
    for s in range(1, B):
        k1, k2 = res[s - 1][0], res[s][0]
        m.SearchForInitialization(k1, bufs[s - 1][1], k2, bufs[s][1], np.stack([k1["x"], k1["y"]], 1), 100)
    t3 = time.perf_counter()
    print("extract(to_host) %.2f ms | extract(device only) %.2f ms | 15 x SearchForInitialization %.2f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
fe.set_profiling(True)
fe.compute_batch(None, (0, 1000), device_ptrs=ptrs, pitch=pitch, to_host=False)
print(fe.get_profile())
