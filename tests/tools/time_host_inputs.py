#!/usr/bin/env python3
"""PCIe-inclusive throughput: pageable HOST images in, keypoints + descriptors on the host out, 32 frames per pass,
4 contexts in flight (vslam_fe_extract_batch_async with on_device = 0).  Not bench.py's `value` (inputs resident in
HBM); DESIGN.md quotes this figure beside it."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vi_slam_amd as V
from vi_slam_amd import synth

W, H, NF, B, NCTX, STEPS = 1241, 376, 1000, 32, 4, 120
frames = [np.ascontiguousarray(synth.make_frame(W, H, step=s)) for s in range(B)]
ptrs = (C.c_void_p * B)(*[f.ctypes.data for f in frames])
ctxs = [V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B) for _ in range(NCTX)]
L = V.lib()


def enqueue(c):
    c._pending = (B, True)
    V._check(L.vslam_fe_extract_batch_async(c._h, B, ptrs, W, 0, 0, 1000, 1))


for t in range(NCTX):
    enqueue(ctxs[t])
for t in range(NCTX):
    ctxs[t].wait()
t0 = time.perf_counter()
for t in range(STEPS + NCTX - 1):
    if t < STEPS:
        enqueue(ctxs[t % NCTX])
    if t - (NCTX - 1) >= 0:
        res = ctxs[(t - (NCTX - 1)) % NCTX].wait()
dt = time.perf_counter() - t0
print({"frames_per_step": B, "contexts": NCTX, "steps": STEPS, "ms_per_step": dt / STEPS * 1e3,
       "frames_per_s_host_images_in_results_on_host": B * STEPS / dt, "keypoints_frame0": len(res[0][0])})
for c in ctxs:
    c.close()
