#!/usr/bin/env python3
"""Dynamic block counts of k_fast_bands (diagnostic build: make EXTRA_HIPFLAGS=-DVSLAM_FAST_COUNT):
    fast_band_counts.py [W H NF B]   -> executions per wave of every instrumented block, per launch of B images"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import vi_slam_amd as V  # noqa: E402
from vi_slam_amd import synth  # noqa: E402

W, H, NF, B = (int(a) for a in (sys.argv[1:5] + ["1241", "376", "1000", "32"][len(sys.argv) - 1:]))
NAMES = ["waves", "sweep_iter", "sweep_store", "net_dark", "unused", "net_bright", "nms_call", "nms_inner", "out_pass",
         "out_bit_iter", "stage2", "chunks", "net_loop_iter", "nD_sum", "nB_sum"]
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B, tuning=dict(fast_kernel=4))
pitch = (W + 127) & ~127
dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(synth.make_frame(W, H, step=s)).cuda()
ptrs = [dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
L = V.lib()
out = (C.c_ulonglong * 32)()
fe.compute_batch_async(ptrs, pitch, (0, 1000), to_host=False)
fe.wait()
L.vslam_dbg_fast_band_counts(out, 1)
fe.compute_batch_async(ptrs, pitch, (0, 1000), to_host=False)
fe.wait()
L.vslam_dbg_fast_band_counts(out, 1)
d = {n: int(out[i]) for i, n in enumerate(NAMES)}
w = max(d["waves"], 1)
print(json.dumps({"per_launch": d, "per_wave": {k: round(v / w, 3) for k, v in d.items()}}))
fe.close()
