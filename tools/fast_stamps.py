#!/usr/bin/env python3
"""Where a FAST workgroup's wave 0 spends its cycles (diagnostic build only:
   make -C vi_slam_amd/csrc clean all EXTRA_HIPFLAGS=-DVSLAM_FAST_STAMPS).  Sums over all workgroups of a few passes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import vi_slam_amd as V  # noqa: E402
from vi_slam_amd import synth  # noqa: E402

W, H, NF, B = 1241, 376, 1000, 32
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
dev = torch.zeros((B, H, 1280), dtype=torch.uint8, device="cuda")
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(synth.make_frame(W, H, step=s)).cuda()
ptrs = [dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
L = V.lib()
out = (C.c_ulonglong * 16)()
for _ in range(3):
    fe.compute_batch_async(ptrs, 1280, (0, 1000), to_host=False)
    fe.wait()
L.vslam_dbg_fast_stamps(out, 1)
N = 5
for _ in range(N):
    fe.compute_batch_async(ptrs, 1280, (0, 1000), to_host=False)
    fe.wait()
L.vslam_dbg_fast_stamps(out, 0)
names = ["prologue+loads+zero", "barrier", "pretest", "barrier", "compaction", "barrier", "scores", "barrier", "nms", "barrier",
         "keep count+scan", "barrier", "output"]
nwg = 1220 * B * N
tot = sum(out[i] for i in range(13))
for i, nm in enumerate(names):
    print("%-22s %8.0f ticks/WG  %5.1f %%" % (nm, out[i] / nwg, 100.0 * out[i] / max(tot, 1)))
print("total %.0f ticks per workgroup (s_memtime ticks; includes second-stage repeats)" % (tot / nwg))
fe.close()
