#!/usr/bin/env python3
"""In-kernel stamps of k_octree_v3 (diagnostic build only:  make -C vi_slam_amd/csrc EXTRA_HIPFLAGS=-DVSLAM_OCT_STAMPS after
touching vslam_octree_kernel.hip, library copied aside and loaded with VSLAM_FE_LIB=..., and VSLAM_OCT_DBG=1 in the
environment): where the level-0 workgroup of slot 0 spends its time.  Seven stamps: start, keys gathered, fine cells
counted, prefix sums, split passes, owners filled, keys selected.
    octree_stamps.py [W H NF B]   -> microseconds since the kernel's first stamp"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import vi_slam_amd as V  # noqa: E402
from vi_slam_amd import synth  # noqa: E402

W, H, NF, B = (int(a) for a in (sys.argv[1:5] + ["1241", "376", "1000", "4"][len(sys.argv) - 1:]))
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
pitch = (W + 127) & ~127
dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(synth.make_frame(W, H, step=s)).cuda()
ptrs = [dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
for _ in range(3):
    fe.compute_batch_async(ptrs, pitch, (0, 0), to_host=False)
    fe.wait()
out = (C.c_ulonglong * 64)()
L = V.lib()
L.vslam_dbg_octree_stamps.argtypes = [C.c_void_p, C.c_void_p]
rc = L.vslam_dbg_octree_stamps(fe._h, out)
t = [out[i] for i in range(60) if out[i]]
print("rc", rc, "nstamps", len(t), "candidates level 0:", len(fe.candidates(0, 0)))
print([round((x - t[0]) / 100.0, 1) for x in t])
fe.close()
