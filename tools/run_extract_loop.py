#!/usr/bin/env python3
"""Run the batched extraction a few times on HBM-resident frames (target for rocprofv3 --pmc passes)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
import vi_slam_amd as V
from vi_slam_amd import synth

stereo = len(sys.argv) > 1 and sys.argv[1] == "stereo"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
NF = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
W = int(sys.argv[5]) if len(sys.argv) > 5 else 1241
H = int(sys.argv[6]) if len(sys.argv) > 6 else 376
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
pitch = (W + 127) & ~127
dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(synth.make_frame(W, H, step=s // 2, right=bool(s & 1))).cuda()
ptrs = [dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
for _ in range(iters):
    if stereo:
        fe.frame_stereo_async(ptrs, pitch, 386.1448, 718.856, to_host=False)
        fe.frame_stereo_wait()
    else:
        fe.compute_batch_async(ptrs, pitch, (0, 0), to_host=False)
        fe.wait()
fe.close()
