#!/usr/bin/env python3
"""Run the matcher kernels a few times (target for rocprofv3 --pmc passes): stereo L<->R on 8 pairs, the mono
initialisation matcher on 15 pairs, and the brute-force top-2 on one pair."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
import vi_slam_amd as V
from vi_slam_amd import synth

W, H, NF, B = 1241, 376, 2000, 16
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
pitch = 1280
dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(synth.make_frame(W, H, step=s // 2, right=bool(s & 1))).cuda()
ptrs = [dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
m = V.FMatcher(fe, 0.9, True)
for _ in range(5):
    fe.frame_stereo_async(ptrs, pitch, 386.1448, 718.856, to_host=False)
    fe.frame_stereo_wait()
    jobs = []
    for s in range(2, B, 2):
        p, c = fe.slot_dev_ptrs(s - 2), fe.slot_dev_ptrs(s)
        jobs.append((p[0], p[1], p[2], c[0], c[1], c[2], 0))
    m.search_init_dev_async(jobs, 100)
    m.search_init_dev_wait([NF] * len(jobs))
    _, d0, n0 = fe.slot_buffers(0)
    _, d1, n1 = fe.slot_buffers(1)
    m.hamming_top2(d0, n0, d1, n1)
fe.close()
