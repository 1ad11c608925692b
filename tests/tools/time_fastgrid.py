#!/usr/bin/env python3
"""The grid FAST detector (vilib FASTGPU's job) on the GPU box: a batch of device-resident KITTI-size frames per call vs
the oracle and vs the REFERENCE's own CPU detector (rosten::fast10_detect_nonmax, oracle/_ref) on one host core."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from oracle import orbo
from vi_slam_amd import synth
from vi_slam_amd.fastgrid import FASTGPU

W, H, B = 1240, 376, 32
LEVELS = int(os.environ.get("FG_LEVELS", "1"))  # fast_cuda.cpp:24-27 uses one level; vilib's test_fast.cpp uses more
frames = [np.ascontiguousarray(synth.make_frame(1241, 376, step=s)[:, :W]) for s in range(B)]
dev = torch.zeros((B, H, 1280), dtype=torch.uint8, device="cuda")
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(frames[s]).cuda()
torch.cuda.synchronize()
ptrs = [dev[s].data_ptr() for s in range(B)]
d = FASTGPU(W, H, max_level=LEVELS, max_batch=B)
for _ in range(3):
    out = d.detect_batch(dev_ptrs=ptrs, pitch=1280)
t0 = time.perf_counter()
N = 50
for _ in range(N):
    out = d.detect_batch(dev_ptrs=ptrs, pitch=1280)
t_dev = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(10):
    out_h = d.detect_batch(frames)
t_host = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
want = orbo.fg_detect(frames[0], (32, 32), 0, LEVELS)
t_oracle = time.perf_counter() - t0
t0 = time.perf_counter()
ref = orbo.ref_fast_detect_nonmax(frames[0], 10, 10, False)
t_ref = time.perf_counter() - t0
px = sum((W >> l) * (H >> l) for l in range(LEVELS))
print({"frames_per_call": B, "levels": LEVELS, "cells": d.cells, "occupied_cells_frame0": int((out[1][0] > 0).sum()),
       "ms_per_call_device_inputs": t_dev * 1e3, "frames_per_s_device_inputs": B / t_dev,
       "ms_per_call_host_inputs": t_host * 1e3, "frames_per_s_host_inputs": B / t_host,
       "algorithmic_GBps_device_inputs": px * B / t_dev / 1e9,
       "oracle_1core_ms_per_frame": t_oracle * 1e3,
       "reference_rosten_fast10_level0_1core_ms_per_frame": None if ref is None else t_ref * 1e3,
       "equal_to_oracle": bool(np.array_equal(out[1][0], want[1]) and np.array_equal(out[0][0], want[0]))})
d.close()
