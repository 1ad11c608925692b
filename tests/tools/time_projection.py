#!/usr/bin/env python3
"""Latency of one SearchByProjection(CurrentFrame, LastFrame) call on the GPU box vs the oracle on one host core
(KITTI-size frames, 2000 features, stereo points as MapPoints)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

W, H, NF = 1241, 376, 2000
FX, FY, CX, CY, BF = 718.856, 718.856, 607.1928, 185.2157, 386.1448
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=4)
L0, R0 = synth.make_stereo_pair(W, H, step=0)
L1, R1 = synth.make_stereo_pair(W, H, step=1)
res = fe.compute_batch([L0, R0, L1, R1])
(u0, d0), (u1, d1) = V.ComputeStereoMatchesBatch(fe, [0, 2], fe, [1, 3], BF, FX)
k0, de0, _ = res[0]
k1, de1, _ = res[2]
k0, de0, k1, de1 = k0.copy(), de0.copy(), k1.copy(), de1.copy()
z = np.where(d0 > 0, d0, 20.0).astype(np.float32)
X = np.stack([(k0["x"] - CX) / FX * z, (k0["y"] - CY) / FY * z, z], 1).astype(np.float32)
zmed = float(np.median(z[d0 > 0]))
Tcw = np.hstack([np.eye(3), np.array([[3.0 / FX * zmed], [1.0 / FY * zmed], [0.0]])]).astype(np.float32)
Tlw = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
flags = np.where(d0 > 0, 3, 0).astype(np.uint8)
cam = (FX, FY, CX, CY, BF, BF / FX)
cur = fe.slot_dev_ptrs(2)
m = V.FMatcher(fe, 0.9, True)
for _ in range(5):
    nm, mc, _d = m.SearchByProjection(Tcw, Tlw, cam, 15, k0, flags, X, de0, cur[0], cur[1], len(k1), u1, False, (W, H))
t0 = time.perf_counter()
N = 200
for _ in range(N):
    nm, mc, _d = m.SearchByProjection(Tcw, Tlw, cam, 15, k0, flags, X, de0, cur[0], cur[1], len(k1), u1, False, (W, H))
t_gpu = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(20):
    wn, wm, _d = orbo.search_by_projection_frame(Tcw, Tlw, cam, 15, k0, flags, X, de0, k1, de1, u1, fe.GetScaleFactors(), W, H)
t_cpu = (time.perf_counter() - t0) / 20
print({"n_last": len(k0), "n_cur": len(k1), "map_points": int((flags & 1).sum()), "nmatches": nm, "oracle_nmatches": wn,
       "gpu_call_ms": t_gpu * 1e3, "oracle_1core_ms": t_cpu * 1e3})
fe.close()
