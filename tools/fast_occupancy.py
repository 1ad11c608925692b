#!/usr/bin/env python3
"""Workgroups of k_fast_cells_v3 in flight over time (diagnostic build only:
   make -C vi_slam_amd/csrc clean all EXTRA_HIPFLAGS=-DVSLAM_FAST_WGREC): every workgroup of the last launch records
   s_memtime at entry/exit, s_memrealtime at entry and its HW_ID; this prints the shader clock, the kernel's span, the
   workgroup lifetime distribution and the average number of workgroups resident per CU."""
import ctypes as C
import os
import sys

import numpy as np

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
for _ in range(4):
    fe.compute_batch_async(ptrs, 1280, (0, 1000), to_host=False)
    fe.wait()
n = 65536
buf = (C.c_ulonglong * (4 * n))()
L.vslam_dbg_fast_wg_records.argtypes = [C.c_void_p, C.c_int]
got = L.vslam_dbg_fast_wg_records(buf, n)
r = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)[:got]
r = r[r[:, 0] != 0]
t0, t1, rt, hw = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 2].astype(np.int64), r[:, 3]
hwid = (hw & np.uint64(0xFFFF)).astype(np.int64)
xcc = ((hw >> np.uint64(16)) & np.uint64(0xF)).astype(np.int64)
dreal = (hw >> np.uint64(32)).astype(np.int64)         # 10-ns ticks
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
simd = (hwid >> 4) & 3
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ncu = len(np.unique(cu_key))
print("workgroups recorded: %d, distinct CUs %d" % (len(r), ncu))
# every XCC has its own counters: times relative to the XCC's first entry
e = np.zeros(len(r)); x = np.zeros(len(r))
for k in np.unique(xcc):
    m = xcc == k
    e[m] = (rt[m] - rt[m].min()) * 0.01          # us
    x[m] = e[m] + dreal[m] * 0.01
    tick_rate = (t1[m] - t0[m]).sum() / max(dreal[m].sum() * 0.01, 1e-9)   # s_memtime ticks per us
    print("XCC %d: %5d workgroups, span %.1f us, s_memtime %.0f ticks/us" % (k, m.sum(), x[m].max(), tick_rate))
life = x - e
span = x.max()
print("workgroup life (us): mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
print("average workgroups in flight per CU: %.2f (sum of lives / span / CUs)" % (life.sum() / span / ncu))
edges = np.linspace(0, span, 41)
prof = []
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(x, b) - np.maximum(e, a), 0, None).sum() / (b - a)
    prof.append(ov / ncu)
print("in flight per CU over the span (40 bins):", " ".join("%.1f" % v for v in prof))
# concurrency inside single CUs: +1 at entry, -1 at exit, sampled over the steady part (first 60 % of the span)
conc_hist = np.zeros(40)
gaps = []
for k in np.unique(cu_key)[::8]:
    m = cu_key == k
    ev = np.concatenate([np.stack([e[m], np.ones(m.sum())], 1), np.stack([x[m], -np.ones(m.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    c = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    ok = ev[:-1, 0] < 0.6 * span
    np.add.at(conc_hist, c[:-1][ok].astype(int), dt[ok])
    # time from an exit to the next entry on this CU (steady part)
    ex = np.sort(x[m]); en = np.sort(e[m])
    for tt in ex[ex < 0.6 * span]:
        j = np.searchsorted(en, tt)
        if j < len(en):
            gaps.append(en[j] - tt)
conc_hist /= conc_hist.sum()
print("time share by workgroups resident on a CU (steady part):", " ".join("%d:%.0f%%" % (i, 100 * v) for i, v in enumerate(conc_hist) if v > 0.005))
gaps = np.array(gaps)
print("exit -> next entry on the same CU (us): mean %.2f  p50 %.2f  p90 %.2f" % (gaps.mean(), *np.percentile(gaps, [50, 90])))
per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
print("workgroups per CU: min %d  mean %.1f  max %d" % (per_cu.min(), per_cu.mean(), per_cu.max()))
print("per SIMD of wave 0:", np.bincount(simd).tolist())
fe.close()
