#!/usr/bin/env python3
"""Frame::ComputeBoW on the GPU box: tree walk (k=10, L=6: the shape of the ORB vocabulary, synthetic content) for
32 image slots in one launch vs the oracle on one host core."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

W, H, NF, B = 1241, 376, 2000, 32
t0 = time.time()
voc = synth.make_vocabulary(10, 6, seed=4)
print("vocabulary: %d nodes, built in %.1f s" % (len(voc["child_start"]), time.time() - t0), flush=True)
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
res = fe.compute_batch([synth.make_frame(W, H, step=s) for s in range(B)])
res = [(k.copy(), d.copy(), m) for k, d, m in res]
vv = V.Vocabulary(voc)
n = [len(r[0]) for r in res]
for _ in range(3):
    vv.transform_slots_async(fe, 0, B, 4)
    out = vv.transform_slots_wait(n, assemble=False)
t0 = time.perf_counter()
for _ in range(20):
    vv.transform_slots_async(fe, 0, B, 4)
    out = vv.transform_slots_wait(n, assemble=False)
t_walk = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(5):
    vv.transform_slots_async(fe, 0, B, 4)
    out = vv.transform_slots_wait(n, assemble=True)
t_full = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
want = orbo.bow_transform(voc, res[0][1], 4)
t_cpu = time.perf_counter() - t0
ok = all(np.array_equal(out[0][k], want[k]) for k in ("word", "nid", "bow_ids", "bow_vals"))
print({"frames": B, "features_per_frame": n[0], "gpu_walk_ms_per_32_frames": t_walk * 1e3,
       "gpu_walk_plus_host_assembly_ms_per_32_frames": t_full * 1e3, "oracle_1core_ms_per_frame": t_cpu * 1e3,
       "matches_oracle": ok})
vv.close(); fe.close()
