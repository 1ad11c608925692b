#!/usr/bin/env python3
"""Host time of the bench's per-step calls when the GPU is NOT the one being waited for: enqueue mono steps, let the GPU
finish (device synchronize), then time collect(); and the enqueue alone.  If enqueue + collect is close to the step time
of the pipelined run, the host thread is the bottleneck, not the GPU."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402

args = argparse.Namespace(batch=32, inflight=4, force_collective=False)
env = {"rank": 0, "world": 1, "local_rank": 0}
name = sys.argv[1] if len(sys.argv) > 1 else "kitti00_mono_1241x376_n1000"
P = bench.Pipeline(name, args, env)
P.run(8)
torch.cuda.synchronize()
N = P.NCTX
te = tc = 0.0
R = 50
t = 8
for r in range(R):
    a = time.perf_counter()
    for i in range(N):
        P.enqueue(t + i)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    for i in range(N):
        P.collect(t + i)
    d = time.perf_counter()
    te += b - a
    tc += d - c
    t += N
print("%s: enqueue %.1f us per step, collect with the GPU already idle %.1f us per step" % (name, te / (R * N) * 1e6, tc / (R * N) * 1e6))
