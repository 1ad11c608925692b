#!/usr/bin/env python3
"""PCIe probe: H2D rate of (a) one hipMemcpyAsync of a pinned block (SDMA), (b) the library's pull kernel reading pinned
memory, (c) D2H of a pinned block; plus the host-side cost of one async copy call."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import vi_slam_amd as V  # noqa: E402

out = {}
n = 32 * 376 * 1241
a = torch.empty(n, dtype=torch.uint8).pin_memory()
b = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for name, src, dst in (("h2d_memcpy_async", a, b), ("d2h_memcpy_async", b, a)):
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = 0.0
    for _ in range(20):
        t1 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        th += time.perf_counter() - t1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out[name] = {"GBps": round(n / dt / 1e9, 2), "ms": round(dt * 1e3, 4), "host_call_us": round(th / 20 * 1e6, 1)}
# the pull kernel through the library: extraction with pinned inputs vs device inputs, pyramid-only cost difference
fe = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=32)
pin = V.PinnedImages(32, 376, 1241, 1241)
pin.array[:] = 128
for where, ptrs, pitch in ((V.IMGS_PINNED, pin.ptrs, 1241),):
    for _ in range(5):
        fe.compute_batch_async(ptrs, pitch, (0, 1000), to_host=False, where=where)
        fe.wait()
    t0 = time.perf_counter()
    for _ in range(20):
        fe.compute_batch_async(ptrs, pitch, (0, 1000), to_host=False, where=where)
        fe.wait()
    out["extract_pinned_ms_per_pass"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
dev = torch.zeros((32, 376, 1280), dtype=torch.uint8, device="cuda")
dptrs = [dev[s].data_ptr() for s in range(32)]
for _ in range(5):
    fe.compute_batch_async(dptrs, 1280, (0, 1000), to_host=False)
    fe.wait()
t0 = time.perf_counter()
for _ in range(20):
    fe.compute_batch_async(dptrs, 1280, (0, 1000), to_host=False)
    fe.wait()
out["extract_device_ms_per_pass"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
out["pull_GBps_estimate"] = round(n / ((out["extract_pinned_ms_per_pass"] - out["extract_device_ms_per_pass"]) * 1e-3) / 1e9, 2)
print(json.dumps(out))
