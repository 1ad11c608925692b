#!/usr/bin/env python3
"""Per-frame latency at batch 1 (SURVEY.md 8d 'reality check'): host image in -> keypoints/descriptors out.
Every figure is the MEDIAN of 7 windows of 100 calls: single windows are sometimes twice as slow on this pool (a mostly
idle GPU changes its clocks; the slow windows hit any of the variants)."""
import sys
import time

sys.path.insert(0, ".")
import vi_slam_amd as V  # noqa: E402
from vi_slam_amd import synth  # noqa: E402

W, H = 1241, 376


def med(f, windows=7, n=100):
    for _ in range(20):
        f()
    ts = []
    for _ in range(windows):
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 4), round(ts[0], 4), round(ts[-1], 4)


out = {}
for nf in (1000, 2000):
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, W, H, max_batch=2)
    img = synth.make_frame(W, H)
    out["mono_extract_n%d_ms_median_min_max" % nf] = med(lambda: fe.compute(img, (0, 1000)))
    # the same call sequence as the pinned variant below (results read in place from the context's pinned block), from a
    # PAGEABLE image: what the library itself adds for the staging copy, without the Python wrapper's output arrays
    import ctypes as C
    hp = (C.c_void_p * 1)(img.ctypes.data)

    def pageable():
        fe.compute_batch_async(hp, W, (0, 1000), where=V.IMGS_HOST)
        fe.wait()
    out["mono_extract_n%d_pageable_image_in_place_ms_median_min_max" % nf] = med(pageable)
    # the same from a caller-owned PINNED image (a capture driver's DMA buffer): no row copy into the context's staging
    pin = V.PinnedImages(1, H, W, W)
    pin.array[0][:] = img

    def pinned():
        fe.compute_batch_async(pin.ptrs, W, (0, 1000), where=V.IMGS_PINNED)
        fe.wait()
    out["mono_extract_n%d_pinned_image_ms_median_min_max" % nf] = med(pinned)
    pin.close()
    if nf == 2000:
        L, R = synth.make_stereo_pair(W, H)

        def stereo():
            fe.compute_batch([L, R])
            V.ComputeStereoMatches(fe, 0, fe, 1, 386.1448, 718.856)
        out["stereo_frame_n2000_ms_median_min_max"] = med(stereo)
    fe.close()
print(out)
# the same synchronous call from a C++ program (tools/latency_c.cpp): the C ABI without the Python wrapper's array handling
import os, subprocess, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
raw = os.path.join(tmp, "frame.raw")
synth.make_frame(W, H).tofile(raw)
exe = os.path.join(tmp, "latency_c")
pkg = os.path.join(root, "vi_slam_amd")
subprocess.run(["g++", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "latency_c.cpp"), "-o", exe, "-L", pkg,
                "-lvslam_fe", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], check=True)
for nf in (1000, 2000):
    print(subprocess.run([exe, raw, str(W), str(H), str(nf)], capture_output=True, text=True, timeout=120).stdout.strip())
