#!/usr/bin/env python3
"""digest(tuning): one context with the given per-context switches (vslam_tuning), in the calling process; run as a
script: one process = one setting of the ENVIRONMENT defaults (read once per process; some switches are process-wide).  Runs a batch of four KITTI-size
frames through the extractor from device memory and from pinned host memory via vslam_fe_stage_images_async, then
SearchForInitialization between consecutive frames on the device, and prints one digest of everything delivered.
tests/test_gpu_round2.py compares the digests of several environments with the default one (which other tests
compare with the oracle)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vi_slam_amd as V  # noqa: E402
from vi_slam_amd import synth  # noqa: E402

def digest(tuning=None):
    """one digest of everything the paths below deliver, for a context created with the given vslam_tuning fields"""
    W, H, NF, B = 1241, 376, 1000, 4
    h = hashlib.sha1()
    fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B, tuning=tuning)
    imgs = [synth.make_frame(W, H, seed=77, step=s) for s in range(B)]
    pitch = (W + 127) & ~127
    dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
    for s in range(B):
        dev[s, :, :W] = torch.from_numpy(imgs[s]).cuda()
    torch.cuda.synchronize()
    pin = V.PinnedImages(B, H, W, W)
    for s in range(B):
        pin.array[s][:] = imgs[s]
    for rep in range(2):
        fe.compute_batch_async([dev[s].data_ptr() for s in range(B)], pitch, (0, 1000))
        for k, d, m in fe.wait(copy=True):
            h.update(k.tobytes()); h.update(d.tobytes()); h.update(str(m).encode())
        # staged twice in a row on one context: with VSLAM_STAGE_AHEAD=1 the second upload is issued behind the first pass
        fe.stage_images_async(pin.ptrs, W, V.IMGS_PINNED)
        fe.compute_batch_async(pin.ptrs, W, (0, 1000), where=V.IMGS_STAGED)
        fe.stage_images_async(pin.ptrs, W, V.IMGS_PINNED)
        res = fe.wait(copy=True)
        for k, d, m in res:
            h.update(k.tobytes()); h.update(d.tobytes())
        fe.compute_batch_async(pin.ptrs, W, (0, 1000), where=V.IMGS_STAGED)
        for k, d, m in fe.wait(copy=True):
            h.update(k.tobytes()); h.update(d.tobytes())
        # matcher between consecutive slots, device-resident
        fe.compute_batch_async([dev[s].data_ptr() for s in range(B)], pitch, (0, 1000))
        counts = [len(k) for k, _, _ in fe.wait()]
        M = V.FMatcher(fe, 0.9, True)
        jobs = []
        for s in range(1, B):
            p, q = fe.slot_dev_ptrs(s - 1), fe.slot_dev_ptrs(s)
            jobs.append((p[0], p[1], p[2], q[0], q[1], q[2], 0))
        M.search_init_dev_async(V.FMatcher.make_init_jobs(jobs), 100, (W, H))
        # vnMatches12 has one entry per keypoint of the FIRST frame of a pair: hash exactly those (entries beyond the count
        # are whatever the context's buffer held before)
        for n, m12, prev in M.search_init_dev_wait([counts[s - 1] for s in range(1, B)]):
            h.update(str(n).encode()); h.update(np.asarray(m12).tobytes())
        # the same step with ONE delivery: the extraction's results ride with the matcher's (want_host = 2), pinned images
        # (second rep: the replayed graph)
        fe.compute_batch_async(pin.ptrs, W, (0, 1000), to_host="with_matcher", where=V.IMGS_PINNED)
        M.search_init_dev_async(V.FMatcher.make_init_jobs(jobs), 100, (W, H))
        for k, d, m in fe.wait(copy=True):
            h.update(k.tobytes()); h.update(d.tobytes()); h.update(str(m).encode())
        for n, m12, prev in M.search_init_dev_wait([counts[s - 1] for s in range(1, B)]):
            h.update(str(n).encode()); h.update(np.asarray(m12).tobytes())
    pin.close()
    fe.close()
    return h.hexdigest()


if __name__ == "__main__":
    print("DIGEST", digest())
