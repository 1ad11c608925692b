#!/usr/bin/env python3
"""Rounds per pair and queries per round of the SearchForInitialization replay wave (k_si_replay) on consecutive synthetic
KITTI frames and on the reference's hut frames.  usage: replay_stats.py (on a GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vi_slam_amd as V
from vi_slam_amd import synth


def run(tag, frames, nf, w, h):
    B = len(frames)
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    dev = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames]
    torch.cuda.synchronize()
    fe.compute_batch_async([t.data_ptr() for t in dev], w, (0, 1000), to_host=False)
    fe.wait()
    jobs = []
    for s in range(1, B):
        p, c = fe.slot_dev_ptrs(s - 1), fe.slot_dev_ptrs(s)
        jobs.append((p[0], p[1], p[2], c[0], c[1], c[2], 0))
    m = V.FMatcher(fe, 0.9, True)
    m.search_init_replay_stats()
    for rep in range(3):
        m.search_init_dev_async(jobs, 100)
        m.search_init_dev_wait([fe.cap] * len(jobs))
    rounds, queries, pairs = m.search_init_replay_stats()
    t0 = time.perf_counter()
    for rep in range(20):
        m.search_init_dev_async(jobs, 100)
        m.search_init_dev_wait([fe.cap] * len(jobs))
    dt = (time.perf_counter() - t0) / 20
    print("%s: %d pairs, %.1f queries and %.1f rounds per pair = %.2f queries per round; %.1f us per matcher call (host clock)"
          % (tag, len(jobs), queries / pairs, rounds / pairs, queries / max(rounds, 1), dt * 1e6))
    fe.close()


if __name__ == "__main__":
    for nf in (1000, 2000):
        run("kitti synthetic N=%d" % nf, [synth.make_frame(1241, 376, seed=5, step=s) for s in range(32)], nf, 1241, 376)
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_images.npz")
    ims = np.load(g)
    run("hut real N=1200", [ims["hut%d" % i] for i in (1, 2, 3, 4, 5)], 1200, 752, 480)
