#!/usr/bin/env python3
"""Stage-by-stage parity probe (GPU box): HIP extractor vs the CPU oracle on synthetic frames."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vi_slam_amd as V
from vi_slam_amd import synth
from oracle import orbo

def main():
    W, H, NF = 1241, 376, 2000
    if len(sys.argv) > 3:
        W, H, NF = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    img = synth.make_frame(W, H)
    fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=4)
    t0 = time.time()
    k, d, mono = fe.compute(img)
    print("gpu compute %.1f ms, n=%d mono=%d" % ((time.time() - t0) * 1e3, len(k), mono))
    e = orbo.Extractor(NF)
    ko, do, monoo = e.compute(img)
    print("oracle n=%d mono=%d" % (len(ko), monoo))
    ok = True
    for l in range(8):
        a, b = fe.mvImagePyramid(l), e.level(l)
        same = a.shape == b.shape and np.array_equal(a, b)
        ab, bb = fe.mvImagePyramid(l, blurred=True), e.level(l, blurred=True)
        sameb = bb is not None and np.array_equal(ab, bb)
        ca, cb = fe.candidates(l), e.candidates(l)
        samec = len(ca) == len(cb) and np.array_equal(ca[["x", "y", "response"]], cb[["x", "y", "response"]])
        print("level %d pyramid %s blur %s cand %d/%d %s" % (l, same, sameb, len(ca), len(cb), samec))
        if not same:
            diff = np.argwhere(a != b)
            print("   first diffs", diff[:5], a[tuple(diff[0])], b[tuple(diff[0])])
        ok &= same and sameb and samec
    n = min(len(k), len(ko))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        eq = len(k) == len(ko) and np.array_equal(k[f], ko[f])
        print("kp.%s equal: %s" % (f, eq))
        if not eq and n:
            bad = np.nonzero(k[f][:n] != ko[f][:n])[0]
            print("   mismatches", len(bad), bad[:5], k[f][bad[:5]], ko[f][bad[:5]])
        ok &= eq
    eqd = d.shape == do.shape and np.array_equal(d, do)
    if not eqd and n:
        badrows = np.nonzero((d[:n] != do[:n]).any(1))[0]
        print("   desc rows differing", len(badrows), badrows[:10])
    print("descriptors equal:", eqd)
    ok &= eqd
    # batch path
    imgs = [synth.make_frame(W, H, step=s) for s in range(4)]
    t0 = time.time()
    res = fe.compute_batch(imgs)
    print("batch of 4: %.1f ms" % ((time.time() - t0) * 1e3))
    for s, (kb, db, mb) in enumerate(res):
        ko, do, mo = e.compute(imgs[s])
        same = len(kb) == len(ko) and all(np.array_equal(kb[f], ko[f]) for f in kb.dtype.names) and np.array_equal(db, do)
        print("batch slot %d n=%d parity %s" % (s, len(kb), same))
        ok &= same
    print("ALL OK" if ok else "MISMATCH")
    return 0 if ok else 1

if __name__ == "__main__":
    sys.exit(main())
