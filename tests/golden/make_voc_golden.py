#!/usr/bin/env python3
"""Generates tests/golden/voc_k8_L3_quicklz.dbow3 and tests/golden/quicklz_packets.npz -- data, not code:
  * the first is vi_slam_amd.synth.make_vocabulary(8, 3, seed=5) written the way DBoW3::Vocabulary::save writes it by
    default (tests/vocfile.py after Vocabulary.cpp:1292-1366) with the chunks compressed by the REFERENCE's QuickLZ
    (oracle/_ref/libref_quicklz.so, built from /root/reference/thirdparty/DBoW3/DBoW3/src/quicklz.c by oracle/Makefile);
  * the second holds byte strings and the packets that compressor makes of them (sizes around the 3/9-byte header
    switch at 216, the 10-byte literal tail, and the 10000-byte chunk size; random, constant, periodic, low-entropy and
    phrase-repeating content), each packet also decoded again by the reference's own decoder.
Needs the reference tree (run in the build container): python tests/golden/make_voc_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import vocfile  # noqa: E402
from vi_slam_amd import synth  # noqa: E402


def samples():
    rng = np.random.default_rng(2024)
    out = []
    for n in (1, 2, 3, 9, 10, 11, 12, 13, 14, 40, 215, 216, 217, 1000, 9999, 10000):
        out.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        out.append(bytes(n))
        out.append((b"abcabcabd" * (n // 9 + 1))[:n])
        out.append(rng.integers(0, 4, n, dtype=np.uint8).tobytes())
        base = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
        out.append(b"".join(base[int(rng.integers(0, 60)):][:int(rng.integers(1, 40))] for _ in range(n))[:n])
    return out


def main():
    R = vocfile.ref_quicklz()
    if R is None:
        raise SystemExit("oracle/_ref/libref_quicklz.so is missing: make -C oracle (needs /root/reference)")
    voc = synth.make_vocabulary(8, 3, seed=5, weighting=0, norm=1)
    vocfile.write_binary(os.path.join(HERE, "voc_k8_L3_quicklz.dbow3"), voc, compressed=True)
    plain, packets = samples(), []
    for d in plain:
        p = vocfile.ref_compress(d, R)
        assert vocfile.ref_decompress(p, R) == d
        packets.append(p)
    cat = lambda xs: np.frombuffer(b"".join(xs), np.uint8)
    off = lambda xs: np.cumsum([0] + [len(x) for x in xs]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "quicklz_packets.npz"), plain=cat(plain), plain_off=off(plain),
                        packets=cat(packets), packets_off=off(packets))
    print("wrote voc_k8_L3_quicklz.dbow3 (%d bytes) and %d packets" %
          (os.path.getsize(os.path.join(HERE, "voc_k8_L3_quicklz.dbow3")), len(packets)))


if __name__ == "__main__":
    main()
