"""include/vslam_shim.hpp: the reference's class names over the C ABI.  CPU: the header compiles and links
against libvslam_fe.so.  GPU: a C++ program using FExtractor / FMatcher / ComputeStereoMatches produces exactly
what the ctypes path (already pinned to the oracle) produces on the same frames."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vi_slam_amd")


def _build(tmp_path):
    exe = str(tmp_path / "shim_demo")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"), "-o", exe, "-L", PKG, "-lvslam_fe",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).view(np.uint8).ravel().tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_shim_header_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    # usage error path only: no GPU call is made
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_shim_program_equals_ctypes_path(tmp_path):
    import vi_slam_amd as V
    from vi_slam_amd import synth
    W, H, NF = 640, 360, 1200
    a, b = synth.make_frame(W, H, step=0), synth.make_frame(W, H, step=1)
    L, R = synth.make_stereo_pair(W, H, step=2)
    paths = []
    for name, im in (("a", a), ("b", b), ("l", L), ("r", R)):
        p = str(tmp_path / (name + ".raw"))
        im.tofile(p)
        paths.append(p)
    exe = _build(tmp_path)
    r = subprocess.run([exe, str(W), str(H)] + paths + [str(NF)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])

    f1 = V.FExtractor(NF, 1.2, 8, 20, 7, W, H)
    f2 = V.FExtractor(NF, 1.2, 8, 20, 7, W, H)
    try:
        k1, d1, mono1 = f1.compute(a, (0, 1000))
        k2, d2, mono2 = f2.compute(b, (0, 1000))
        _, pd1, _ = f1.slot_buffers(0)
        _, pd2, _ = f2.slot_buffers(0)
        nm, m12, pm = V.FMatcher(f2, 0.9, True).SearchForInitialization(k1, pd1, k2, pd2,
                                                                       np.stack([k1["x"], k1["y"]], 1), 100)
        z = np.float32(8.0)
        xw = np.stack([(k1["x"] - np.float32(W / 2)) * z * (np.float32(1) / np.float32(500)),
                       (k1["y"] - np.float32(H / 2)) * z * (np.float32(1) / np.float32(500)),
                       np.full(len(k1), z, np.float32)], 1).astype(np.float32)
        Tcw = np.array([[1, 0, 0, np.float32(3.0) / np.float32(500.0) * np.float32(8.0)],
                        [0, 1, 0, np.float32(1.0) / np.float32(500.0) * np.float32(8.0)], [0, 0, 1, 0]], np.float32)
        T0 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
        ck, cd, _ = f2.slot_dev_ptrs(0)
        nsbp, msbp, _ = V.FMatcher(f2, 0.9, True).SearchByProjection(
            Tcw, T0, (500.0, 500.0, W / 2, H / 2, 40.0, 0.08), 15, k1, np.full(len(k1), 3, np.uint8), xw, d1, ck, cd,
            len(k2), None, True, (W, H))
        kL, dL, _ = f1.compute(L)
        kR, dR, _ = f2.compute(R)
        u, dep = V.ComputeStereoMatches(f1, 0, f2, 0, 386.1448, 718.856)
        fish, _, _, fish_n = V.FMatcher(f1).ComputeStereoFishEyeCandidates(f1.slot_buffers(0)[1], len(kL), 0,
                                                                             f2.slot_buffers(0)[1], len(kR), 0)
        lvl3 = f1.mvImagePyramid(3)
        from vi_slam_amd.fastgrid import FASTGPU
        aw = np.ascontiguousarray(a[:H & ~3, :W & ~3])
        g1 = FASTGPU(W & ~3, H & ~3)
        g3 = FASTGPU(W & ~3, H & ~3, max_level=3)
        try:
            p1, s1, l1 = g1.detect(aw)
            p3, s3, l3 = g3.detect(aw)
        finally:
            g1.close()
            g3.close()
    finally:
        f1.close()
        f2.close()
    occ = np.nonzero(s1 > 0)[0]
    kf = np.zeros(len(occ), V.KP_DTYPE)
    kf["x"], kf["y"], kf["size"], kf["angle"], kf["response"], kf["octave"], kf["class_id"] = (
        p1[occ, 0], p1[occ, 1], 7.0, -1.0, s1[occ], l1[occ], -1)
    assert got["fast_n"] == len(occ) and len(occ) > 50 and got["fast_kp"] == _fnv(kf)
    occ3 = np.nonzero(s3 > 0)[0]
    flat = np.stack([p3[occ3, 0], p3[occ3, 1], s3[occ3], l3[occ3]], 1).astype(np.float64)
    assert got["fast3_n"] == len(occ3) and got["fast3"] == _fnv(flat)
    # the C ABI section: STAGED passes with imgs == NULL (capture, then graph replay) under a per-context tuning equal the
    # plain pass; the shallow grid made the quadtree split below it; the fisheye candidates equal the ctypes path
    assert got["staged_ok"] == 1 and got["oct_problems"] == 16 and got["oct_deep"] >= 2
    assert got["fish_n"] == fish_n and fish_n > 50 and got["fish"] == _fnv(fish)
    assert got["n1"] == len(k1) and got["n2"] == len(k2) and got["mono1"] == mono1 and got["mono2"] == mono2
    assert got["rc_empty"] == -1
    assert got["kp1"] == _fnv(k1) and got["desc1"] == _fnv(d1) and got["kp2"] == _fnv(k2) and got["desc2"] == _fnv(d2)
    assert got["nmatches"] == nm and nm > 30 and got["m12"] == _fnv(m12) and got["prev"] == _fnv(pm)
    assert got["dd01"] == int(np.unpackbits(d1[0] ^ d1[1]).sum())
    assert got["nL"] == len(kL) and got["nR"] == len(kR)
    assert got["nstereo"] == int((u >= 0).sum()) and got["nstereo"] > 100
    assert got["uR"] == _fnv(u) and got["depth"] == _fnv(dep)
    assert got["lvl3"] == [lvl3.shape[1], lvl3.shape[0], _fnv(lvl3)]
    assert got["nsbp"] == nsbp and nsbp > 50 and got["sbp"] == _fnv(msbp)
    assert got["levels"] == 8 and abs(got["sf7"] - 3.5831816196) < 1e-6  # mvScaleFactor[7], SURVEY.md 8
