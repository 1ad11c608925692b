"""GPU (-m gpu): the mapping-thread matchers on the device vs the oracle, bit-exact:
FMatcher::SearchForTriangulation (fmatcher.cpp:1242-1482 / :1484-1725) and the search half of FMatcher::Fuse
(fmatcher.cpp:1918-2119, Sim3 overload :2121-2243), plus the glibc logf MapPoint::PredictScale needs."""
import ctypes
import ctypes.util

import numpy as np
import pytest

import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

pytestmark = pytest.mark.gpu

W, H, NF = 1241, 376, 2000
FX, FY, CX, CY, BF = 718.856, 718.856, 607.1928, 185.2157, 386.1448


@pytest.fixture(scope="module")
def scene():
    """Two consecutive stereo frames as KeyFrames: keypoints, descriptors, mvuRight/mvDepth and FeatureVectors."""
    fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=4)
    L0, R0 = synth.make_stereo_pair(W, H, step=0)
    L1, R1 = synth.make_stereo_pair(W, H, step=1)
    res = fe.compute_batch([L0, R0, L1, R1])
    (u0, d0), (u1, d1) = V.ComputeStereoMatchesBatch(fe, [0, 2], fe, [1, 3], BF, FX)
    k0, de0, _ = res[0]
    k1, de1, _ = res[2]
    voc = synth.make_vocabulary(10, 4, seed=31)
    vv = V.Vocabulary(voc)
    vv.transform_slots_async(fe, 0, 3, 2)
    bw = vv.transform_slots_wait([len(res[s][0]) for s in range(3)])
    vv.close()
    yield dict(fe=fe, k0=k0.copy(), de0=de0.copy(), u0=u0.copy(), z0=d0.copy(), k1=k1.copy(), de1=de1.copy(),
               u1=u1.copy(), z1=d1.copy(), fv0=bw[0], fv1=bw[2], dev0=fe.slot_dev_ptrs(0), dev1=fe.slot_dev_ptrs(2),
               sf=fe.GetScaleFactors(), sig2=fe.GetScaleSigmaSquares(), isig2=fe.GetInverseScaleSigmaSquares())
    fe.close()


def test_device_logf_is_glibc_logf(scene):
    """The device restatement, the oracle's and the platform libm agree bit for bit."""
    rng = np.random.default_rng(1)
    x = np.concatenate([
        np.exp(rng.uniform(-20, 20, 200000)), rng.uniform(0.5, 2.0, 200000), 1.0 + rng.uniform(-1e-3, 1e-3, 50000),
        np.array([1.0, 1.2, 0.1, 10.0, 1e-30, 1e30, np.float32(1.17549435e-38), np.float32(3.4028235e38)]),
    ]).astype(np.float32)
    got = V.dbg_logf(scene["fe"], x)
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.logf.restype = ctypes.c_float
    libm.logf.argtypes = [ctypes.c_float]
    idx = rng.integers(0, len(x), 20000)
    want = np.array([libm.logf(float(v)) for v in x[idx]], np.float32)
    assert np.array_equal(got[idx].view(np.uint32), want.view(np.uint32))
    ora = np.array([orbo.logf(float(v)) for v in x[idx]], np.float32)
    assert np.array_equal(ora.view(np.uint32), want.view(np.uint32))
    # outside the domain PredictScale can reach with positive distances the device helper is loud
    bad = V.dbg_logf(scene["fe"], np.array([0.0, -1.0, np.inf, np.nan, 1e-45], np.float32))
    assert np.all(np.isnan(bad))


def _tri(scene, F12, ep, has1, has2, u1, u2, only_stereo=False, coarse=False, ori=True, swap=False):
    a, b = ("0", "1") if not swap else ("1", "0")
    ka, da, fva, deva = scene["k" + a], scene["de" + a], scene["fv" + a], scene["dev" + a]
    kb, db, fvb, devb = scene["k" + b], scene["de" + b], scene["fv" + b], scene["dev" + b]
    m = V.FMatcher(scene["fe"], 0.6, ori)
    nm, pairs, m12 = m.SearchForTriangulation(ka, deva[1], has1, u1, fva, kb, devb[1], has2, u2, fvb, F12, ep,
                                              only_stereo, coarse)
    wn, wm = orbo.search_for_triangulation(ka, da, has1, u1, fva, kb, db, has2, u2, fvb, scene["sf"], scene["sig2"],
                                           F12, ep, only_stereo, coarse, ori)
    assert nm == wn, (nm, wn)
    assert np.array_equal(m12, wm)
    assert len(pairs) == nm and np.array_equal(pairs[:, 1], m12[pairs[:, 0]])
    assert np.all(np.diff(pairs[:, 0]) > 0)
    return nm, m12


def test_search_for_triangulation_equals_oracle(scene):
    n0, n1 = len(scene["k0"]), len(scene["k1"])
    rng = np.random.default_rng(5)
    # the synthetic scene moves by (+3, +1) px per step: epipolar lines are parallel to (3, 1)
    F = np.array([[0, 0, 1], [0, 0, -3], [-1, 3, 0]], np.float32)
    none0, none1 = np.zeros(n0, np.uint8), np.zeros(n1, np.uint8)
    mono0, mono1 = np.full(n0, -1, np.float32), np.full(n1, -1, np.float32)
    nm, m12 = _tri(scene, F, (-5000.0, -5000.0), none0, none1, mono0, mono1)
    assert nm > 300
    # most matches follow the scene motion
    good = m12 >= 0
    dx = scene["k1"]["x"][m12[good]] - scene["k0"]["x"][good]
    assert np.mean(np.abs(dx - 3.0) < 2.5) > 0.7
    # the epipole inside the image removes mono-mono pairs around it; stereo pairs are exempt
    i1 = int(np.nonzero(good)[0][len(dx) // 2])
    ep_hit = (float(scene["k1"]["x"][m12[i1]]) + 1.0, float(scene["k1"]["y"][m12[i1]]))
    nm_ep, m_ep = _tri(scene, F, ep_hit, none0, none1, mono0, mono1)
    assert m_ep[i1] != m12[i1]
    _, m_st = _tri(scene, F, ep_hit, none0, none1, np.where(np.arange(n0) == i1, 50.0, -1.0).astype(np.float32), mono1)
    assert m_st[i1] == m12[i1]
    _tri(scene, F, (600.0, 180.0), none0, none1, scene["u0"], scene["u1"])
    # MapPoints on both sides, bOnlyStereo, bCoarse, no orientation check
    has0 = (rng.random(n0) < 0.4).astype(np.uint8)
    has1 = (rng.random(n1) < 0.4).astype(np.uint8)
    _tri(scene, F, (600.0, 180.0), has0, has1, scene["u0"], scene["u1"])
    _tri(scene, F, (600.0, 180.0), has0, has1, scene["u0"], scene["u1"], only_stereo=True)
    _tri(scene, F, (600.0, 180.0), has0, has1, scene["u0"], scene["u1"], ori=False)
    # a wrong geometry: only bCoarse lets descriptors through
    Fbad = np.array([[0, 0, 3], [0, 0, 1], [-3, -1, 40]], np.float32)
    nb, _ = _tri(scene, Fbad, (-5000.0, -5000.0), none0, none1, mono0, mono1)
    nc, _ = _tri(scene, Fbad, (-5000.0, -5000.0), none0, none1, mono0, mono1, coarse=True)
    assert nb < nc and nc >= nm
    # degenerate line (a = b = 0 -> den == 0 -> never accepted), and the other direction
    nz, _ = _tri(scene, np.zeros((3, 3), np.float32), (0.0, 0.0), none0, none1, mono0, mono1)
    assert nz == 0
    _tri(scene, F.T.copy(), (-5000.0, -5000.0), none1, none0, mono1, mono0, swap=True)


def test_search_for_triangulation_ties_take_the_last_candidate(scene):
    """Identical descriptors under one node: `dist > bestDist` skips, equality replaces -> the last one wins."""
    fe = scene["fe"]
    n = 40
    kps = np.zeros(n, V.KP_DTYPE)
    kps["x"] = 100 + 10 * np.arange(n)
    kps["y"] = 100
    desc = np.tile(np.arange(32, dtype=np.uint8), (n, 1))
    fv = dict(fv_nodes=np.array([7], np.int32), fv_off=np.array([0, n], np.int32), fv_feat=np.arange(n, dtype=np.int32)[::-1].copy())
    import torch
    dd = torch.from_numpy(desc).cuda()
    none, mono = np.zeros(n, np.uint8), np.full(n, -1, np.float32)
    m = V.FMatcher(fe, 0.6, False)
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # horizontal epipolar lines
    nm, pairs, m12 = m.SearchForTriangulation(kps, dd.data_ptr(), none, mono, fv, kps, dd.data_ptr(), none, mono, fv, F,
                                              (-1e4, -1e4))
    wn, wm = orbo.search_for_triangulation(kps, desc, none, mono, fv, kps, desc, none, mono, fv, scene["sf"],
                                           scene["sig2"], F, (-1e4, -1e4), False, False, False)
    assert nm == wn == n and np.array_equal(m12, wm)
    assert np.all(m12 == fv["fv_feat"][-1])  # the last feature of the node's list, for every query


def _fuse_points(scene, rng, valid_p=0.9):
    """MapPoints = frame 0's stereo points in its camera frame (= world), with the scale-invariance range
    MapPoint::UpdateNormalAndDepth would give them (mappoint.cpp: dist * scale[level], / scale[nlevels-1])."""
    k0, z0, sf = scene["k0"], scene["z0"], scene["sf"]
    z = np.where(z0 > 0, z0, 25.0).astype(np.float32)
    X = np.stack([(k0["x"] - CX) / FX * z, (k0["y"] - CY) / FY * z, z], 1).astype(np.float32)
    pts = np.zeros(len(k0), V.FUSE_POINT_DTYPE)
    pts["pos"] = X
    d = np.linalg.norm(X, axis=1).astype(np.float32)
    pts["normal"] = X / d[:, None]
    pts["max_distance"] = 1.2 * d * sf[k0["octave"]]
    pts["min_distance"] = 0.8 * d * sf[k0["octave"]] / sf[-1]
    pts["valid"] = (rng.random(len(k0)) < valid_p).astype(np.int32)
    return pts


def _fuse(scene, pts, desc, Rcw, tcw, Ow, th, sim3=False, gemm_float=False, u_right=True):
    m = V.FMatcher(scene["fe"], 0.6, True)
    lsf = float(np.log(np.float32(1.2)).astype(np.float32))
    cam = (FX, FY, CX, CY, BF)
    ur = scene["u1"] if u_right else None
    bi, bd = m.FuseSearch(pts, desc, scene["dev1"][0], scene["dev1"][1], len(scene["k1"]), ur, Rcw, tcw, Ow, cam, th, lsf,
                          (W, H), sim3, gemm_float)
    wi, wd = orbo.fuse_search(pts, desc, scene["k1"], scene["de1"],
                              scene["u1"] if u_right else np.full(len(scene["k1"]), -1, np.float32), scene["sf"],
                              scene["isig2"], Rcw, tcw, Ow, cam, th, lsf, W, H, sim3, not gemm_float)
    assert np.array_equal(bi, wi)
    assert np.array_equal(bd, np.minimum(wd, np.where(wi >= 0, 255, 256)))
    return bi, bd


def test_fuse_search_equals_oracle(scene):
    rng = np.random.default_rng(8)
    pts = _fuse_points(scene, rng)
    zmed = float(np.median(scene["z0"][scene["z0"] > 0]))
    R = np.eye(3, dtype=np.float32)
    t = np.array([3.0 / FX * zmed, 1.0 / FY * zmed, 0.0], np.float32)  # the scene's (+3, +1) px per step
    Ow = (-R.T @ t).astype(np.float32)
    bi, bd = _fuse(scene, pts, scene["de0"], R, t, Ow, 3.0)
    hit = (bi >= 0) & (bd <= 50)
    assert hit.sum() > 300
    assert np.all(bi[pts["valid"] == 0] == -1)
    _fuse(scene, pts, scene["de0"], R, t, Ow, 4.0, sim3=True)
    _fuse(scene, pts, scene["de0"], R, t, Ow, 3.0, gemm_float=True)
    _fuse(scene, pts, scene["de0"], R, t, Ow, 3.0, u_right=False)
    # a rotated, backwards-moving camera: some points behind it, many outside the image, other levels predicted
    c, s = np.cos(0.15), np.sin(0.15)
    R2 = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    t2 = np.array([0.4, -0.1, -12.0], np.float32)
    _fuse(scene, pts, scene["de0"], R2, t2, (-R2.T @ t2).astype(np.float32), 3.0)
    # normals that fail the 60 degree test for half of the points, and ranges that exclude a third of them
    pts2 = pts.copy()
    flip = rng.random(len(pts)) < 0.5
    pts2["normal"][flip] *= -1.0
    far = rng.random(len(pts)) < 0.33
    pts2["max_distance"][far] *= 0.3
    b2, _ = _fuse(scene, pts2, scene["de0"], R, t, Ow, 3.0)
    assert np.all(b2[flip] == -1)
    # degenerate records: zero position at the camera centre, zero ranges, NaN
    pts3 = pts[:8].copy()
    pts3["pos"][0] = 0
    pts3["max_distance"][1] = 0
    pts3["min_distance"][2] = 0
    pts3["pos"][3] = np.nan
    pts3["pos"][4, 2] = 0
    pts3["valid"] = 1
    _fuse(scene, pts3, scene["de0"][:8], np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32), 3.0)


def test_fuse_search_finds_a_keyframes_own_points(scene):
    """Property at full size: a KeyFrame's own stereo points, seen from its own pose with its own descriptors, come
    back as themselves with distance 0."""
    k1, z1, sf = scene["k1"], scene["z1"], scene["sf"]
    sel = np.nonzero(z1 > 0)[0]
    z = z1[sel]
    X = np.stack([(k1["x"][sel] - CX) / FX * z, (k1["y"][sel] - CY) / FY * z, z], 1).astype(np.float32)
    d = np.linalg.norm(X, axis=1).astype(np.float32)
    pts = np.zeros(len(sel), V.FUSE_POINT_DTYPE)
    pts["pos"], pts["normal"], pts["valid"] = X, X / d[:, None], 1
    # ceil() lands one level above the octave (or on the top level): the keypoint's own level passes the gate
    pts["max_distance"] = (d * sf[k1["octave"][sel]] * np.float32(1.03)).astype(np.float32)
    pts["min_distance"] = 0.0
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    bi, bd = _fuse(scene, pts, scene["de1"][sel], I3, z3, z3, 3.0)
    # two keypoints of one window can share a descriptor; the first in window order wins, so compare through distance
    assert np.all(bd[bi >= 0] == 0) and np.mean(bi >= 0) > 0.95
    same = bi == sel
    assert np.mean(same) > 0.9


def test_mapping_matchers_empty_inputs(scene):
    fe = scene["fe"]
    m = V.FMatcher(fe, 0.6, True)
    e_k = np.zeros(0, V.KP_DTYPE)
    e_fv = dict(fv_nodes=np.zeros(0, np.int32), fv_off=np.zeros(1, np.int32), fv_feat=np.zeros(0, np.int32))
    F = np.eye(3, dtype=np.float32)
    nm, pairs, m12 = m.SearchForTriangulation(e_k, scene["dev0"][1], np.zeros(0, np.uint8), np.zeros(0, np.float32), e_fv,
                                              scene["k1"], scene["dev1"][1], np.zeros(len(scene["k1"]), np.uint8),
                                              scene["u1"], scene["fv1"], F, (0, 0))
    assert nm == 0 and len(pairs) == 0 and len(m12) == 0
    nm, pairs, m12 = m.SearchForTriangulation(scene["k0"], scene["dev0"][1], np.zeros(len(scene["k0"]), np.uint8),
                                              scene["u0"], scene["fv0"], e_k, scene["dev1"][1], np.zeros(0, np.uint8),
                                              np.zeros(0, np.float32), e_fv, F, (0, 0))
    assert nm == 0 and np.all(m12 == -1)
    bi, bd = m.FuseSearch(np.zeros(0, V.FUSE_POINT_DTYPE), np.zeros((0, 32), np.uint8), scene["dev1"][0], scene["dev1"][1],
                          len(scene["k1"]), scene["u1"], F, np.zeros(3), np.zeros(3), (FX, FY, CX, CY, BF), 3.0, 0.18)
    assert len(bi) == 0
    pts = np.zeros(3, V.FUSE_POINT_DTYPE)
    pts["valid"] = 1
    bi, bd = m.FuseSearch(pts, np.zeros((3, 32), np.uint8), scene["dev1"][0], scene["dev1"][1], 0, None, F, np.zeros(3),
                          np.zeros(3), (FX, FY, CX, CY, BF), 3.0, 0.18)
    assert np.all(bi == -1) and np.all(bd == 256)


def test_search_by_sim3_equals_oracle(scene):
    """FMatcher::SearchBySim3 (fmatcher.cpp:2245-2469): both directions on the device (sim3 = 2 of the Fuse kernel), the
    agreement check on the host; the two KeyFrames are the two stereo frames with their stereo points as MapPoints."""
    rng = np.random.default_rng(33)
    fe, sf = scene["fe"], scene["sf"]
    lsf = float(np.log(np.float32(1.2)).astype(np.float32))
    zmed = float(np.median(scene["z0"][scene["z0"] > 0]))
    # KeyFrame 1 = world; KeyFrame 2 is displaced by the scene motion: T2w = [I | t]
    t2w = np.array([3.0 / FX * zmed, 1.0 / FY * zmed, 0.0], np.float32)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)

    def points(k, zdep, Rw, tw):
        z = np.where(zdep > 0, zdep, 25.0).astype(np.float32)
        Xc = np.stack([(k["x"] - CX) / FX * z, (k["y"] - CY) / FY * z, z], 1).astype(np.float32)
        Xw = ((Xc - tw) @ Rw).astype(np.float32)  # Rw^T (Xc - tw)
        d = np.linalg.norm(Xc, axis=1).astype(np.float32)
        pts = np.zeros(len(k), V.FUSE_POINT_DTYPE)
        pts["pos"] = Xw
        pts["max_distance"] = np.float32(1.2) * d * sf[k["octave"]]
        pts["min_distance"] = np.float32(0.8) * d * sf[k["octave"]] / sf[-1]
        pts["valid"] = (rng.random(len(k)) < 0.85).astype(np.int32)
        return pts

    p1 = points(scene["k0"], scene["z0"], I3, z3)
    p2 = points(scene["k1"], scene["z1"], I3, t2w)
    m = V.FMatcher(fe, 0.75, True)
    for s12, th in ((1.0, 7.5), (1.03, 7.5), (0.97, 10.0)):
        # camera 1 from camera 2: x1 = s12 * R12 * x2 + t12 with R12 = I, t12 = -t2w (up to the scale error under test)
        R12, t12 = I3, (-t2w).astype(np.float32)
        for gf in (False, True):
            nF, m12, (sR21, t21, sR12) = m.SearchBySim3(p1, scene["de0"], scene["dev0"][0], scene["dev0"][1], len(scene["k0"]),
                                                        I3, z3, p2, scene["de1"], scene["dev1"][0], scene["dev1"][1],
                                                        len(scene["k1"]), I3, t2w, s12, R12, t12, th, (FX, FY, CX, CY), lsf,
                                                        (W, H), gf)
            wn, wm = orbo.search_by_sim3(p1["valid"], p1["pos"], p1["min_distance"], p1["max_distance"], scene["de0"],
                                         scene["k0"], I3, z3, p2["valid"], p2["pos"], p2["min_distance"], p2["max_distance"],
                                         scene["de1"], scene["k1"], I3, t2w, sR12, t12, sR21, t21, (FX, FY, CX, CY), th, lsf,
                                         sf, W, H, not gf)
            assert nF == wn and np.array_equal(m12, wm), (s12, th, gf, nF, wn)
        if s12 == 1.0:
            assert nF > 80
            assert np.all(p1["valid"][m12 >= 0] == 1) and np.all(p2["valid"][m12[m12 >= 0]] == 1)
