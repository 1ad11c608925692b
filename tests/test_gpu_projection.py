"""GPU (-m gpu): FMatcher::SearchByProjection(CurrentFrame, LastFrame) on the device vs the oracle
(fmatcher.cpp:2471-2687) -- the tracking matcher of TrackWithMotionModel.  Bit-exact match tables."""
import numpy as np
import pytest

import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

pytestmark = pytest.mark.gpu

W, H, NF = 1241, 376, 2000
FX, FY, CX, CY, BF = 718.856, 718.856, 607.1928, 185.2157, 386.1448
MB = BF / FX


@pytest.fixture(scope="module")
def scene():
    """Two consecutive stereo frames: the last frame's MapPoints are its stereo points (depth from
    ComputeStereoMatches), expressed in the last camera (= world)."""
    fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=4)
    L0, R0 = synth.make_stereo_pair(W, H, step=0)
    L1, R1 = synth.make_stereo_pair(W, H, step=1)
    res = fe.compute_batch([L0, R0, L1, R1])
    (u0, d0), (u1, d1) = V.ComputeStereoMatchesBatch(fe, [0, 2], fe, [1, 3], BF, FX)
    k0, de0, _ = res[0]
    k1, de1, _ = res[2]
    z = np.where(d0 > 0, d0, 20.0).astype(np.float32)
    X = np.stack([(k0["x"] - CX) / FX * z, (k0["y"] - CY) / FY * z, z], 1).astype(np.float32)
    yield dict(fe=fe, k0=k0.copy(), de0=de0.copy(), k1=k1.copy(), de1=de1.copy(), u1=u1.copy(), X=X, z=z,
               has_depth=d0 > 0, cur=fe.slot_dev_ptrs(2), sf=fe.GetScaleFactors())
    fe.close()


def _pose(tx=0.0, ty=0.0, tz=0.0, yaw=0.0):
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    return np.hstack([R, np.array([[tx], [ty], [tz]], np.float32)]).astype(np.float32)


def _run(scene, Tcw, th, flags, mono=False, check_ori=True, u_right=True, occupied=None, gemm_float=False):
    Tlw = _pose()
    cam = (FX, FY, CX, CY, BF, MB)
    m = V.FMatcher(scene["fe"], 0.9, check_ori)
    ur = scene["u1"] if u_right else None
    nm, mc, dirs = m.SearchByProjection(Tcw, Tlw, cam, th, scene["k0"], flags, scene["X"], scene["de0"],
                                        scene["cur"][0], scene["cur"][1], len(scene["k1"]), ur, mono, (W, H),
                                        occupied, gemm_float)
    wn, wm, wd = orbo.search_by_projection_frame(
        Tcw, Tlw, cam, th, scene["k0"], flags, scene["X"], scene["de0"], scene["k1"], scene["de1"],
        scene["u1"] if u_right else np.full(len(scene["k1"]), -1, np.float32), scene["sf"], W, H, mono=mono,
        check_ori=check_ori, occupied=occupied, gemm_double=not gemm_float)
    assert dirs == wd
    assert nm == wn, (nm, wn)
    assert np.array_equal(mc, wm)
    return nm, mc, dirs


def test_projection_matches_with_stereo_gate(scene):
    # the synthetic scene shifts by (+3, +1) px per step; at the median depth that is a small sideways translation
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    Tcw = _pose(tx=3.0 / FX * zmed, ty=1.0 / FY * zmed)
    flags = np.where(scene["has_depth"], 3, 0).astype(np.uint8)
    nm, mc, dirs = _run(scene, Tcw, 15, flags)
    assert dirs == (False, False) and nm > 200
    _run(scene, Tcw, 7, flags)
    _run(scene, Tcw, 30, flags)  # the "2*th" retry of tracking.cpp
    _run(scene, Tcw, 15, flags, gemm_float=True)


def test_forward_backward_level_rules_and_mono(scene):
    flags = np.full(len(scene["k0"]), 3, np.uint8)
    nm_f, _, d_f = _run(scene, _pose(tz=-1.0), 15, flags)   # camera moved forward: tlc.z > mb
    nm_b, _, d_b = _run(scene, _pose(tz=1.0), 15, flags)    # backward
    assert d_f == (True, False) and d_b == (False, True)
    nm_m, _, d_m = _run(scene, _pose(tz=-1.0), 15, flags, mono=True, u_right=False)
    assert d_m == (False, False)
    _run(scene, _pose(tx=0.05, yaw=0.01), 15, flags, check_ori=False)


def test_mappoint_flags_observations_and_initial_occupancy(scene):
    rng = np.random.default_rng(7)
    n0, n1 = len(scene["k0"]), len(scene["k1"])
    flags = rng.integers(0, 4, n0).astype(np.uint8)          # no MapPoint / outlier / temporal point (no observations)
    occ = (rng.random(n1) < 0.2).astype(np.uint8)
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    Tcw = _pose(tx=3.0 / FX * zmed, ty=1.0 / FY * zmed)
    nm, mc, _ = _run(scene, Tcw, 15, flags, occupied=occ)
    assert nm > 50
    assert not np.any((mc >= 0) & (occ == 1))                # occupied keypoints are never reassigned
    assert np.all((flags[mc[mc >= 0]] & 1) == 1)
    # every MapPoint without observations: the same current keypoint may be claimed repeatedly (last writer wins)
    _run(scene, Tcw, 30, np.full(n0, 1, np.uint8))


@pytest.mark.parametrize("topm", ["1", "2"])
def test_sorted_prefix_exhaustion_rescans(scene, monkeypatch, topm):
    monkeypatch.setenv("VSLAM_SBP_TOPM", topm)
    flags = np.full(len(scene["k0"]), 3, np.uint8)
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    m = V.FMatcher(scene["fe"], 0.9, True)
    m.search_init_fallbacks()
    _run(scene, _pose(tx=3.0 / FX * zmed, ty=1.0 / FY * zmed), 30, flags)
    assert m.search_init_fallbacks() > 0


def test_degenerate_inputs(scene):
    fe = scene["fe"]
    m = V.FMatcher(fe, 0.9, True)
    cam = (FX, FY, CX, CY, BF, MB)
    nm, mc, _ = m.SearchByProjection(_pose(), _pose(), cam, 15, scene["k0"][:0], np.zeros(0, np.uint8),
                                     np.zeros((0, 3), np.float32), np.zeros((0, 32), np.uint8), scene["cur"][0],
                                     scene["cur"][1], len(scene["k1"]), None, False, (W, H))
    assert nm == 0 and np.all(mc == -1)
    # points behind the camera / outside the image produce nothing
    flags = np.full(len(scene["k0"]), 3, np.uint8)
    nm, mc, _ = _run(scene, _pose(tz=-1000.0), 15, flags)
    assert nm == 0
