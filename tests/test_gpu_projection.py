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


@pytest.fixture
def tune(scene):
    """per-call switches of the shared context (vslam_fe_set_tuning), restored to the defaults afterwards"""
    yield scene["fe"].set_tuning
    scene["fe"].set_tuning(sbp_sequential=0, sbp_topm=8)


def test_sequential_replay_path_agrees(scene, tune):
    """sbp_sequential skips the parallel (deferred-acceptance) resolution: one wave walks the queries in order."""
    tune(sbp_sequential=1)
    rng = np.random.default_rng(3)
    flags = rng.integers(0, 4, len(scene["k0"])).astype(np.uint8)
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    _run(scene, _pose(tx=3.0 / FX * zmed, ty=1.0 / FY * zmed), 30, flags)
    _run(scene, _pose(tz=-1.0), 15, np.full(len(scene["k0"]), 3, np.uint8))


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
def test_sorted_prefix_exhaustion_rescans(scene, tune, topm):
    tune(sbp_topm=int(topm))
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


def _dev_read(ptr, nbytes):
    """Copy raw device memory to a numpy byte array (tests only)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.zeros(nbytes, np.uint8)
    assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), nbytes, 2) == 0  # hipMemcpyDeviceToHost
    return out


def test_device_resident_tracking_chain_equals_oracle():
    """Frame::Frame(stereo) x 4 -> UnprojectStereo of every frame -> SearchByProjection(frame s, frame s-1) for
    s = 1..3, all enqueued without touching the host: one extraction+stereo enqueue, one unprojection kernel, one
    pass of the two matcher kernels.  Compared with the oracle chain."""
    import torch
    nst = 4
    fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=2 * nst)
    try:
        frames = [synth.make_stereo_pair(W, H, step=s) for s in range(nst)]
        pitch = 1280
        dev = torch.zeros((2 * nst, H, pitch), dtype=torch.uint8, device="cuda")
        for s in range(nst):
            dev[2 * s, :, :W] = torch.from_numpy(frames[s][0]).cuda()
            dev[2 * s + 1, :, :W] = torch.from_numpy(frames[s][1]).cuda()
        torch.cuda.synchronize()
        fe.frame_stereo_async([dev[i].data_ptr() for i in range(2 * nst)], pitch, BF, FX)
        # world = camera of each frame's own pose; poses: frame s sits at s * (dx, dy, 0)
        zmed = 12.0
        dx, dy = 3.0 / FX * zmed, 1.0 / FY * zmed
        Twc = [np.hstack([np.eye(3), np.array([[-s * dx], [-s * dy], [0.0]])]).astype(np.float32) for s in range(nst)]
        Tcw = [np.hstack([np.eye(3), np.array([[s * dx], [s * dy], [0.0]])]).astype(np.float32) for s in range(nst)]
        invfx, invfy = np.float32(1.0) / np.float32(FX), np.float32(1.0) / np.float32(FY)
        fe.stereo_points_async(Twc, (CX, CY, float(invfx), float(invfy)), observations=True)
        m = V.FMatcher(fe, 0.9, True)
        jobs = []
        for s in range(1, nst):
            lk, ld, ln = fe.slot_dev_ptrs(2 * (s - 1))
            ck, cd, cn = fe.slot_dev_ptrs(2 * s)
            x, f, _, _ = fe.stereo_points_buffers(s - 1)
            _, _, ur, _ = fe.stereo_points_buffers(s)
            fwd, bwd = orbo.search_by_projection_frame(Tcw[s], Tcw[s - 1], (FX, FY, CX, CY, BF, MB), 15,
                                                       np.zeros(0, V.KP_DTYPE), np.zeros(0, np.uint8),
                                                       np.zeros((0, 3), np.float32), np.zeros((0, 32), np.uint8),
                                                       np.zeros(0, V.KP_DTYPE), np.zeros((0, 32), np.uint8),
                                                       np.zeros(0, np.float32), fe.GetScaleFactors(), W, H)[2]
            jobs.append(dict(Tcw=Tcw[s], cam=(FX, FY, CX, CY, BF), th=15, forward=fwd, backward=bwd, img=(W, H),
                             last_kps=lk, n_last=ln, last_flags=f, last_x3dw=x, mp_desc=ld, cur_kps=ck, cur_desc=cd,
                             n_cur=cn, cur_u_right=ur))
        m.search_by_projection_dev_async(jobs)
        feats, st = fe.frame_stereo_wait()
        feats = [(k.copy(), d.copy()) for k, d in feats]
        st = [(u.copy(), d.copy()) for u, d in st]
        out = m.search_by_projection_dev_wait([len(feats[2 * s][0]) for s in range(1, nst)])
        cap = fe.cap
        for s in range(nst):
            kL = feats[2 * s][0]
            wx, wf = orbo.unproject_stereo(kL, st[s][1], Twc[s], CX, CY, float(invfx), float(invfy))
            x, f, _, _ = fe.stereo_points_buffers(s)
            gx = _dev_read(x, cap * 12).view(np.float32).reshape(cap, 3)[:len(kL)]
            gf = _dev_read(f, cap)[:len(kL)]
            assert np.array_equal(gf, wf * 3) and np.array_equal(gx[wf > 0], wx[wf > 0]), s
        for j, s in enumerate(range(1, nst)):
            k0, d0 = feats[2 * (s - 1)]
            k1, d1 = feats[2 * s]
            wx, wf = orbo.unproject_stereo(k0, st[s - 1][1], Twc[s - 1], CX, CY, float(invfx), float(invfy))
            wn, wm, _ = orbo.search_by_projection_frame(Tcw[s], Tcw[s - 1], (FX, FY, CX, CY, BF, MB), 15, k0, wf * 3, wx,
                                                        d0, k1, d1, st[s][0], fe.GetScaleFactors(), W, H)
            assert out[j][0] == wn and np.array_equal(out[j][1], wm), s
            assert wn > 100
    finally:
        fe.close()


# ---------------------------------------------------------------- SearchByProjection(F, vpMapPoints) (local map)
def _local_map(scene, seed, jitter=2.0, frac_in_view=0.85):
    """MapPoints as Tracking::SearchLocalPoints hands them over: the last frame's stereo points seen from the
    current pose, with what Frame::isInFrustum would have stored (projection, predicted level, viewing cosine)."""
    rng = np.random.default_rng(seed)
    k0 = scene["k0"]
    n = len(k0)
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    mps = np.zeros(n, V.MP_TRACK_DTYPE)
    mps["proj_x"] = (k0["x"] + 3.0 + rng.normal(0, jitter, n)).astype(np.float32)
    mps["proj_y"] = (k0["y"] + 1.0 + rng.normal(0, jitter, n)).astype(np.float32)
    mps["proj_xr"] = (mps["proj_x"] - BF / np.maximum(scene["z"], 1.0)).astype(np.float32)
    mps["view_cos"] = rng.choice(np.array([0.9, 0.9985, 1.0], np.float32), n)
    mps["level"] = np.clip(k0["octave"] + rng.integers(-1, 2, n), 0, 7)
    inview = rng.random(n) < frac_in_view
    obs = rng.random(n) < 0.9
    mps["flags"] = (inview.astype(np.uint32)) | (obs.astype(np.uint32) << 1)
    return mps


@pytest.mark.parametrize("th,nnratio,seed", [(1.0, 0.8, 1), (3.0, 0.8, 2), (5.0, 0.6, 3)])
def test_local_map_matcher_equals_oracle(scene, th, nnratio, seed):
    mps = _local_map(scene, seed)
    rng = np.random.default_rng(100 + seed)
    occ = (rng.random(len(scene["k1"])) < 0.3).astype(np.uint8)   # matches TrackWithMotionModel already made
    m = V.FMatcher(scene["fe"], nnratio, True)
    m.search_init_fallbacks()
    for occupied in (None, occ):
        nm, mc = m.SearchByProjectionMapPoints(mps, scene["de0"], scene["cur"][0], scene["cur"][1], len(scene["k1"]),
                                               scene["u1"], th, occupied, (W, H))
        wn, wm = orbo.search_by_projection_mappoints(mps, scene["de0"], scene["k1"], scene["de1"], scene["u1"],
                                                     scene["sf"], W, H, th, nnratio, occupied)
        assert nm == wn and np.array_equal(mc, wm), (th, nnratio, occupied is not None)
        assert nm > 100
        if occupied is not None:
            assert not np.any((mc >= 0) & (occ == 1))
    if th == 1.0:
        assert m.search_init_fallbacks() == 0  # tracking's usual window: resolved by the parallel fixpoint alone


@pytest.mark.parametrize("env", [{"sbp_sequential": 1}, {"sbp_topm": 2}, {"sbp_topm": 1}])
def test_local_map_matcher_sequential_and_short_prefix(scene, tune, env):
    tune(**env)
    mps = _local_map(scene, 9, jitter=4.0)
    mps["flags"] |= np.uint32(1)
    m = V.FMatcher(scene["fe"], 0.8, True)
    nm, mc = m.SearchByProjectionMapPoints(mps, scene["de0"], scene["cur"][0], scene["cur"][1], len(scene["k1"]),
                                           None, 5.0, None, (W, H))
    wn, wm = orbo.search_by_projection_mappoints(mps, scene["de0"], scene["k1"], scene["de1"],
                                                 np.full(len(scene["k1"]), -1, np.float32), scene["sf"], W, H, 5.0, 0.8)
    assert nm == wn and np.array_equal(mc, wm)


def test_local_map_matcher_rejects_small_ratio(scene):
    m = V.FMatcher(scene["fe"], 0.3, True)
    with pytest.raises(V.VslamError):
        m.SearchByProjectionMapPoints(_local_map(scene, 1), scene["de0"], scene["cur"][0], scene["cur"][1],
                                      len(scene["k1"]), None, 1.0, None, (W, H))


def test_tracking_matchers_against_golden_fixture(golden_dir):
    """Device results vs the committed real-image fixture (tests/golden/tracking_hut_320x240.npz), no oracle call."""
    import os
    p = np.load(os.path.join(golden_dir, "pipeline_hut_320x240.npz"))
    g = np.load(os.path.join(golden_dir, "tracking_hut_320x240.npz"))
    fe = V.FExtractor(500, 1.2, 8, 20, 7, 320, 240, max_batch=1)
    try:
        kC, dC, _ = fe.compute(g["C"])
        assert all(np.array_equal(kC[f], g["kC"][f]) for f in kC.dtype.names) and np.array_equal(dC, g["dC"])
        ck, cd, _ = fe.slot_dev_ptrs(0)
        m = V.FMatcher(fe, 0.8, True)
        T0 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
        n, mc, _ = m.SearchByProjection(g["Tcw"], T0, tuple(float(v) for v in g["cam"]), 15, p["kL"], g["flags"], g["x3"],
                                        p["dL"], ck, cd, len(kC), None, False, (320, 240))
        assert n == int(g["sbp_nmatches"]) and np.array_equal(mc, g["sbp_match"])
        n, mc = m.SearchByProjectionMapPoints(g["mps"], p["dL"], ck, cd, len(kC), None, 3.0, g["occ"], (320, 240))
        assert n == int(g["mp_nmatches"]) and np.array_equal(mc, g["mp_match"])
        assert np.array_equal(V.ComputeDistinctiveDescriptors(fe, g["dist_desc"], g["dist_off"]), g["dist_best"])
    finally:
        fe.close()


# ---------------------------------------------------------------- relocalisation matcher (fmatcher.cpp:2689-2811)
def _kf_points(scene, rng=None):
    """The last frame as a KeyFrame: its stereo points as MapPoints with the scale-invariance range
    MapPoint::UpdateNormalAndDepth gives them (dist * scale[level] upwards, / scale[nlevels-1] downwards)."""
    k0, sf = scene["k0"], scene["sf"]
    X = scene["X"]
    d = np.linalg.norm(X, axis=1).astype(np.float32)
    mx = (np.float32(1.2) * d * sf[k0["octave"]]).astype(np.float32)
    mn = (np.float32(0.8) * d * sf[k0["octave"]] / sf[-1]).astype(np.float32)
    return mn, mx


def _run_kf(scene, Tcw, th, orb_dist, flags, mn, mx, check_ori=True, occupied=None, gemm_float=False):
    R, t = Tcw[:, :3], Tcw[:, 3]
    Ow = (-R.T @ t).astype(np.float32)
    lsf = float(np.log(np.float32(1.2)).astype(np.float32))
    m = V.FMatcher(scene["fe"], 0.9, check_ori)
    nm, mc = m.SearchByProjectionKeyFrame(Tcw, Ow, (FX, FY, CX, CY), th, orb_dist, lsf, scene["k0"], flags, scene["X"], mn,
                                          mx, scene["de0"], scene["cur"][0], scene["cur"][1], len(scene["k1"]), occupied,
                                          (W, H), gemm_float)
    wn, wm = orbo.search_by_projection_keyframe(Tcw, Ow, (FX, FY, CX, CY), th, orb_dist, lsf, scene["k0"], flags, scene["X"],
                                                mn, mx, scene["de0"], scene["k1"], scene["de1"], scene["sf"], W, H, check_ori,
                                                occupied, not gemm_float)
    assert nm == wn, (nm, wn)
    assert np.array_equal(mc, wm)
    return nm, mc


def test_search_by_projection_keyframe_equals_oracle(scene):
    rng = np.random.default_rng(12)
    mn, mx = _kf_points(scene)
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    Tcw = _pose(tx=3.0 / FX * zmed, ty=1.0 / FY * zmed)
    flags = np.ones(len(scene["k0"]), np.uint8)
    nm, mc = _run_kf(scene, Tcw, 10, 100, flags, mn, mx)       # tracking.cpp Relocalization: th 10, ORBdist 100
    assert nm > 200
    _run_kf(scene, Tcw, 3, 64, flags, mn, mx)                   # second pass: th 3, ORBdist 64
    _run_kf(scene, Tcw, 10, 100, flags, mn, mx, check_ori=False)
    _run_kf(scene, Tcw, 10, 100, flags, mn, mx, gemm_float=True)
    # sAlreadyFound / bad MapPoints, and keypoints of the current frame that already hold a MapPoint
    fl = (rng.random(len(flags)) < 0.6).astype(np.uint8)
    occ = (rng.random(len(scene["k1"])) < 0.3).astype(np.uint8)
    n2, m2 = _run_kf(scene, Tcw, 10, 100, fl, mn, mx, occupied=occ)
    assert np.all(m2[occ == 1] == -1) and np.all(fl[m2[m2 >= 0]] == 1)
    # ranges that exclude a third of the points, a tight accept threshold, a wide window (prefix exhaustion)
    mx2 = mx.copy()
    mx2[rng.random(len(mx)) < 0.33] *= 0.3
    _run_kf(scene, Tcw, 10, 30, flags, mn, mx2)
    _run_kf(scene, Tcw, 40, 100, flags, mn, mx)
    # this overload has no depth test: a camera looking the other way still "projects"
    _run_kf(scene, _pose(tz=-60.0), 10, 100, flags, mn, mx)
    _run_kf(scene, _pose(tx=0.5, yaw=0.1, tz=-3.0), 10, 100, flags, mn, mx)


def test_search_by_projection_keyframe_sequential_path(scene, tune):
    tune(sbp_sequential=1)
    mn, mx = _kf_points(scene)
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    flags = np.ones(len(scene["k0"]), np.uint8)
    _run_kf(scene, _pose(tx=3.0 / FX * zmed, ty=1.0 / FY * zmed), 20, 80, flags, mn, mx)


# ---------------------------------------------------------------- loop-closing matchers (fmatcher.cpp:750-863, :865-981)
def _run_sim3(scene, Tcw, th, ratio, flags, normals, mn, mx, variant=0, matched=None, gemm_float=False):
    R, t = Tcw[:, :3], Tcw[:, 3]
    Ow = (-R.T @ t).astype(np.float32)
    lsf = float(np.log(np.float32(1.2)).astype(np.float32))
    m = V.FMatcher(scene["fe"], 0.9, True)
    nm, mk = m.SearchByProjectionSim3(Tcw, Ow, (FX, FY, CX, CY), th, ratio, lsf, flags, scene["X"], normals, mn, mx,
                                      scene["de0"], scene["cur"][0], scene["cur"][1], len(scene["k1"]), matched, (W, H),
                                      variant, gemm_float)
    wn, wm = orbo.search_by_projection_sim3(Tcw, Ow, (FX, FY, CX, CY), th, ratio, lsf, flags, scene["X"], normals, mn, mx,
                                            scene["de0"], scene["k1"], scene["de1"], scene["sf"], W, H, variant, matched,
                                            not gemm_float)
    assert nm == wn, (nm, wn)
    assert np.array_equal(mk, wm)
    return nm, mk


def test_search_by_projection_sim3_overloads_equal_oracle(scene):
    rng = np.random.default_rng(21)
    mn, mx = _kf_points(scene)
    X = scene["X"]
    normals = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    zmed = float(np.median(scene["z"][scene["has_depth"]]))
    Tcw = _pose(tx=3.0 / FX * zmed, ty=1.0 / FY * zmed)
    flags = np.ones(len(X), np.uint8)
    nm, mk = _run_sim3(scene, Tcw, 8, 1.5, flags, normals, mn, mx)          # loopclosing.cpp: th 8, ratioHamming 1.5
    assert nm > 200 and len(set(mk[mk >= 0].tolist())) == nm                  # a candidate lands on one keypoint at most
    _run_sim3(scene, Tcw, 8, 1.5, flags, normals, mn, mx, variant=1)
    _run_sim3(scene, Tcw, 3, 1.0, flags, normals, mn, mx)
    _run_sim3(scene, Tcw, 8, 0.45, flags, normals, mn, mx, variant=1)        # threshold 22.5 -> distances <= 22
    _run_sim3(scene, Tcw, 8, 1.5, flags, normals, mn, mx, gemm_float=True)
    fl = (rng.random(len(flags)) < 0.6).astype(np.uint8)
    matched = (rng.random(len(scene["k1"])) < 0.3).astype(np.uint8)
    n2, m2 = _run_sim3(scene, Tcw, 8, 1.5, fl, normals, mn, mx, matched=matched)
    assert np.all(m2[matched == 1] == -1)
    nr2 = normals.copy()
    flip = rng.random(len(X)) < 0.5
    nr2[flip] *= -1.0                                                         # viewing angle beyond 60 degrees
    n3, m3 = _run_sim3(scene, Tcw, 8, 1.5, flags, nr2, mn, mx)
    assert not np.any(np.isin(m3[m3 >= 0], np.nonzero(flip)[0]))
    _run_sim3(scene, _pose(tz=-60.0), 8, 1.5, flags, normals, mn, mx)         # everything behind the camera
    _run_sim3(scene, _pose(tx=0.5, yaw=0.1, tz=-3.0), 30, 1.5, flags, normals, mn, mx, variant=1)
