"""GPU (-m gpu): Hamming matcher kernels, stereo L<->R matching and the initialisation matcher vs the
CPU oracle (bit-exact: indices, distances, mvuRight/mvDepth floats)."""
import os

import numpy as np
import pytest

import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    f = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=8)
    yield f
    f.close()


def _dev(arr):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    torch.cuda.synchronize()
    return t


def _top2_ref(q, t):
    dm = orbo.hamming_matrix(q, t).astype(np.int64)
    key = dm * 65536 + np.arange(t.shape[0])[None, :]
    order = np.argsort(key, axis=1, kind="stable")[:, :2]
    idx = order.astype(np.int32)
    dist = np.take_along_axis(dm, order, 1).astype(np.int32)
    return idx, dist


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (63, 257), (64, 256), (500, 300), (2011, 2013), (300, 5000)])
def test_hamming_top2_and_matrix_random(fe, nq, nt):
    rng = np.random.default_rng(nq * 7 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    t[rng.integers(0, nt, max(nt // 10, 1))] = q[rng.integers(0, nq, max(nt // 10, 1))]  # exact duplicates -> ties
    dq, dt = _dev(q), _dev(t)
    m = V.FMatcher(fe)
    idx, dist = m.hamming_top2(dq.data_ptr(), nq, dt.data_ptr(), nt)
    widx, wdist = _top2_ref(q, t)
    if nt == 1:
        assert np.array_equal(idx[:, 0], widx[:, 0]) and np.array_equal(dist[:, 0], wdist[:, 0])
        assert np.all(idx[:, 1] == -1) and np.all(dist[:, 1] == 2**31 - 1)
    else:
        assert np.array_equal(idx, widx) and np.array_equal(dist, wdist)
    dm = m.hamming_matrix(dq.data_ptr(), nq, dt.data_ptr(), nt)
    assert np.array_equal(dm, np.minimum(orbo.hamming_matrix(q, t), 255).astype(np.uint8))


def test_hamming_extremes_and_empty(fe):
    m = V.FMatcher(fe)
    q = np.zeros((3, 32), np.uint8)
    t = np.full((2, 32), 255, np.uint8)
    dq, dt = _dev(q), _dev(t)
    idx, dist = m.hamming_top2(dq.data_ptr(), 3, dt.data_ptr(), 2)
    assert np.all(dist == 256) and np.array_equal(idx, np.tile([0, 1], (3, 1)))
    assert np.all(m.hamming_matrix(dq.data_ptr(), 3, dt.data_ptr(), 2) == 255)  # saturates
    idx, dist = m.hamming_top2(dq.data_ptr(), 3, 0, 0)
    assert np.all(idx == -1)
    idx, dist = m.hamming_top2(0, 0, dt.data_ptr(), 2)
    assert idx.shape == (0, 2)


def test_hamming_top2_batch_ragged_problems_in_one_launch(fe):
    """vslam_hamming_top2_batch: independent problems of different sizes (incl. empty query / train sets, exact duplicates,
    sizes that end inside a 256-descriptor tile or a 64-query tile) == the one-problem path == the oracle"""
    rng = np.random.default_rng(20260101)
    shapes = [(1, 1), (64, 256), (63, 257), (500, 300), (2011, 2013), (300, 5000), (0, 10), (7, 0), (129, 1025), (1, 2)]
    host, dev, probs = [], [], []
    for nq, nt in shapes:
        q = rng.integers(0, 256, (max(nq, 1), 32), dtype=np.uint8)
        t = rng.integers(0, 256, (max(nt, 1), 32), dtype=np.uint8)
        if nq and nt:
            t[rng.integers(0, nt, max(nt // 10, 1))] = q[rng.integers(0, nq, max(nt // 10, 1))]  # ties
        dq, dt = _dev(q), _dev(t)
        host.append((q[:nq], t[:nt]))
        dev.append((dq, dt))
        probs.append((dq.data_ptr() if nq else 0, nq, dt.data_ptr() if nt else 0, nt))
    m = V.FMatcher(fe)
    out = m.hamming_top2_batch(probs)
    for (q, t), (idx, dist), (nq, nt) in zip(host, out, shapes):
        assert idx.shape == (nq, 2)
        if nq == 0:
            continue
        if nt == 0:
            assert np.all(idx == -1) and np.all(dist == 2**31 - 1)
            continue
        widx, wdist = _top2_ref(q, t)
        if nt == 1:
            assert np.array_equal(idx[:, 0], widx[:, 0]) and np.array_equal(dist[:, 0], wdist[:, 0])
            assert np.all(idx[:, 1] == -1) and np.all(dist[:, 1] == 2**31 - 1)
        else:
            assert np.array_equal(idx, widx) and np.array_equal(dist, wdist), (nq, nt)
    # the same problems one at a time
    for pr, (idx, dist) in zip(probs, out):
        if pr[1]:
            i1, d1 = m.hamming_top2(*pr)
            assert np.array_equal(i1, idx) and np.array_equal(d1, dist)
    with pytest.raises(V.VslamError):
        m.hamming_top2_batch([probs[1]] * 33)  # more than VSLAM_MAX_TOP2_JOBS


def test_hamming_top2_batch_sixteen_stereo_pairs_full_size(fe):
    """the brute-force matches of the 16 stereo pairs of a step (frame.cpp:1167-1174 per frame) in one launch, at
    BASELINE size, against the oracle on two pairs and through size-independent properties on all"""
    rng = np.random.default_rng(5)
    N = 2000
    base = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    host, dev, probs = [], [], []
    for p in range(16):
        l = base.copy()
        flip = rng.integers(0, 256, (N, 32), dtype=np.uint8) & rng.integers(0, 256, (N, 32), dtype=np.uint8) & rng.integers(0, 256, (N, 32), dtype=np.uint8)
        perm = rng.permutation(N)
        r = (l ^ flip)[perm]  # every left descriptor has a near twin somewhere on the right
        dl, dr = _dev(l), _dev(r)
        host.append((l, r, perm))
        dev.append((dl, dr))
        probs.append((dl.data_ptr(), N, dr.data_ptr(), N))
    out = V.FMatcher(fe).hamming_top2_batch(probs)
    for p in (0, 15):
        widx, wdist = _top2_ref(host[p][0], host[p][1])
        assert np.array_equal(out[p][0], widx) and np.array_equal(out[p][1], wdist)
    for p in range(16):
        l, r, perm = host[p]
        idx, dist = out[p]
        inv = np.empty(N, np.int64)
        inv[perm] = np.arange(N)
        assert np.mean(idx[:, 0] == inv) > 0.99            # the twin is the nearest
        assert np.all(dist[:, 0] <= dist[:, 1])
        d0 = np.unpackbits(l ^ r[idx[:, 0]], axis=1).sum(1)
        assert np.array_equal(d0, dist[:, 0])                # the reported distance is the distance to the reported index


def test_hamming_on_real_descriptors_full_size(fe):
    """BASELINE size (2000 x 2000): extractor output fed straight from HBM."""
    L, R = synth.make_stereo_pair(1241, 376)
    (kL, dL, _), (kR, dR, _) = fe.compute_batch([L, R])
    pk0, pd0, n0 = fe.slot_buffers(0)
    pk1, pd1, n1 = fe.slot_buffers(1)
    assert (n0, n1) == (len(kL), len(kR))
    idx, dist = V.FMatcher(fe).hamming_top2(pd0, n0, pd1, n1)
    widx, wdist = _top2_ref(dL, dR)
    assert np.array_equal(idx, widx) and np.array_equal(dist, wdist)
    # size-independent properties: self-match is distance 0 at own index; symmetric matrix
    idx, dist = V.FMatcher(fe).hamming_top2(pd0, n0, pd0, n0)
    assert np.all(dist[:, 0] == 0)
    dm = V.FMatcher(fe).hamming_matrix(pd0, n0, pd0, n0)
    assert np.array_equal(dm, dm.T) and np.all(np.diag(dm) == 0)


def _stereo_ref(L, R, nf, bf, fx):
    eL, eR = orbo.Extractor(nf), orbo.Extractor(nf)
    kL, dL, _ = eL.compute(L)
    kR, dR, _ = eR.compute(R)
    u, dep, bi, bs = orbo.stereo(eL, eR, kL, dL, kR, dR, bf, fx)
    return kL, u, dep


def test_stereo_matches_kitti_same_context(fe):
    L, R = synth.make_stereo_pair(1241, 376, step=1)
    fe.compute_batch([L, R])
    u, d = V.ComputeStereoMatches(fe, 0, fe, 1, 386.1448, 718.856)
    kL, wu, wd = _stereo_ref(L, R, 2000, 386.1448, 718.856)
    assert np.array_equal(u, wu) and np.array_equal(d, wd)
    ok = u >= 0
    assert ok.sum() > 300
    truth = synth.row_disparity(376)[kL["y"][ok].astype(int)]
    assert np.mean(np.abs((kL["x"][ok] - u[ok]) - truth) < 1.5) > 0.95


def test_stereo_two_contexts_like_left_right_extractors():
    """The reference owns mpORBextractorLeft / mpORBextractorRight (tracking.cpp:1087-1090)."""
    L, R = synth.make_stereo_pair(752, 480, seed=4)
    feL = V.FExtractor(1200, 1.2, 8, 20, 7, 752, 480)
    feR = V.FExtractor(1200, 1.2, 8, 20, 7, 752, 480)
    try:
        feL.compute(L)
        feR.compute(R)
        u, d = V.ComputeStereoMatches(feL, 0, feR, 0, 40.0, 435.2)
        _, wu, wd = _stereo_ref(L, R, 1200, 40.0, 435.2)
        assert np.array_equal(u, wu) and np.array_equal(d, wd) and (u >= 0).sum() > 100
    finally:
        feL.close()
        feR.close()


def test_stereo_batch_of_pairs(fe):
    frames = [synth.make_stereo_pair(1241, 376, step=s) for s in range(4)]
    imgs = [im for pair in frames for im in pair]
    fe.compute_batch(imgs)
    res = V.ComputeStereoMatchesBatch(fe, [0, 2, 4, 6], fe, [1, 3, 5, 7], 386.1448, 718.856)
    for s in range(4):
        _, wu, wd = _stereo_ref(frames[s][0], frames[s][1], 2000, 386.1448, 718.856)
        assert np.array_equal(res[s][0], wu) and np.array_equal(res[s][1], wd), s


def test_stereo_golden_real_images(golden_dir):
    g = np.load(os.path.join(golden_dir, "pipeline_hut_320x240.npz"))
    f = V.FExtractor(500, 1.2, 8, 20, 7, 320, 240, max_batch=2)
    try:
        f.compute_batch([g["L"], g["R"]])
        u, d = V.ComputeStereoMatches(f, 0, f, 1, 40.0, 400.0)
        assert np.array_equal(u, g["uRight"]) and np.array_equal(d, g["depth"])
    finally:
        f.close()


def test_stereo_no_matches_when_right_is_unrelated(fe):
    L = synth.make_frame(1241, 376, seed=1)
    R = np.full((376, 1241), 90, np.uint8)  # flat right image: no right keypoints at all
    fe.compute_batch([L, R])
    u, d = V.ComputeStereoMatches(fe, 0, fe, 1, 386.1448, 718.856)
    assert len(u) > 1000 and np.all(u == -1) and np.all(d == -1)


@pytest.mark.parametrize("nf,lap", [(1000, (0, 1000)), (2000, (0, 0))])
def test_search_for_initialization(nf, lap):
    a = synth.make_frame(1241, 376, step=0)
    b = synth.make_frame(1241, 376, step=1)
    f = V.FExtractor(nf, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
    try:
        (k1, d1, _), (k2, d2, _) = f.compute_batch([a, b], lap)
        _, pd1, n1 = f.slot_buffers(0)
        _, pd2, n2 = f.slot_buffers(1)
        prev = np.stack([k1["x"], k1["y"]], 1)
        m = V.FMatcher(f, 0.9, True)
        nm, m12, pm = m.SearchForInitialization(k1, pd1, k2, pd2, prev, 100)
        wn, wm, wp = orbo.search_for_initialization(k1, d1, k2, d2, 1241, 376, window=100, nnratio=0.9)
        assert nm == wn and np.array_equal(m12, wm) and np.array_equal(pm, wp) and nm > 50
        # second call with the updated vbPrevMatched and a tighter window (tracking.cpp:2323 loop)
        nm2, m12b, _ = m.SearchForInitialization(k1, pd1, k2, pd2, pm, 20)
        wn2, wm2, _ = orbo.search_for_initialization(k1, d1, k2, d2, 1241, 376, prev_matched=wp, window=20)
        assert nm2 == wn2 and np.array_equal(m12b, wm2)
    finally:
        f.close()


def test_search_for_initialization_batch_equals_single(fe):
    frames = [synth.make_frame(1241, 376, step=s) for s in range(5)]
    res = fe.compute_batch(frames, (0, 1000))
    bufs = [fe.slot_buffers(s) for s in range(5)]
    m = V.FMatcher(fe, 0.9, True)
    pairs = [(res[s][0], bufs[s][1], res[s + 1][0], bufs[s + 1][1], np.stack([res[s][0]["x"], res[s][0]["y"]], 1))
             for s in range(4)]
    out = m.SearchForInitializationBatch(pairs, 100)
    for s in range(4):
        wn, wm, wp = orbo.search_for_initialization(res[s][0], res[s][1], res[s + 1][0], res[s + 1][1], 1241, 376,
                                                    window=100, nnratio=0.9)
        assert out[s][0] == wn and np.array_equal(out[s][1], wm) and np.array_equal(out[s][2], wp), s
        assert wn > 50


def test_frame_stereo_async_pipeline(fe):
    """Frame::Frame(stereo) hot section in one enqueue: extraction of L,R and ComputeStereoMatches."""
    import torch
    frames = [synth.make_stereo_pair(1241, 376, step=s) for s in range(3)]
    pitch = 1280
    dev = torch.zeros((6, 376, pitch), dtype=torch.uint8, device="cuda")
    for s in range(3):
        dev[2 * s, :, :1241] = torch.from_numpy(frames[s][0]).cuda()
        dev[2 * s + 1, :, :1241] = torch.from_numpy(frames[s][1]).cuda()
    torch.cuda.synchronize()
    fe.frame_stereo_async([dev[i].data_ptr() for i in range(6)], pitch, 386.1448, 718.856)
    feats, st = fe.frame_stereo_wait()
    for s in range(3):
        eL, eR = orbo.Extractor(2000), orbo.Extractor(2000)
        kL, dL, _ = eL.compute(frames[s][0])
        kR, dR, _ = eR.compute(frames[s][1])
        wu, wd, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, 386.1448, 718.856)
        assert all(np.array_equal(feats[2 * s][0][f], kL[f]) for f in kL.dtype.names)
        assert np.array_equal(feats[2 * s][1], dL) and np.array_equal(feats[2 * s + 1][1], dR)
        assert np.array_equal(st[s][0], wu) and np.array_equal(st[s][1], wd)
        assert (wu >= 0).sum() > 300


def test_search_init_host_replay_path_agrees():
    """init_match_host: distance matrices on the GPU, order-dependent replay on the host."""
    a, b = synth.make_frame(1241, 376, step=3), synth.make_frame(1241, 376, step=4)
    f = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=2, tuning=dict(init_match_host=1))
    try:
        (k1, d1, _), (k2, d2, _) = f.compute_batch([a, b], (0, 1000))
        _, pd1, _ = f.slot_buffers(0)
        _, pd2, _ = f.slot_buffers(1)
        nm, m12, pm = V.FMatcher(f, 0.9, True).SearchForInitialization(k1, pd1, k2, pd2,
                                                                       np.stack([k1["x"], k1["y"]], 1), 100)
        wn, wm, wp = orbo.search_for_initialization(k1, d1, k2, d2, 1241, 376, window=100, nnratio=0.9)
        assert nm == wn and np.array_equal(m12, wm) and np.array_equal(pm, wp)
    finally:
        f.close()


def test_search_init_device_jobs_async(fe):
    """Device-pointer jobs straight from the slot buffers: frame s-1 -> frame s for a whole batch, plus a
    window-20 / ratio-0.6 / no-orientation variant; everything equals the oracle."""
    frames = [synth.make_frame(1241, 376, step=s) for s in range(6)]
    res = fe.compute_batch(frames, (0, 1000))
    jobs = []
    for s in range(1, 6):
        p, c = fe.slot_dev_ptrs(s - 1), fe.slot_dev_ptrs(s)
        jobs.append((p[0], p[1], p[2], c[0], c[1], c[2], 0))
    m = V.FMatcher(fe, 0.9, True)
    m.search_init_dev_async(jobs, 100)
    out = m.search_init_dev_wait([len(res[s - 1][0]) for s in range(1, 6)], want_prev=True)
    for j, s in enumerate(range(1, 6)):
        wn, wm, wp = orbo.search_for_initialization(res[s - 1][0], res[s - 1][1], res[s][0], res[s][1], 1241, 376,
                                                    window=100, nnratio=0.9)
        assert out[j][0] == wn and np.array_equal(out[j][1], wm) and np.array_equal(out[j][2], wp), s
        assert wn > 50
    m2 = V.FMatcher(fe, 0.6, False)
    m2.search_init_dev_async(jobs[:2], 20)
    out = m2.search_init_dev_wait([len(res[0][0]), len(res[1][0])])
    for j in range(2):
        wn, wm, _ = orbo.search_for_initialization(res[j][0], res[j][1], res[j + 1][0], res[j + 1][1], 1241, 376,
                                                   window=20, nnratio=0.6, check_ori=False)
        assert out[j][0] == wn and np.array_equal(out[j][1], wm), j


@pytest.mark.parametrize("topm", ["1", "2", "8"])
def test_search_init_sorted_prefix_and_rescan_path(topm):
    """k_si_replay works from a sorted prefix of init_topm candidates per query and re-scans the whole
    window when the prefix is exhausted: a prefix of 1 or 2 forces that path, the result must not change."""
    frames = [synth.make_frame(1241, 376, seed=11, step=s) for s in range(3)]
    f = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=3, tuning=dict(init_topm=int(topm)))
    try:
        res = f.compute_batch(frames, (0, 1000))
        jobs = []
        for s in (1, 2):
            p, c = f.slot_dev_ptrs(s - 1), f.slot_dev_ptrs(s)
            jobs.append((p[0], p[1], p[2], c[0], c[1], c[2], 0))
        m = V.FMatcher(f, 0.9, True)
        m.search_init_fallbacks()
        m.search_init_dev_async(jobs, 100)
        out = m.search_init_dev_wait([len(res[0][0]), len(res[1][0])], want_prev=True)
        nfb = m.search_init_fallbacks()
        for j in range(2):
            wn, wm, wp = orbo.search_for_initialization(res[j][0], res[j][1], res[j + 1][0], res[j + 1][1], 1241, 376,
                                                        window=100, nnratio=0.9)
            assert out[j][0] == wn and np.array_equal(out[j][1], wm) and np.array_equal(out[j][2], wp), j
            assert wn > 50
        nq = sum(int((res[j][0]["octave"] == 0).sum()) for j in range(2))
        if topm == "8":
            assert nfb <= nq // 10, (nfb, nq)  # the prefix almost always decides
        else:
            assert nfb > 0, "tiny prefix must exercise the re-scan path"
    finally:
        f.close()


def test_search_init_large_feature_count():
    """The mono initialisation extractor uses 5 x nFeatures (tracking.cpp:1093): 2170 octave-0 keypoints."""
    a, b = synth.make_frame(1241, 376, seed=5, step=0), synth.make_frame(1241, 376, seed=5, step=1)
    f = V.FExtractor(10000, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
    try:
        (k1, d1, _), (k2, d2, _) = f.compute_batch([a, b], (0, 1000))
        _, pd1, _ = f.slot_buffers(0)
        _, pd2, _ = f.slot_buffers(1)
        nm, m12, pm = V.FMatcher(f, 0.9, True).SearchForInitialization(k1, pd1, k2, pd2,
                                                                       np.stack([k1["x"], k1["y"]], 1), 100)
        wn, wm, wp = orbo.search_for_initialization(k1, d1, k2, d2, 1241, 376, window=100, nnratio=0.9)
        assert nm == wn and np.array_equal(m12, wm) and np.array_equal(pm, wp) and nm > 300
    finally:
        f.close()


def test_pack_slots_layout_and_matcher_on_packed_buffer(fe):
    """vslam_fe_pack_slots: the layout an RCCL all-gather moves; the device matcher must work on it."""
    import torch
    frames = [synth.make_frame(1241, 376, step=s) for s in range(3)]
    res = fe.compute_batch(frames, (0, 1000))
    sb = fe.slot_bytes
    buf = torch.zeros(3 * sb, dtype=torch.uint8, device="cuda")
    fe.pack_slots(3, buf.data_ptr(), sb)
    host = buf.cpu().numpy()
    for s in range(3):
        k, d, mono = res[s]
        hdr = host[s * sb:s * sb + 16].view(np.int32)
        assert list(hdr) == [len(k), mono, fe.cap, 0]
        pk = host[s * sb + 16:s * sb + 16 + len(k) * 28].view(V.KP_DTYPE)
        assert all(np.array_equal(pk[f], k[f]) for f in k.dtype.names)
        off = s * sb + 16 + fe.cap * 28
        assert np.array_equal(host[off:off + len(k) * 32].reshape(-1, 32), d)
    # frame 0 -> frame 1 with frame 0 read from the packed buffer (as a neighbour rank's slot would be)
    base = buf.data_ptr()
    cur = fe.slot_dev_ptrs(1)
    m = V.FMatcher(fe, 0.9, True)
    m.search_init_dev_async([(base + 16, base + 16 + fe.cap * 28, base, cur[0], cur[1], cur[2], 0)], 100)
    out = m.search_init_dev_wait([len(res[0][0])])
    wn, wm, _ = orbo.search_for_initialization(res[0][0], res[0][1], res[1][0], res[1][1], 1241, 376, window=100,
                                               nnratio=0.9)
    assert out[0][0] == wn and np.array_equal(out[0][1], wm)


def test_odd_feature_count_keeps_buffers_aligned():
    # nfeatures not a multiple of 4: the slot capacity is rounded so packed descriptors stay 16-byte aligned
    a, b = synth.make_frame(640, 360, seed=2, step=0), synth.make_frame(640, 360, seed=2, step=1)
    f = V.FExtractor(777, 1.2, 8, 20, 7, 640, 360, max_batch=2)
    try:
        assert f.cap % 4 == 0
        (k1, d1, _), (k2, d2, _) = f.compute_batch([a, b], (0, 1000))
        e = orbo.Extractor(777)
        ko, do, _ = e.compute(a, lap=(0, 1000))
        assert all(np.array_equal(k1[x], ko[x]) for x in ko.dtype.names) and np.array_equal(d1, do)
        p, c = f.slot_dev_ptrs(0), f.slot_dev_ptrs(1)
        m = V.FMatcher(f, 0.9, True)
        m.search_init_dev_async([(p[0], p[1], p[2], c[0], c[1], c[2], 0)], 100)
        out = m.search_init_dev_wait([len(k1)])
        wn, wm, _ = orbo.search_for_initialization(k1, d1, k2, d2, 640, 360, window=100, nnratio=0.9)
        assert out[0][0] == wn and np.array_equal(out[0][1], wm)
    finally:
        f.close()


def test_compute_distinctive_descriptors_batched(fe):
    """MapPoint::ComputeDistinctiveDescriptors (mappoint.cpp:322-390): sets of 0..300 observations; real descriptors
    with a few bits flipped per observation (ties in the medians are common: 'first wins' must hold)."""
    res = fe.compute_batch([synth.make_frame(1241, 376, step=0)])
    base = res[0][1]
    rng = np.random.default_rng(21)
    sizes = [0, 1, 2, 3, 4, 5, 8, 17, 64, 65, 100, 256, 300] + list(rng.integers(1, 40, 200))
    descs, off = [], [0]
    for n in sizes:
        d = np.repeat(base[rng.integers(0, len(base))][None, :], n, 0).copy()
        for i in range(n):
            for f in rng.integers(0, 256, rng.integers(0, 40)):
                d[i, f // 8] ^= np.uint8(1 << (f % 8))
        descs.append(d)
        off.append(off[-1] + n)
    desc = np.concatenate(descs, 0)
    got = V.ComputeDistinctiveDescriptors(fe, desc, off)
    want = orbo.distinctive_descriptors(desc, off)
    assert np.array_equal(got, want)
    assert got[0] == -1 and got[1] == 0


def test_compute_bow_tree_walk_and_vectors(fe):
    """Frame::ComputeBoW (DBoW3 Vocabulary::transform, levelsup 4): tree walk on the GPU for real descriptors and
    for descriptors engineered to tie between children; BowVector / FeatureVector equal the restatement."""
    voc = synth.make_vocabulary(10, 5, seed=11)           # 111 111 nodes, 100 000 words
    res = fe.compute_batch([synth.make_frame(1241, 376, step=s) for s in range(3)])
    V_ = V.Vocabulary(voc)
    try:
        for s in range(3):
            _, pd, n = fe.slot_buffers(s)
            got = V_.transform(fe, pd, n, 4)
            want = orbo.bow_transform(voc, res[s][1], 4)
            for k in ("word", "weight", "nid", "bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
                assert np.array_equal(got[k], want[k]), (s, k)
            assert abs(got["bow_vals"].sum() - 1.0) < 1e-9 and len(got["bow_ids"]) > 500
        # all slots in one launch, counts read from HBM
        V_.transform_slots_async(fe, 0, 3, 4)
        out = V_.transform_slots_wait([len(res[s][0]) for s in range(3)])
        for s in range(3):
            want = orbo.bow_transform(voc, res[s][1], 4)
            for k in ("word", "nid", "bow_ids", "bow_vals", "fv_feat"):
                assert np.array_equal(out[s][k], want[k]), (s, k)
    finally:
        V_.close()
    # ties: every child of the root identical -> the first child must win at every such node
    tie = synth.make_vocabulary(4, 3, seed=2)
    tie["desc"][1:5] = tie["desc"][1]
    Vt = V.Vocabulary(tie)
    try:
        import torch
        q = np.repeat(tie["desc"][1][None, :], 8, 0)
        dq = torch.from_numpy(q).cuda()
        got = Vt.transform(fe, dq.data_ptr(), 8, 1)
        want = orbo.bow_transform(tie, q, 1)
        assert np.array_equal(got["word"], want["word"]) and np.array_equal(got["nid"], want["nid"])
    finally:
        Vt.close()


def test_vocabulary_loaded_from_dbow3_files(fe, tmp_path, golden_dir):
    """Vocabulary::load (Vocabulary.cpp:1084-1112) -> ComputeBoW: the plain binary stream, the committed stream whose
    chunks the reference's QuickLZ compressed, and the text form give the BowVector / FeatureVector of the oracle fed
    with the arrays the files were written from (text: weights rounded to float, as load_fromtxt does)."""
    import vocfile
    res = fe.compute_batch([synth.make_frame(1241, 376, step=5)])
    _, pd, n = fe.slot_buffers(0)
    desc = res[0][1].copy()
    voc = synth.make_vocabulary(8, 3, seed=5)
    cases = [(os.path.join(golden_dir, "voc_k8_L3_quicklz.dbow3"), voc)]
    vocfile.write_binary(str(tmp_path / "plain.dbow3"), voc)
    cases.append((str(tmp_path / "plain.dbow3"), voc))
    vocfile.write_text(str(tmp_path / "ORBvoc.txt"), voc)
    as_text = dict(voc, weight=voc["weight"].astype(np.float32).astype(np.float64))
    cases.append((str(tmp_path / "ORBvoc.txt"), as_text))
    for path, v in cases:
        vv = V.Vocabulary.load(path)
        try:
            assert (vv.L, vv.weighting, vv.norm, vv.n_nodes) == (3, 0, 1, 585)
            got = vv.transform(fe, pd, n, 2)
            want = orbo.bow_transform(v, desc, 2)
            for k in ("word", "weight", "nid", "bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
                assert np.array_equal(got[k], want[k]), (path, k)
        finally:
            vv.close()
    with pytest.raises(V.VslamError) as e:
        V.Vocabulary.load(str(tmp_path / "nothing.dbow3"))
    assert e.value.code == V.ERR_INVALID and "cannot open" in str(e.value)


@pytest.mark.parametrize("k,L,levelsup,ratio,ori", [(10, 4, 2, 0.7, True), (10, 5, 4, 0.9, False), (4, 3, 3, 0.75, True)])
def test_search_by_bow_equals_oracle(fe, k, L, levelsup, ratio, ori):
    """FMatcher::SearchByBoW (fmatcher.cpp:546-748): ComputeBoW of both frames on the device, then one wave per
    shared vocabulary node; (4,3,3): a single node holds every feature (more than 64 candidates per lane group)."""
    voc = synth.make_vocabulary(k, L, seed=17)
    res = fe.compute_batch([synth.make_frame(1241, 376, step=s) for s in range(2)])
    res = [(a.copy(), b.copy(), c) for a, b, c in res]
    vv = V.Vocabulary(voc)
    try:
        vv.transform_slots_async(fe, 0, 2, levelsup)
        bw = vv.transform_slots_wait([len(res[0][0]), len(res[1][0])])
        for s in range(2):
            want = orbo.bow_transform(voc, res[s][1], levelsup)
            assert np.array_equal(bw[s]["fv_feat"], want["fv_feat"]) and np.array_equal(bw[s]["fv_nodes"], want["fv_nodes"])
        rng = np.random.default_rng(4)
        flags = (rng.random(len(res[0][0])) < 0.8).astype(np.uint8)
        _, dk, _ = fe.slot_buffers(0)
        _, df, _ = fe.slot_buffers(1)
        m = V.FMatcher(fe, ratio, ori)
        nm, mf = m.SearchByBoW(res[0][0], dk, flags, bw[0], res[1][0], df, bw[1])
        wn, wm = orbo.search_by_bow(res[0][0], res[0][1], flags, bw[0], res[1][0], res[1][1], bw[1], ratio, ori)
        assert nm == wn and np.array_equal(mf, wm)
        assert nm > 100 and np.all(flags[mf[mf >= 0]] == 1)
    finally:
        vv.close()


def test_search_by_bow_keyframes_equals_oracle(fe):
    """FMatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (fmatcher.cpp:1100-1240): MapPoint flags on both sides,
    strict threshold, result indexed by the first KeyFrame."""
    voc = synth.make_vocabulary(10, 4, seed=23)
    res = fe.compute_batch([synth.make_frame(1241, 376, step=s) for s in (3, 4)])
    res = [(a.copy(), b.copy(), c) for a, b, c in res]
    vv = V.Vocabulary(voc)
    try:
        vv.transform_slots_async(fe, 0, 2, 2)
        bw = vv.transform_slots_wait([len(res[0][0]), len(res[1][0])])
        rng = np.random.default_rng(9)
        f1 = (rng.random(len(res[0][0])) < 0.7).astype(np.uint8)
        f2 = (rng.random(len(res[1][0])) < 0.7).astype(np.uint8)
        _, d1, _ = fe.slot_buffers(0)
        _, d2, _ = fe.slot_buffers(1)
        for ratio, ori in ((0.8, True), (0.6, False)):
            nm, m12 = V.FMatcher(fe, ratio, ori).SearchByBoWKeyFrames(res[0][0], d1, f1, bw[0], res[1][0], d2, f2, bw[1])
            wn, wm = orbo.search_by_bow_keyframes(res[0][0], res[0][1], f1, bw[0], res[1][0], res[1][1], f2, bw[1], ratio, ori)
            assert nm == wn and np.array_equal(m12, wm)
            assert nm > 50 and np.all(f1[m12 >= 0] == 1) and np.all(f2[m12[m12 >= 0]] == 1)
    finally:
        vv.close()


def test_empty_frames_through_every_entry_point():
    """A flat image yields no keypoints; every matcher entry point must accept empty frames on either side (and a
    pure-noise image must not overflow anything)."""
    W, H = 640, 360
    fe = V.FExtractor(500, 1.2, 8, 20, 7, W, H, max_batch=4)
    flat = np.full((H, W), 127, np.uint8)
    tex = synth.make_frame(W, H)
    noise = np.random.default_rng(0).integers(0, 256, (H, W), dtype=np.uint8)
    res = fe.compute_batch([flat, tex, noise, flat])
    for s, im in enumerate([flat, tex, noise, flat]):
        ref = orbo.Extractor(500).compute(im)
        assert len(ref[0]) == len(res[s][0]) and np.array_equal(ref[1], res[s][1]), s
    # stereo with an empty left / empty right
    for (a, b) in [(0, 1), (1, 0), (0, 3), (2, 1)]:
        u, d = V.ComputeStereoMatches(fe, a, fe, b, 386.0, 718.0)
    m = V.FMatcher(fe, 0.9, True)
    bufs = [fe.slot_buffers(s) for s in range(4)]
    for (a, b) in [(0, 1), (1, 0), (0, 3), (1, 2)]:
        k1, k2 = res[a][0], res[b][0]
        prev = np.stack([k1["x"], k1["y"]], 1) if len(k1) else np.zeros((0, 2), np.float32)
        nm, m12, pm = m.SearchForInitialization(k1, bufs[a][1], k2, bufs[b][1], prev, 100)
        wn, wm, wp = orbo.search_for_initialization(k1, res[a][1], k2, res[b][1], W, H, window=100, nnratio=0.9)
        assert nm == wn and np.array_equal(m12, wm), (a, b)
    # device-resident jobs with empty frames
    jobs = []
    for (a, b) in [(0, 1), (1, 0), (0, 3), (1, 2)]:
        p, c = fe.slot_dev_ptrs(a), fe.slot_dev_ptrs(b)
        jobs.append((p[0], p[1], p[2], c[0], c[1], c[2], 0))
    m.search_init_dev_async(jobs, 100)
    out = m.search_init_dev_wait([len(res[a][0]) for a in (0, 1, 0, 1)])
    # BoW on empty
    voc = synth.make_vocabulary(10, 3)
    vv = V.Vocabulary(voc)
    vv.transform_slots_async(fe, 0, 4, 2)
    bw = vv.transform_slots_wait([len(r[0]) for r in res])
    nm, mf = m.SearchByBoW(res[0][0], bufs[0][1], np.zeros(0, np.uint8), bw[0], res[1][0], bufs[1][1], bw[1])
    nm, mf = m.SearchByBoW(res[1][0], bufs[1][1], np.ones(len(res[1][0]), np.uint8), bw[1], res[0][0], bufs[0][1], bw[0])
    vv.close(); fe.close()


@pytest.mark.parametrize("pattern,disp", [("checker5", 23), ("checker16", 10), ("binary_noise", 0), ("dots", 10), ("blocks", 23)])
def test_matchers_on_tie_heavy_patterns(pattern, disp):
    """Periodic / binary images make Hamming and SAD ties the rule: 'first wins' and the window order must hold in
    ComputeStereoMatches and SearchForInitialization."""
    W, H = 752, 480
    yy, xx = np.mgrid[0:H, 0:W + 64]
    rng = np.random.default_rng(5)
    base = {
        "checker5": lambda: (((xx // 5 + yy // 5) & 1) * 255).astype(np.uint8),
        "checker16": lambda: (((xx // 16 + yy // 16) & 1) * 255).astype(np.uint8),
        "binary_noise": lambda: (rng.integers(0, 2, xx.shape) * 255).astype(np.uint8),
        "dots": lambda: np.where(((xx % 7) == 3) & ((yy % 7) == 3), 255, 30).astype(np.uint8),
        "blocks": lambda: rng.integers(0, 256, (H // 8 + 1, (W + 64) // 8 + 1)).repeat(8, 0).repeat(8, 1)[:H, :W + 64].astype(np.uint8),
    }[pattern]()
    L = np.ascontiguousarray(base[:, 32:32 + W])
    R = np.ascontiguousarray(base[:, 32 + disp:32 + disp + W])
    fe = V.FExtractor(1500, 1.2, 8, 20, 7, W, H, max_batch=2)
    try:
        res = fe.compute_batch([L, R])
        res = [(k.copy(), d.copy(), mm) for k, d, mm in res]
        u, dep = V.ComputeStereoMatches(fe, 0, fe, 1, 386.0, 718.0)
        eL, eR = orbo.Extractor(1500), orbo.Extractor(1500)
        kL, dL, _ = eL.compute(L)
        kR, dR, _ = eR.compute(R)
        wu, wd, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, 386.0, 718.0)
        assert np.array_equal(u, wu) and np.array_equal(dep, wd)
        _, p0, _ = fe.slot_buffers(0)
        _, p1, _ = fe.slot_buffers(1)
        prev = np.stack([res[0][0]["x"], res[0][0]["y"]], 1)
        nm, m12, pm = V.FMatcher(fe, 0.9, True).SearchForInitialization(res[0][0], p0, res[1][0], p1, prev, 100)
        wn, wm, wp = orbo.search_for_initialization(kL, dL, kR, dR, W, H, window=100, nnratio=0.9)
        assert nm == wn and np.array_equal(m12, wm) and np.array_equal(pm, wp)
    finally:
        fe.close()


@pytest.mark.parametrize("lap", [(0, 0), (300, 900)])
def test_stereo_fisheye_candidates_equal_knn_plus_ratio_test(fe, lap):
    """Frame::ComputeStereoFishEyeMatches (frame.cpp:1149-1174) up to the ratio test: knnMatch(k=2) of the lapping-area
    descriptors (rows from monoIndex on: FExtractor::compute stores the lapping-area keypoints at the tail, :1118-1127) +
    Lowe 0.7.  Synthetic pair and the reference's hut_stereo frames 01 / 02 (real pixels)."""
    L, R = synth.make_stereo_pair(1241, 376, seed=5)
    (kL, dL, mL), (kR, dR, mR) = fe.compute_batch([L, R], lap)
    if lap == (0, 0):
        assert mL == len(kL)  # nothing in the lapping area: every keypoint is "mono" -> match everything instead
        mL = mR = 0
    _, pd0, n0 = fe.slot_buffers(0)
    _, pd1, n1 = fe.slot_buffers(1)
    m = V.FMatcher(fe)
    l2r, d0, d1, nc = m.ComputeStereoFishEyeCandidates(pd0, n0, mL, pd1, n1, mR)
    wl, w0, w1, wn = orbo.stereo_fisheye_candidates(dL, mL, dR, mR)
    assert nc == wn and nc > 50
    assert np.array_equal(l2r, wl) and np.array_equal(d0, w0) and np.array_equal(d1, w1)
    assert np.all(l2r[:mL] == -1) and np.all((l2r == -1) | (l2r >= mR))
    # degenerate shapes: empty lapping area, a single right descriptor (knnMatch returns < 2 matches -> no candidate)
    l2r, _, _, nc = m.ComputeStereoFishEyeCandidates(pd0, n0, n0, pd1, n1, mR)
    assert nc == 0 and np.all(l2r == -1)
    l2r, _, _, nc = m.ComputeStereoFishEyeCandidates(pd0, n0, mL, pd1, n1, n1 - 1)
    assert nc == 0 and np.all(l2r == -1)


def test_stereo_fisheye_candidates_on_the_reference_hut_frames():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_images.npz"))
    f = V.FExtractor(1200, 1.2, 8, 20, 7, 752, 480, max_batch=2)
    try:
        (kL, dL, mL), (kR, dR, mR) = f.compute_batch([z["hut1"], z["hut2"]], (100, 650))
        _, pd0, n0 = f.slot_buffers(0)
        _, pd1, n1 = f.slot_buffers(1)
        l2r, d0, d1, nc = V.FMatcher(f).ComputeStereoFishEyeCandidates(pd0, n0, mL, pd1, n1, mR)
        wl, w0, w1, wn = orbo.stereo_fisheye_candidates(dL, mL, dR, mR)
        assert nc == wn and nc > 20 and np.array_equal(l2r, wl) and np.array_equal(d0, w0) and np.array_equal(d1, w1)
    finally:
        f.close()


def test_deferred_delivery_rides_with_the_init_matcher(fe):
    """want_host = 2 (to_host="with_matcher"): the extraction's keypoints / descriptors leave the device together with the
    outputs of the SearchForInitialization that follows on the context -- one transfer per step; without a matcher call
    the wait delivers.  A graph replay (pinned images, second pass) must defer too."""
    B = fe.max_batch
    frames = [synth.make_frame(1241, 376, seed=23, step=s) for s in range(B)]
    e = orbo.Extractor(2000)
    want = [e.compute(f, lap=(0, 1000)) for f in frames]
    pin = V.PinnedImages(B, 376, 1241, 1241)
    try:
        for s in range(B):
            pin.array[s][:] = frames[s]
        m = V.FMatcher(fe, 0.9, True)
        for rep in range(3):  # rep 0 captures the graph, rep 1 replays it, rep 2: no matcher call follows
            sent0 = fe.delivery_stats()
            fe.compute_batch_async(pin.ptrs, 1241, (0, 1000), to_host="with_matcher", where=V.IMGS_PINNED)
            if rep < 2:
                jobs = []
                for s in range(1, B):
                    p, c = fe.slot_dev_ptrs(s - 1), fe.slot_dev_ptrs(s)
                    jobs.append((p[0], p[1], p[2], c[0], c[1], c[2], 0))
                m.search_init_dev_async(jobs, 100)
            res = fe.wait(copy=True)
            sent = fe.delivery_stats()
            assert sent[0] - sent0[0] == 1, (rep, sent0, sent)  # ONE transfer for the step, matcher outputs included
            assert sent[1] - sent0[1] >= B * fe.cap * 60
            for s in range(B):
                assert len(res[s][0]) == len(want[s][0]), (rep, s)
                assert all(np.array_equal(res[s][0][f], want[s][0][f]) for f in want[s][0].dtype.names), (rep, s)
                assert np.array_equal(res[s][1], want[s][1]) and res[s][2] == want[s][2], (rep, s)
            if rep < 2:
                out = m.search_init_dev_wait([len(want[s - 1][0]) for s in range(1, B)])
                for j, s in enumerate(range(1, B)):
                    wn, wm, _ = orbo.search_for_initialization(want[s - 1][0], want[s - 1][1], want[s][0], want[s][1], 1241, 376,
                                                               window=100, nnratio=0.9)
                    assert out[j][0] == wn and np.array_equal(out[j][1], wm), (rep, j)
    finally:
        pin.close()
    # the same step with separate deliveries: two transfers
    sent0 = fe.delivery_stats()
    dev = [_dev(f) for f in frames]
    fe.compute_batch_async([t.data_ptr() for t in dev], 1241, (0, 1000), to_host=True)
    m.search_init_dev_async(jobs, 100)
    fe.wait()
    m.search_init_dev_wait([len(want[s - 1][0]) for s in range(1, B)])
    assert fe.delivery_stats()[0] - sent0[0] == 2


def _handmade_frames(rng, n1, n2, nproto, q_flips, s_flips, area=60, chain=False):
    """keypoints crowded into one window, descriptors = a few prototypes with bit flips: most queries see several slots
    within TH_LOW and steal from each other; chain=True: every query is a little closer to slot 0 than the one before"""
    def kps(n):
        k = np.zeros(n, V.KP_DTYPE)
        k["x"] = 300 + rng.integers(0, area, n)
        k["y"] = 150 + rng.integers(0, area, n)
        k["angle"] = rng.integers(0, 360, n).astype(np.float32)
        k["size"] = 31.0
        k["octave"] = np.where(rng.random(n) < 0.1, 1, 0)
        return k

    def flip(d, nb):
        d = d.copy()
        bits = rng.choice(256, nb, replace=False)
        for b in bits:
            d[b >> 3] ^= 1 << (b & 7)
        return d

    proto = rng.integers(0, 256, (nproto, 32), dtype=np.uint8)
    k1, k2 = kps(n1), kps(n2)
    if chain:
        k1["octave"] = 0
        k2["octave"] = 0
        d2 = np.stack([proto[0]] + [np.bitwise_not(proto[0])] * (n2 - 1))
        d1 = np.stack([flip(proto[0], max(45 - i, 1)) for i in range(n1)])
    else:
        d2 = np.stack([flip(proto[rng.integers(nproto)], int(rng.integers(0, s_flips + 1))) for _ in range(n2)])
        d1 = np.stack([flip(proto[rng.integers(nproto)], int(rng.integers(0, q_flips + 1))) for _ in range(n1)])
    return k1, np.ascontiguousarray(d1), k2, np.ascontiguousarray(d2)


@pytest.mark.parametrize("topm", [0, 2, 16])
def test_search_init_replay_under_heavy_contention(topm):
    """The replay decides all undecided queries per round and commits out of query order: hand-made frames where nearly
    every query has several candidates within TH_LOW, long steal chains on one slot (more acceptances than a slot's entry
    list holds) and lists that run out -- vnMatches12 / vbPrevMatched / nmatches equal the sequential oracle."""
    rng = np.random.default_rng(77 + topm)
    f = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=1, tuning=dict(init_topm=topm) if topm else None)
    try:
        cases = [_handmade_frames(rng, 200, 30, 6, 45, 10), _handmade_frames(rng, 400, 120, 3, 40, 25),
                 _handmade_frames(rng, 150, 150, 40, 30, 5, area=200), _handmade_frames(rng, 60, 4, 1, 0, 0, chain=True),
                 _handmade_frames(rng, 300, 8, 2, 48, 2, area=20), _handmade_frames(rng, 1, 1, 1, 3, 0), _handmade_frames(rng, 250, 60, 2, 50, 50)]
        for ratio, ori in ((0.9, True), (0.6, False), (1.0, True)):
            m = V.FMatcher(f, ratio, ori)
            dev = [(_dev(c[1]), _dev(c[3])) for c in cases]
            pairs = [(c[0], dv[0].data_ptr(), c[2], dv[1].data_ptr(), np.stack([c[0]["x"], c[0]["y"]], 1)) for c, dv in zip(cases, dev)]
            out = m.SearchForInitializationBatch(pairs, 100)
            nacc = 0
            for j, c in enumerate(cases):
                wn, wm, wp = orbo.search_for_initialization(c[0], c[1], c[2], c[3], 1241, 376, window=100, nnratio=ratio, check_ori=ori)
                assert out[j][0] == wn and np.array_equal(out[j][1], wm) and np.array_equal(out[j][2], wp), (topm, ratio, ori, j)
                nacc += wn
            assert nacc > 20
            rounds, queries, npairs = m.search_init_replay_stats()
            assert npairs == len(cases) and queries > 900 and rounds >= npairs
    finally:
        f.close()


def test_stereo_step_leaves_in_one_transfer_and_the_block_region_has_one_owner():
    """A full stereo batch with host delivery: counts | keypoints | descriptors | mvuRight | mvDepth are one block and one
    transfer (plain launches and the replayed graph of pinned images); a SearchForInitialization on the same context then
    takes buffers of its own, and a context whose init matcher came first keeps the stereo outputs separate."""
    import torch
    W, H, bf, fx = 1241, 376, 386.1448, 718.856
    frames = [synth.make_stereo_pair(W, H, step=s) for s in range(2)]
    want = []
    for L, R in frames:
        eL, eR = orbo.Extractor(1500), orbo.Extractor(1500)
        kL, dL, _ = eL.compute(L)
        kR, dR, _ = eR.compute(R)
        wu, wd, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, bf, fx)
        want.append((kL, dL, kR, dR, wu, wd))
    dev = [torch.from_numpy(np.ascontiguousarray(im)).cuda() for pair in frames for im in pair]
    torch.cuda.synchronize()
    pin = V.PinnedImages(4, H, W, W)
    for i, im in enumerate([im for pair in frames for im in pair]):
        pin.array[i][:] = im

    def check(fe, expect_transfers):
        for rep, (ptrs, where) in enumerate([([t.data_ptr() for t in dev], V.IMGS_DEVICE), (pin.ptrs, V.IMGS_PINNED), (pin.ptrs, V.IMGS_PINNED)]):
            s0 = fe.delivery_stats()
            fe.frame_stereo_async(ptrs, W, bf, fx, where=where)
            feats, st = fe.frame_stereo_wait()
            assert fe.delivery_stats()[0] - s0[0] == expect_transfers, (rep, expect_transfers)
            for s, (kL, dL, kR, dR, wu, wd) in enumerate(want):
                assert all(np.array_equal(feats[2 * s][0][f], kL[f]) for f in kL.dtype.names), (rep, s)
                assert np.array_equal(feats[2 * s][1], dL) and np.array_equal(feats[2 * s + 1][1], dR), (rep, s)
                assert np.array_equal(st[s][0], wu) and np.array_equal(st[s][1], wd), (rep, s)

    def init_pair(fe):
        p, c = fe.slot_dev_ptrs(0), fe.slot_dev_ptrs(2)  # left image of frame 0 -> left image of frame 1
        m = V.FMatcher(fe, 0.9, True)
        m.search_init_dev_async([(p[0], p[1], p[2], c[0], c[1], c[2], 0)], 100)
        out = m.search_init_dev_wait([len(want[0][0])])
        wn, wm, _ = orbo.search_for_initialization(want[0][0], want[0][1], want[1][0], want[1][1], W, H, window=100, nnratio=0.9)
        assert out[0][0] == wn and np.array_equal(out[0][1], wm)

    try:
        a = V.FExtractor(1500, 1.2, 8, 20, 7, W, H, max_batch=4)
        try:
            check(a, 1)
            init_pair(a)   # the region belongs to the stereo outputs: the matcher allocates its own
            check(a, 1)
            init_pair(a)
        finally:
            a.close()
        b = V.FExtractor(1500, 1.2, 8, 20, 7, W, H, max_batch=4)
        try:
            b.compute_batch([im for pair in frames for im in pair])
            init_pair(b)   # init matcher first: it owns the region, stereo outputs travel on their own
            check(b, 2)
            init_pair(b)
        finally:
            b.close()
    finally:
        pin.close()


def test_search_init_replay_random_frames_many_seeds():
    """48 random hand-made problems (sizes, prototype counts, flip ranges, crowding drawn per case), windows centred on
    vbPrevMatched positions that differ from the keypoints' own: every output equals the sequential oracle."""
    rng = np.random.default_rng(2025)
    f = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=1)
    try:
        m = V.FMatcher(f, 0.9, True)
        total = 0
        for batch in range(3):
            cases, prevs = [], []
            for _ in range(16):
                n1, n2 = int(rng.integers(1, 210)), int(rng.integers(1, 210))
                c = _handmade_frames(rng, n1, n2, int(rng.integers(1, 12)), int(rng.integers(0, 60)), int(rng.integers(0, 40)),
                                     area=int(rng.integers(10, 250)))
                cases.append(c)
                prevs.append((np.stack([c[0]["x"], c[0]["y"]], 1) + rng.integers(-40, 41, (n1, 2))).astype(np.float32))
            dev = [(_dev(c[1]), _dev(c[3])) for c in cases]
            pairs = [(c[0], dv[0].data_ptr(), c[2], dv[1].data_ptr(), pv) for c, dv, pv in zip(cases, dev, prevs)]
            out = m.SearchForInitializationBatch(pairs, 100)
            for j, (c, pv) in enumerate(zip(cases, prevs)):
                wn, wm, wp = orbo.search_for_initialization(c[0], c[1], c[2], c[3], 1241, 376, prev_matched=pv, window=100, nnratio=0.9)
                assert out[j][0] == wn and np.array_equal(out[j][1], wm) and np.array_equal(out[j][2], wp), (batch, j)
                total += wn
        assert total > 100
    finally:
        f.close()
