"""The reference's own test images at FULL size (hut_stereo/01-05.png 752x480, lenna.png 512x512, chessboard_798_798.png --
the tie-heavy one) through the whole path: extraction at N in {1200, 2000, 5 x 1000}, ComputeStereoMatches and
SearchForInitialization.  tests/golden/real_images.npz holds the gray images, real_expected.npz the oracle's outputs
(generator: tests/golden/make_golden.py real; the reference's PNGs exist only in the build container).

CPU (-m "not gpu"): the oracle reproduces the committed outputs.
GPU (-m gpu): the HIP path equals the oracle AND the committed outputs bit for bit, in batches, with the quadtree statistics
(vslam_fe_octree_stats: on how many (slot, level) problems nodes had to be split below the kernel's fine grid) reported
per case."""
import hashlib
import os

import numpy as np
import pytest

from oracle import orbo

HERE = os.path.dirname(os.path.abspath(__file__))
BF, FX = 822.5 * 0.4, 822.5
CASES = [("hut1", 1200, (0, 0)), ("hut2", 1200, (0, 0)), ("hut3", 2000, (0, 0)), ("hut4", 2000, (0, 0)), ("hut5", 2000, (0, 0)),
         ("lenna", 1200, (0, 0)), ("lenna", 2000, (0, 1000)), ("chess", 1200, (0, 0)), ("chess", 2000, (0, 1000)),
         ("hut1", 5000, (0, 1000)), ("hut2", 5000, (0, 1000)), ("hut4", 5000, (0, 1000)), ("hut5", 5000, (0, 1000))]
STEREO_PAIRS = [("hut1", "hut2", 1200), ("hut3", "hut4", 2000), ("hut4", "hut5", 2000)]
INIT_PAIRS = [("hut1", "hut2", 5000), ("hut4", "hut5", 5000)]


def _key(name, nf, lap):
    return "%s_n%d_lap%d_%d" % (name, nf, lap[0], lap[1])


@pytest.fixture(scope="module")
def real():
    ims = np.load(os.path.join(HERE, "golden", "real_images.npz"))
    exp = np.load(os.path.join(HERE, "golden", "real_expected.npz"))
    return {k: ims[k] for k in ims.files}, exp


def _sha(d):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(d).tobytes()).digest(), np.uint8)


def _check_against_golden(exp, key, k, d, mono):
    g = exp[key + "_kps"]
    assert len(k) == len(g), key
    for f in g.dtype.names:
        assert np.array_equal(k[f], g[f]), (key, f)
    assert np.array_equal(_sha(d), exp[key + "_desc_sha256"]), key
    assert int(mono) == int(exp[key + "_mono"]), key


def test_fixture_shapes(real):
    ims, _ = real
    assert ims["hut1"].shape == (480, 752) and ims["lenna"].shape == (512, 512) and ims["chess"].shape == (798, 798)
    assert all(v.dtype == np.uint8 for v in ims.values())


@pytest.mark.parametrize("case", CASES, ids=lambda c: _key(*c))
def test_oracle_reproduces_the_committed_outputs(real, case):
    ims, exp = real
    name, nf, lap = case
    e = orbo.Extractor(nf)
    k, d, m = e.compute(ims[name], lap=lap)
    _check_against_golden(exp, _key(*case), k, d, m)
    assert np.array_equal(np.asarray([len(e.candidates(l)) for l in range(8)], np.int32), exp[_key(*case) + "_ncand"])


def test_oracle_stereo_and_init_on_real_pairs(real):
    ims, exp = real
    for a, b, nf in STEREO_PAIRS:
        eL, eR = orbo.Extractor(nf), orbo.Extractor(nf)
        kL, dL, _ = eL.compute(ims[a])
        kR, dR, _ = eR.compute(ims[b])
        u, dep, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, BF, FX)
        assert np.array_equal(u, exp["stereo_%s_%s_n%d_uRight" % (a, b, nf)])
        assert np.array_equal(dep, exp["stereo_%s_%s_n%d_depth" % (a, b, nf)])
    for a, b, nf in INIT_PAIRS:
        e = orbo.Extractor(nf)
        k1, d1, _ = e.compute(ims[a], lap=(0, 1000))
        k2, d2, _ = e.compute(ims[b], lap=(0, 1000))
        nm, m12, _ = orbo.search_for_initialization(k1, d1, k2, d2, 752, 480, window=100)
        assert nm == int(exp["init_%s_%s_n%d_nmatches" % (a, b, nf)]) and nm > 100
        assert np.array_equal(m12, exp["init_%s_%s_n%d_matches" % (a, b, nf)])


# ------------------------------------------------------------------------------------------------ GPU


def _same(res, ko, do, tag):
    k, d = res[0], res[1]
    assert len(k) == len(ko), tag
    for f in k.dtype.names:
        assert np.array_equal(k[f], ko[f]), (tag, f)
    assert np.array_equal(d, do), tag


@pytest.mark.gpu
@pytest.mark.parametrize("nf,lap", [(1200, (0, 0)), (2000, (0, 0)), (2000, (0, 1000)), (5000, (0, 1000))])
def test_gpu_hut_batch_equals_oracle_and_golden(real, nf, lap):
    """all five 752x480 frames in one batch (the many-slot quadtree instantiation), every committed case among them"""
    import vi_slam_amd as V
    ims, exp = real
    names = ["hut%d" % i for i in range(1, 6)]
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, 752, 480, max_batch=5)
    try:
        res = fe.compute_batch([ims[n] for n in names], lap)
        prob, handed, masks = fe.octree_stats()
        print("\nquadtree hut x5 N=%d: %d problems, %d split below the grid, level masks %s" % (nf, prob, handed, [hex(m) for m in masks[:5]]))
        assert prob == 5 * 8
        for s, n in enumerate(names):
            e = orbo.Extractor(nf)
            ko, do, mo = e.compute(ims[n], lap=lap)
            _same(res[s], ko, do, "%s N=%d" % (n, nf))
            assert res[s][2] == mo
            if (n, nf, lap) in CASES:
                _check_against_golden(exp, _key(n, nf, lap), res[s][0], res[s][1], res[s][2])
    finally:
        fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,nf,lap", [c for c in CASES if c[0] in ("lenna", "chess")], ids=lambda v: str(v))
def test_gpu_lenna_and_chessboard_equal_oracle_and_golden(real, name, nf, lap):
    """single images (the keys-in-registers quadtree instantiation); the chessboard has NO level-0 corner and fewer
    candidates than the quota on every level, so every node is split down to single keys"""
    import vi_slam_amd as V
    ims, exp = real
    im = ims[name]
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, im.shape[1], im.shape[0], max_batch=2)
    try:
        for batch in (1, 2):
            res = fe.compute_batch([im] * batch, lap)
            prob, handed, masks = fe.octree_stats()
            print("\nquadtree %s N=%d batch %d: %d problems, %d split below the grid so far, masks %s" % (name, nf, batch, prob, handed, [hex(m) for m in masks[:batch]]))
            for s in range(batch):
                _check_against_golden(exp, _key(name, nf, lap), res[s][0], res[s][1], res[s][2])
        e = orbo.Extractor(nf)
        ko, do, _ = e.compute(im, lap=lap)
        _same(res[0], ko, do, "%s N=%d" % (name, nf))
    finally:
        fe.close()


@pytest.mark.gpu
def test_gpu_real_stereo_pairs_and_init_matches(real):
    import vi_slam_amd as V
    ims, exp = real
    for a, b, nf in STEREO_PAIRS:
        fe = V.FExtractor(nf, 1.2, 8, 20, 7, 752, 480, max_batch=2)
        try:
            fe.compute_batch([ims[a], ims[b]])
            u, d = V.ComputeStereoMatches(fe, 0, fe, 1, BF, FX)
            assert np.array_equal(u, exp["stereo_%s_%s_n%d_uRight" % (a, b, nf)]), (a, b, nf)
            assert np.array_equal(d, exp["stereo_%s_%s_n%d_depth" % (a, b, nf)]), (a, b, nf)
            assert (u >= 0).sum() > 20
        finally:
            fe.close()
    for a, b, nf in INIT_PAIRS:
        fe = V.FExtractor(nf, 1.2, 8, 20, 7, 752, 480, max_batch=2)
        try:
            (k1, d1, _), (k2, d2, _) = fe.compute_batch([ims[a], ims[b]], (0, 1000))
            p, c = fe.slot_dev_ptrs(0), fe.slot_dev_ptrs(1)
            m = V.FMatcher(fe, 0.9, True)
            m.search_init_dev_async([(p[0], p[1], p[2], c[0], c[1], c[2], 0)], 100)
            out = m.search_init_dev_wait([len(k1)])
            assert out[0][0] == int(exp["init_%s_%s_n%d_nmatches" % (a, b, nf)])
            assert np.array_equal(out[0][1], exp["init_%s_%s_n%d_matches" % (a, b, nf)])
        finally:
            fe.close()
