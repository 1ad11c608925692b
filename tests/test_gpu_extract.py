"""GPU (-m gpu): the HIP extractor behind the C ABI vs the CPU oracle, bit-exact, stage by stage.
Mirrors the reference's only test idea: CPU implementation as oracle, exact compare."""
import os

import numpy as np
import pytest

import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

from conftest import kp_equal

pytestmark = pytest.mark.gpu


def _assert_same(res, ref, tag=""):
    k, d, m = res
    ko, do, mo = ref
    assert len(k) == len(ko), tag
    for f in k.dtype.names:
        assert np.array_equal(k[f], ko[f]), (tag, f)
    assert np.array_equal(d, do), tag
    assert m == mo, tag


@pytest.fixture(scope="module")
def kitti():
    fe = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=8)
    yield fe
    fe.close()


def test_tables_getters(kitti):
    t = orbo.Extractor(2000).tables()
    assert kitti.GetLevels() == 8
    assert np.array_equal(kitti.GetScaleFactors(), t["scale"])
    assert np.array_equal(kitti.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(kitti.GetScaleSigmaSquares(), t["sigma2"])
    assert np.array_equal(kitti.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert np.array_equal(kitti.features_per_level(), t["quota"])


def test_stagewise_parity_kitti(kitti):
    img = synth.make_frame(1241, 376)
    res = kitti.compute(img)
    e = orbo.Extractor(2000)
    ref = e.compute(img)
    for l in range(8):
        assert np.array_equal(kitti.mvImagePyramid(l), e.level(l)), ("pyramid", l)
        assert np.array_equal(kitti.mvImagePyramid(l, blurred=True), e.level(l, blurred=True)), ("blur", l)
        ca, cb = kitti.candidates(l), e.candidates(l)
        assert len(ca) == len(cb) and len(ca) > 100
        for f in ("x", "y", "response"):
            assert np.array_equal(ca[f], cb[f]), ("candidates", l, f)
    _assert_same(res, ref)
    assert len(res[0]) >= 2000


def test_lapping_area_mono_order(kitti):
    img = synth.make_frame(1241, 376, step=2)
    _assert_same(kitti.compute(img, (0, 1000)), orbo.Extractor(2000).compute(img, lap=(0, 1000)), "lap")


def test_batch_equals_single_and_oracle(kitti):
    imgs = [synth.make_frame(1241, 376, step=s, right=bool(s & 1)) for s in range(8)]
    res = kitti.compute_batch(imgs)
    e = orbo.Extractor(2000)
    for s in range(8):
        _assert_same(res[s], e.compute(imgs[s]), "slot %d" % s)
    # a second, different batch on the same context (stateful pyramids must be fully rewritten)
    imgs2 = [synth.make_frame(1241, 376, seed=99, step=s) for s in range(3)]
    res2 = kitti.compute_batch(imgs2)
    for s in range(3):
        _assert_same(res2[s], e.compute(imgs2[s]), "second batch %d" % s)


def test_device_resident_input_zero_copy(kitti):
    import torch
    imgs = [synth.make_frame(1241, 376, step=s) for s in range(2)]
    pitch = 1280
    dev = torch.zeros((2, 376, pitch), dtype=torch.uint8, device="cuda")
    for s in range(2):
        dev[s, :, :1241] = torch.from_numpy(imgs[s]).cuda()
    torch.cuda.synchronize()
    res = kitti.compute_batch(None, device_ptrs=[dev[s].data_ptr() for s in range(2)], pitch=pitch)
    e = orbo.Extractor(2000)
    for s in range(2):
        _assert_same(res[s], e.compute(imgs[s]), "zero-copy %d" % s)
        assert np.array_equal(kitti.mvImagePyramid(0, slot=s), imgs[s])


@pytest.mark.parametrize("cfg", [
    (1241, 376, 1000, 20, 7),    # BASELINE config 2
    (1920, 1080, 4000, 20, 7),   # BASELINE config 5
    (752, 480, 500, 20, 7),      # hut_stereo size
    (640, 480, 10000, 20, 7),    # 5 x nFeatures initialisation extractor (tracking.cpp:1093)
    (321, 203, 300, 35, 12),     # odd size, other thresholds
    (1920, 1080, 120, 20, 7),    # far more FAST cells per level than quadtree nodes (cell-offset scratch)
    (1241, 376, 60, 20, 7),
])
def test_other_geometries(cfg):
    w, h, nf, ini, mn = cfg
    img = synth.make_frame(w, h, seed=w + nf)
    fe = V.FExtractor(nf, 1.2, 8, ini, mn, w, h)
    try:
        _assert_same(fe.compute(img), orbo.Extractor(nf, ini_th=ini, min_th=mn).compute(img), str(cfg))
    finally:
        fe.close()


@pytest.mark.parametrize("cfg", [(1241, 376, 2000, 4, -1), (1920, 1080, 4000, 3, -1), (640, 360, 600, 2, 1), (752, 480, 5000, 2, 3)])
def test_quadtree_with_the_counting_walk_as_its_own_launch(cfg):
    """vslam_tuning.oct_precount = 1 (k_oct_count: a level's keys dealt to up to eight workgroups by rows of leaves, the
    quadtree kernel starts from the counters) and fast_kernel = 4 for small batches too: same keypoints and descriptors as the
    oracle, also with a forced shallow / deep fine grid (every node splits below it)"""
    w, h, nf, b, fd = cfg
    tn = dict(oct_precount=1, fast_kernel=4)
    if fd >= 0:
        tn["oct_fine_depth"] = fd
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, w, h, max_batch=b, tuning=tn)
    try:
        imgs = [synth.make_frame(w, h, seed=5, step=s) for s in range(b)]
        out = fe.compute_batch(imgs)
        e = orbo.Extractor(nf)
        for s in (0, b - 1):
            _assert_same(out[s], e.compute(imgs[s]), (cfg, s))
    finally:
        fe.close()


def test_low_texture_cells_fall_back_to_min_threshold():
    # faint texture only: no FAST corner at 20 anywhere, cells must rerun at 7 (fextractor.cpp:803-807)
    rng = np.random.default_rng(8)
    img = (128 + rng.integers(-9, 10, (240, 320))).astype(np.uint8)
    img[100:140, 150:200] += 60  # one strong structure: its cells use iniTh, the rest fall back
    fe = V.FExtractor(500, 1.2, 8, 20, 7, 320, 240)
    try:
        res = fe.compute(img)
        _assert_same(res, orbo.Extractor(500).compute(img), "fallback")
        assert (res[0]["response"] < 20).any() and (res[0]["response"] >= 20).any()
    finally:
        fe.close()


def test_flat_image_gives_no_keypoints():
    fe = V.FExtractor(500, 1.2, 8, 20, 7, 320, 240)
    try:
        k, d, m = fe.compute(np.full((240, 320), 100, np.uint8))
        assert len(k) == 0 and d.shape == (0, 32) and m == 0
    finally:
        fe.close()


def test_golden_real_image_crop(golden_dir):
    g = np.load(os.path.join(golden_dir, "pipeline_hut_320x240.npz"))
    fe = V.FExtractor(500, 1.2, 8, 20, 7, 320, 240)
    try:
        k, d, m = fe.compute(g["L"])
        assert kp_equal(k, g["kL"]) and np.array_equal(d, g["dL"])
        assert np.array_equal(fe.mvImagePyramid(3), g["lvl3"])
        assert np.array_equal(fe.mvImagePyramid(3, blurred=True), g["lvl3_blur"])
        k, d, m = fe.compute(g["L"], (0, 1000))
        assert kp_equal(k, g["kM"]) and np.array_equal(d, g["dM"]) and m == int(g["monoIndex"])
    finally:
        fe.close()


def test_knobs_fma_and_legacy_gauss_taps():
    img = synth.make_frame(640, 360, seed=5)
    fe = V.FExtractor(800, 1.2, 8, 20, 7, 640, 360, flags=V.FLAG_ATAN_FMA, gauss_taps=[18, 34, 49, 55, 49, 34, 18])
    try:
        ref = orbo.Extractor(800, taps=[18, 34, 49, 55, 49, 34, 18], atan_fma=1).compute(img)
        _assert_same(fe.compute(img), ref, "knobs")
    finally:
        fe.close()


def test_asymmetric_gauss_taps_take_the_general_column_pass():
    """k_blur7_v2 folds mirrored rows when the taps are symmetric; any other kernel (sum 256) takes the 7-product form."""
    img = synth.make_frame(640, 360, seed=6)
    taps = [10, 30, 50, 60, 52, 36, 18]
    fe = V.FExtractor(800, 1.2, 8, 20, 7, 640, 360, gauss_taps=taps)
    try:
        _assert_same(fe.compute(img), orbo.Extractor(800, taps=taps).compute(img), "asymmetric taps")
    finally:
        fe.close()


def test_device_trig_is_glibc_exact(kitti):
    """Every float the rotation can see: angle(deg in [0,360]) * (float)(pi/180) plus a dense sweep."""
    rng = np.random.default_rng(0)
    deg = np.concatenate([np.linspace(0, 360, 200001, dtype=np.float32),
                          rng.uniform(0, 360, 300000).astype(np.float32)])
    x = deg * np.float32(np.pi / 180.0)
    bits = rng.integers(0, np.float32(6.5).view(np.uint32), 500000, dtype=np.uint32).view(np.float32)
    x = np.concatenate([x, bits])
    s, c = V.dbg_sincos(kitti, x)
    L = orbo.lib()
    want_s = np.array([L.orbo_sinf(float(v)) for v in x[:20000]], np.float32)
    want_c = np.array([L.orbo_cosf(float(v)) for v in x[:20000]], np.float32)
    assert np.array_equal(s[:20000], want_s) and np.array_equal(c[:20000], want_c)
    # numpy's float32 sin/cos call the same libm on this platform only loosely; check the rest to 1 ulp
    assert np.all(np.abs(s - np.sin(x.astype(np.float64))) <= np.spacing(np.abs(s)) + 1e-45)
    assert np.all(np.abs(c - np.cos(x.astype(np.float64))) <= np.spacing(np.abs(c)) + 1e-45)


def test_device_fast_atan2_equals_oracle(kitti):
    rng = np.random.default_rng(1)
    y = rng.integers(-2_900_000, 2_900_000, 200000).astype(np.float32)
    x = rng.integers(-2_900_000, 2_900_000, 200000).astype(np.float32)
    y[:10] = 0
    x[5:15] = 0
    for fma in (0, 1):
        a = V.dbg_fast_atan2(kitti, y, x, fma)
        want = np.array([orbo.fast_atan2(float(yy), float(xx), fma) for yy, xx in zip(y[:30000], x[:30000])],
                        np.float32)
        assert np.array_equal(a[:30000], want)
        assert np.all((a >= 0) & (a <= 360))


def test_errors(kitti):
    with pytest.raises(V.VslamError):
        kitti.compute(np.zeros((100, 100), np.uint8))
    with pytest.raises(V.VslamError):
        kitti.compute_batch([np.zeros((376, 1241), np.uint8)] * 9)  # > max_batch
    with pytest.raises(V.VslamError) as ei:
        V.FExtractor(500, 1.2, 8, 20, 7, 200, 600)  # portrait: nIni == 0 in the reference
    assert ei.value.code == V.ERR_UNSUPPORTED


def test_host_quadtree_fallback_path_matches_too():
    """VSLAM_FLAG_HOST_OCTREE keeps DistributeOctTree on the host worker pool (the path taken when a
    level's node list cannot fit LDS); both placements must give the reference's result."""
    imgs = [synth.make_frame(1241, 376, step=s) for s in range(3)]
    fe = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=3, flags=V.FLAG_HOST_OCTREE)
    try:
        res = fe.compute_batch(imgs, (0, 1000))
        e = orbo.Extractor(2000)
        for s in range(3):
            _assert_same(res[s], e.compute(imgs[s], lap=(0, 1000)), "host quadtree %d" % s)
    finally:
        fe.close()


def test_async_split_api_two_contexts_in_flight():
    import torch
    imgs = [synth.make_frame(1241, 376, step=s) for s in range(4)]
    pitch = 1280
    dev = torch.zeros((4, 376, pitch), dtype=torch.uint8, device="cuda")
    for s in range(4):
        dev[s, :, :1241] = torch.from_numpy(imgs[s]).cuda()
    torch.cuda.synchronize()
    a = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
    b = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
    try:
        a.compute_batch_async([dev[0].data_ptr(), dev[1].data_ptr()], pitch, (0, 1000))
        b.compute_batch_async([dev[2].data_ptr(), dev[3].data_ptr()], pitch, (0, 1000))
        ra = a.wait(copy=True)
        rb = b.wait(copy=True)
        e = orbo.Extractor(1000)
        for s, r in enumerate(ra + rb):
            _assert_same(r, e.compute(imgs[s], lap=(0, 1000)), "async %d" % s)
    finally:
        a.close()
        b.close()


def test_large_feature_count_uses_a_big_node_list():
    # 5 x nFeatures initialisation extractor at KITTI size: level-0 quota 2170 -> ~2.2k list entries in LDS
    img = synth.make_frame(1241, 376, seed=21)
    fe = V.FExtractor(10000, 1.2, 8, 20, 7, 1241, 376)
    try:
        _assert_same(fe.compute(img, (0, 1000)), orbo.Extractor(10000).compute(img, lap=(0, 1000)), "n10000")
    finally:
        fe.close()


def test_two_host_threads_two_contexts_like_frame_ctor():
    """Frame::Frame(stereo) runs the left and right extractor on two std::threads (frame.cpp:107-111): two
    contexts driven concurrently from two host threads must not disturb each other."""
    import threading
    L, R = synth.make_stereo_pair(752, 480, seed=5)
    fl = V.FExtractor(800, 1.2, 8, 20, 7, 752, 480)
    fr = V.FExtractor(800, 1.2, 8, 20, 7, 752, 480)
    try:
        wl, wr = orbo.Extractor(800).compute(L), orbo.Extractor(800).compute(R)
        out = {}

        def run(name, fe, img, n):
            res = []
            for _ in range(n):
                k, d, m = fe.compute(img)
                res.append((k.copy(), d.copy(), m))
            out[name] = res

        ta = threading.Thread(target=run, args=("l", fl, L, 25))
        tb = threading.Thread(target=run, args=("r", fr, R, 25))
        ta.start(); tb.start(); ta.join(); tb.join()
        for r in out["l"]:
            _assert_same(r, wl, "left thread")
        for r in out["r"]:
            _assert_same(r, wr, "right thread")
    finally:
        fl.close()
        fr.close()


def test_randomised_geometries_and_parameters():
    """Seeded sweep over image sizes, pyramid shapes, thresholds and feature counts (exercises the kernel
    generation fallbacks: wide FAST cells, coarse scale factors, few levels); everything bit-exact vs the oracle."""
    rng = np.random.default_rng(20250215)
    done = 0
    for trial in range(40):
        w = int(rng.integers(200, 1400))
        h = int(rng.integers(150, min(w, 800)))       # landscape (nIni >= 1)
        nlevels = int(rng.integers(3, 9))
        sf = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
        nf = int(rng.choice([80, 300, 1000, 2500]))
        ini = int(rng.integers(10, 40))
        mn = int(rng.integers(3, ini + 1))
        try:
            fe = V.FExtractor(nf, sf, nlevels, ini, mn, w, h)
        except V.VslamError:
            continue  # geometry rejected at create (smallest level too small): the reference would misbehave too
        try:
            img = synth.make_frame(w, h, seed=1000 + trial)
            lap = (0, 0) if trial & 1 else (0, int(rng.integers(0, w)))
            ref = orbo.Extractor(nf, scale=sf, nlevels=nlevels, ini_th=ini, min_th=mn).compute(img, lap=lap)
            _assert_same(fe.compute(img, lap), ref, str((w, h, nlevels, sf, nf, ini, mn, lap)))
            done += 1
        finally:
            fe.close()
    assert done >= 25


def test_host_image_with_row_padding_and_context_churn():
    """Host images whose rows are padded (pitch > width) go through the pinned staging path row by row; creating and
    destroying many contexts must neither fail nor disturb results."""
    import ctypes as C
    w, h = 500, 300
    img = synth.make_frame(w, h, seed=77)
    padded = np.zeros((h, 640), np.uint8)
    padded[:, :w] = img
    padded[:, w:] = 255  # garbage beyond the row end must never be read as image
    ref = orbo.Extractor(400).compute(img)
    for round_ in range(12):
        fe = V.FExtractor(400, 1.2, 8, 20, 7, w, h)
        try:
            cap = fe.cap
            kps = np.zeros(cap, V.KP_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            n, mono = C.c_int(), C.c_int()
            rc = V.lib().vslam_fe_extract(fe._h, padded.ctypes.data_as(C.c_void_p), C.c_size_t(640), 0, 0,
                                          kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                          C.byref(n), C.byref(mono))
            assert rc == 0
            _assert_same((kps[:n.value], desc[:n.value], mono.value), ref, "padded pitch, round %d" % round_)
        finally:
            fe.close()


def test_dense_and_padded_host_images_alternate_on_one_context():
    """Dense host images (pitch == width) are staged densely -- one copy per image, their own captured graph -- padded
    ones row by row; alternating between the two on ONE context (single image and a batch, which takes the DMA route)
    must replay the right graph each time.  Odd width: dense rows start at unaligned addresses."""
    import ctypes as C
    w, h = 501, 300
    imgs = [synth.make_frame(w, h, seed=80 + i) for i in range(4)]
    refs = [orbo.Extractor(400).compute(im) for im in imgs]
    fe = V.FExtractor(400, 1.2, 8, 20, 7, w, h, max_batch=4)
    try:
        cap = fe.cap
        for rep in range(3):
            for pitch in (w, 640, w, 1024):
                for nimg in (1, 4):
                    buf = np.full((nimg, h, pitch), 255, np.uint8)
                    for s in range(nimg):
                        buf[s, :, :w] = imgs[s]
                    ptrs = (C.c_void_p * nimg)(*[buf[s].ctypes.data for s in range(nimg)])
                    kps = np.zeros((nimg, cap), V.KP_DTYPE)
                    desc = np.zeros((nimg, cap, 32), np.uint8)
                    n, mono = (C.c_int * nimg)(), (C.c_int * nimg)()
                    kp = (C.c_void_p * nimg)(*[kps[s].ctypes.data for s in range(nimg)])
                    dp = (C.c_void_p * nimg)(*[desc[s].ctypes.data for s in range(nimg)])
                    rc = V.lib().vslam_fe_extract_batch(fe._h, nimg, ptrs, C.c_size_t(pitch), V.IMGS_HOST, 0, 0, kp, dp, cap, n, mono)
                    assert rc == 0, V.lib().vslam_last_error()
                    for s in range(nimg):
                        _assert_same((kps[s, :n[s]], desc[s, :n[s]], mono[s]), refs[s], "pitch %d, %d images, image %d, rep %d" % (pitch, nimg, s, rep))
    finally:
        fe.close()


def test_small_geometries_are_rejected_or_exact():
    """Tiny images / few levels: either vslam_fe_create refuses the geometry (no 30-px FAST cell fits anywhere -- the
    reference divides by that count) or the result equals the oracle; never a failed launch."""
    ok = rejected = 0
    for (w, h) in [(64, 48), (80, 64), (100, 70), (128, 96), (160, 120), (97, 97), (200, 60), (400, 100), (70, 66)]:
        for nl in (1, 2, 4, 8):
            for nf in (20, 200):
                try:
                    fe = V.FExtractor(nf, 1.2, nl, 20, 7, w, h)
                except V.VslamError as e:
                    assert e.code in (-5, -1), e  # VSLAM_ERR_UNSUPPORTED / INVALID, with a message
                    rejected += 1
                    continue
                try:
                    img = synth.make_frame(w, h, seed=w * 7 + h)
                    _assert_same(fe.compute(img), orbo.Extractor(nf, nlevels=nl).compute(img), str((w, h, nl, nf)))
                    ok += 1
                finally:
                    fe.close()
    assert ok >= 30 and rejected >= 20


@pytest.mark.parametrize("nf", [1000, 5000])
def test_adversarial_patterns(nf):
    """Checkerboards (massive score and size ties in the quadtree), binary noise (densest candidates), stripes and
    gradients (no corners at all), isolated dots: bit-exact vs the oracle."""
    W, H = 752, 480
    yy, xx = np.mgrid[0:H, 0:W]
    rng = np.random.default_rng(3)
    imgs = {
        "checker2": (((xx // 2 + yy // 2) & 1) * 255).astype(np.uint8),
        "checker3": (((xx // 3 + yy // 3) & 1) * 255).astype(np.uint8),
        "checker5": (((xx // 5 + yy // 5) & 1) * 255).astype(np.uint8),
        "checker16": (((xx // 16 + yy // 16) & 1) * 255).astype(np.uint8),
        "binary_noise": (rng.integers(0, 2, (H, W)) * 255).astype(np.uint8),
        "salt": np.where(rng.random((H, W)) < 0.02, 255, 0).astype(np.uint8),
        "stripes_v": ((xx // 4 & 1) * 200 + 20).astype(np.uint8),
        "gradient": ((xx * 255) // W).astype(np.uint8),
        "dots": np.where(((xx % 7) == 3) & ((yy % 7) == 3), 255, 30).astype(np.uint8),
        "ramp_noise": np.clip((xx * 255) // W + rng.integers(-25, 26, (H, W)), 0, 255).astype(np.uint8),
    }
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, W, H)
    try:
        e = orbo.Extractor(nf)
        for name, im in imgs.items():
            _assert_same(fe.compute(im), e.compute(im), name)
    finally:
        fe.close()
