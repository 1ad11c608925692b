"""CPU: the product's host-side logic (vi_slam_amd/csrc/vslam_host.cpp, built GPU-free into
libvslam_host.so) against the oracle, and the C ABI library's export table."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    L = C.CDLL(V.HOST_LIB_PATH)
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def host_octree(H, xyr, W, Hh, N):
    xyr = np.ascontiguousarray(xyr, np.int32).reshape(-1, 3)
    out = np.zeros((len(xyr) + 8, 3), np.int32)
    n = H.vslamh_octree(_p(xyr), len(xyr), W, Hh, N, _p(out), len(out))
    assert n >= 0
    return out[:n]


@pytest.mark.parametrize("nf", [500, 1000, 2000, 4000, 10000])
def test_tables_equal_oracle(H, nf):
    sf, isf, s2, is2 = (np.zeros(8, np.float32) for _ in range(4))
    q = np.zeros(8, np.int32)
    um = np.zeros(16, np.int32)
    dn = C.c_int()
    H.vslamh_tables(nf, C.c_float(1.2), 8, _p(sf), _p(isf), _p(s2), _p(is2), _p(q), _p(um), C.byref(dn))
    t = orbo.Extractor(nf).tables()
    assert np.array_equal(sf, t["scale"]) and np.array_equal(isf, t["inv_scale"])
    assert np.array_equal(s2, t["sigma2"]) and np.array_equal(is2, t["inv_sigma2"])
    assert np.array_equal(q, t["quota"]) and np.array_equal(um, t["umax"])
    assert dn.value == 749  # pixels of the radius-15 disc


@pytest.mark.parametrize("size", [(1241, 376), (1920, 1080), (752, 480), (321, 203)])
def test_resize_tables_reproduce_oracle_resize(H, size):
    w, h = size
    e = orbo.Extractor(1000)
    img = synth.make_frame(w, h, seed=w)
    e.pyramid_only(img)
    for l in range(1, 8):
        src, want = e.level(l - 1), e.level(l)
        lw, lh = C.c_int(), C.c_int()
        H.vslamh_level_size(1000, C.c_float(1.2), 8, w, h, l, C.byref(lw), C.byref(lh))
        assert (lw.value, lh.value) == want.shape[::-1]
        got = np.zeros_like(want)
        H.vslamh_resize_with_tables(_p(src), src.shape[1], src.shape[0], C.c_size_t(src.shape[1]), _p(got),
                                    lw.value, lh.value, C.c_size_t(lw.value))
        assert np.array_equal(got, want)


def test_cell_grid_counts_kitti(H):
    # executed cells per level after the skip rules of fextractor.cpp:785,794
    sizes = [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    counts = []
    for l, (w, h) in enumerate(sizes):
        out = np.zeros((4096, 5), np.uint16)
        n = H.vslamh_cells(l, w, h, _p(out), 4096)
        counts.append(n)
        c = out[:n]
        # interiors (window minus the 3-px ring) tile [19, w-19) x [19, h-19) exactly
        cover = np.zeros((h, w), np.int32)
        for _, x0, y0, x1, y1 in c:
            cover[y0 + 3:y1 - 3, x0 + 3:x1 - 3] += 1
        assert np.all(cover[19:h - 19, 19:w - 19] == 1)
        cover[19:h - 19, 19:w - 19] = 0
        assert not cover.any()
    assert counts == [429, 297, 189, 132, 72, 45, 36, 20] and sum(counts) == 1220


@pytest.mark.parametrize("size", [(1241, 376), (1920, 1080), (752, 480), (640, 360), (512, 512), (179, 100), (143, 143), (95, 70)])
@pytest.mark.parametrize("per", [1, 2, 3, 4])
def test_bands_cover_every_cell_once(H, size, per):
    """k_fast_bands' work list (vslam::build_bands): consecutive cells of one cell row, constant pitch, interiors of a
    band at most 128 px wide, every cell in exactly one band, band windows = the union of their cells' windows"""
    w0, h0 = size
    for l in range(8):
        w, h = int(round(w0 / 1.2 ** l)), int(round(h0 / 1.2 ** l))
        cells = np.zeros((8192, 5), np.uint16)
        nc = H.vslamh_cells(l, w, h, _p(cells), 8192)
        bands = np.zeros((8192, 8), np.uint32)
        maxw = 128  # k_fast_bands' interior columns (vslam_fe.hip)
        nb = H.vslamh_bands(l, w, h, per, maxw, _p(bands), 8192)
        if nc == 0:
            assert nb == 0
            continue
        cells, bands = cells[:nc].astype(int), bands[:nb].astype(int)
        seen = np.zeros(nc, int)
        for cell0, lev, ncell, wcell, x0, y0, ww, wh in bands:
            assert 1 <= ncell <= per and lev == l and ww - 6 <= maxw
            for k in range(ncell):
                c = cells[cell0 + k]
                seen[cell0 + k] += 1
                assert c[2] == y0 and c[4] == y0 + wh            # one cell row
                assert c[1] == x0 + k * wcell                    # constant pitch
                # cell k's interior columns inside the band's interior: [k * wcell, min((k + 1) * wcell, iw))
                assert c[3] - c[1] - 6 == min((k + 1) * wcell, ww - 6) - k * wcell > 0
            assert cells[cell0 + ncell - 1][3] == x0 + ww
            assert (wcell * ((65536 + wcell - 1) // wcell)) >> 16 >= 1  # the kernel's x / wcell by multiplication ...
            rc = (65536 + wcell - 1) // wcell
            assert all((x * rc) >> 16 == x // wcell for x in range(256))   # ... is exact on its domain
        assert np.all(seen == 1)
        assert np.all(np.diff(bands[:, 0]) > 0)


def test_octree_equals_oracle_on_extractor_candidates(H):
    for (w, h, nf) in [(1241, 376, 2000), (1241, 376, 1000), (752, 480, 500), (640, 480, 10000)]:
        e = orbo.Extractor(nf)
        e.compute(synth.make_frame(w, h, seed=7 + w))
        q = e.tables()["quota"]
        for l in range(8):
            c = e.candidates(l)
            lw, lh = e.level(l).shape[::-1]
            xyr = np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)
            r = host_octree(H, xyr, lw - 32, lh - 32, int(q[l]))
            o = orbo.distribute_octree(c, 16, lw - 16, 16, lh - 16, int(q[l]))
            O = np.stack([o["x"], o["y"], o["response"]], 1).astype(np.int32).reshape(-1, 3)
            assert np.array_equal(O, r), (w, h, nf, l)


@pytest.mark.parametrize("W,Hh", [(1209, 344), (1002, 281), (314, 73), (1888, 1048), (720, 448), (178, 102), (766, 766), (65, 40)])
def test_quadtree_path_tables_equal_the_literal_halvings(H, W, Hh):
    """k_octree_v4 finds a key's leaf of the implicit quadtree as xs[x] | ys[y] (vslam::build_oct_lut) instead of walking D
    halvings per key: every (x, y) of the level, depths 1..7, against DivideNode's arithmetic spelled out (oct_key_path)"""
    for D in range(1, 8):
        assert H.vslamh_oct_lut_check(W, Hh, D) == 0, D
    # the literal walk itself: two keys in the same leaf at depth d share every shallower leaf
    rng = np.random.default_rng(4)
    H.vslamh_oct_key_path.restype = C.c_uint
    for _ in range(200):
        x, y = int(rng.integers(0, W)), int(rng.integers(0, Hh))
        prev = None
        for d in range(0, 13):
            pth = H.vslamh_oct_key_path(x, y, W, Hh, d)
            if prev is not None:
                assert pth >> 2 == prev
            prev = pth


def test_octree_random_and_edge_cases(H):
    rng = np.random.default_rng(9)
    assert len(host_octree(H, np.zeros((0, 3)), 1209, 344, 100)) == 0
    assert H.vslamh_octree(None, 0, 100, 400, 10, None, 0) == -1  # nIni == 0 (portrait): unsupported
    for trial in range(40):
        W, Hh = int(rng.integers(60, 1900)), int(rng.integers(40, 400))
        if round(W / Hh) < 1:
            continue
        n = int(rng.integers(1, 3000))
        N = int(rng.integers(1, 600))
        pts = np.unique(np.stack([rng.integers(0, W, n), rng.integers(0, Hh, n)], 1), axis=0)
        rng.shuffle(pts)
        # few distinct responses -> many ties: exercises first-wins and the equal-size node tie-break
        resp = rng.integers(7, 12 if trial % 2 else 250, len(pts))
        k = np.zeros(len(pts), orbo.KP_DTYPE)
        k["x"], k["y"], k["response"] = pts[:, 0], pts[:, 1], resp
        o = orbo.distribute_octree(k, 16, 16 + W, 16, 16 + Hh, N)
        O = np.stack([o["x"], o["y"], o["response"]], 1).astype(np.int32).reshape(-1, 3)
        r = host_octree(H, np.column_stack([pts, resp]), W, Hh, N)
        assert np.array_equal(O, r), trial


def test_grid_query_equals_oracle(H):
    e = orbo.Extractor(1000)
    k, d, _ = e.compute(synth.make_frame(1241, 376), lap=(0, 1000))
    rng = np.random.default_rng(3)
    for _ in range(50):
        x, y = float(rng.uniform(-50, 1300)), float(rng.uniform(-50, 420))
        r = float(rng.choice([10, 50, 100]))
        lv = int(rng.integers(-1, 3))
        want = orbo.grid_query(k, 1241, 376, x, y, r, max(lv, 0), lv)
        out = np.zeros(len(k) + 1, np.int32)
        n = H.vslamh_grid_query(_p(k), len(k), 1241, 376, C.c_float(x), C.c_float(y), C.c_float(r),
                                max(lv, 0), lv, _p(out), len(out))
        assert np.array_equal(out[:n], want)


def test_search_init_replay_equals_oracle(H):
    e = orbo.Extractor(1000)
    k1, d1, _ = e.compute(synth.make_frame(1241, 376, step=0), lap=(0, 1000))
    k2, d2, _ = e.compute(synth.make_frame(1241, 376, step=1), lap=(0, 1000))
    dm = np.minimum(orbo.hamming_matrix(d1, d2), 255).astype(np.uint8)
    for window, ratio, ori in [(100, 0.9, 1), (30, 0.9, 0), (100, 0.6, 1)]:
        nm_o, m_o, pm_o = orbo.search_for_initialization(k1, d1, k2, d2, 1241, 376, window=window,
                                                         nnratio=ratio, check_ori=bool(ori))
        pm = np.stack([k1["x"], k1["y"]], 1).astype(np.float32).copy()
        m = np.zeros(len(k1), np.int32)
        nm = H.vslamh_search_init(_p(k1), len(k1), _p(k2), len(k2), _p(dm), 1241, 376, _p(pm), _p(m), window,
                                  C.c_float(ratio), ori)
        assert nm == nm_o and np.array_equal(m, m_o) and np.array_equal(pm, pm_o)
        assert nm > 20


# ---------------------------------------------------------------- C ABI library (no GPU needed)
def _declared_functions(header="vslam_fe.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vslam_\w+)\s*\(", src)))


def test_header_and_python_mirror_agree():
    assert _declared_functions() == sorted(V.ABI_SYMBOLS)


def test_product_library_exports_every_declared_symbol():
    assert os.path.exists(V.LIB_PATH), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    L = C.CDLL(V.LIB_PATH)
    for name in _declared_functions() + _declared_functions("vslam_fastgrid.h"):
        assert hasattr(L, name), name


def test_fastgrid_create_rejects_bad_parameters_and_has_no_cpu_fallback():
    """include/vslam_fastgrid.h: parameter checks mirror the reference's assertions (fast_gpu.cpp:75,
    detector_base_gpu.cpp:61-62, pyramid_pool.cpp:58-59); without a device the constructor fails."""
    import torch
    from vi_slam_amd import fastgrid
    assert _declared_functions("vslam_fastgrid.h") == sorted(
        ["vslam_fg_create", "vslam_fg_destroy", "vslam_fg_grid", "vslam_fg_detect", "vslam_fg_detect_batch",
         "vslam_fg_level_copy", "vslam_fg_response_copy"])
    for bad in (dict(cell_size_width=48), dict(min_arc_length=8), dict(min_arc_length=13), dict(max_level=0), dict(score=3),
                dict(max_level=3, image_width=130), dict(max_batch=0), dict(threshold=-1.0)):
        kw = dict(image_width=128, image_height=64)
        kw.update(bad)
        with pytest.raises(V.VslamError) as ei:
            fastgrid.FASTGPU(**kw)
        assert ei.value.code == V.ERR_INVALID
    if torch.cuda.device_count() == 0:
        with pytest.raises(V.VslamError):
            fastgrid.FASTGPU(128, 64)


def test_create_rejects_bad_parameters_and_missing_gpu():
    import torch
    L = V.lib()
    h = C.c_void_p()
    bad = V._Params(0, 376, 1000, 1.2, 8, 20, 7, 0, 1, 0, (C.c_int32 * 7)(*[0] * 7))
    assert L.vslam_fe_create(C.byref(bad), C.byref(h)) == V.ERR_INVALID and not h.value
    bad = V._Params(1241, 376, 1000, 1.2, 8, 20, 7, 0, 99, 0, (C.c_int32 * 7)(*[0] * 7))
    assert L.vslam_fe_create(C.byref(bad), C.byref(h)) == V.ERR_INVALID
    if torch.cuda.device_count() == 0:
        # no CPU fallback: without a device the product refuses to construct
        with pytest.raises(V.VslamError) as ei:
            V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376)
        assert ei.value.code == V.ERR_NO_DEVICE


@pytest.mark.parametrize("weighting,norm", [(0, 1), (0, 0), (1, 2), (2, 1), (3, 0)])
def test_bow_assemble_equals_dbow3_restatement(weighting, norm):
    """vslam_bow_assemble (host half of Frame::ComputeBoW, GPU-free) vs the oracle's restatement of
    DBoW3::Vocabulary::transform: BowVector / FeatureVector identical, doubles bit for bit."""
    import vi_slam_amd as V
    from oracle import orbo
    from vi_slam_amd import synth
    voc = synth.make_vocabulary(6, 3, seed=3, stop_fraction=0.1, weighting=weighting, norm=norm)
    rng = np.random.default_rng(5)
    leaves = np.nonzero(voc["child_count"] == 0)[0]
    d = voc["desc"][rng.choice(leaves, 400)].copy()
    for i in range(len(d)):
        for f in rng.integers(0, 256, 8):
            d[i, f >> 3] ^= np.uint8(1 << (f & 7))
    want = orbo.bow_transform(voc, d, 2)
    got = V.bow_assemble(weighting, norm, want["word"], want["weight"], want["nid"])
    for k in ("bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
        assert np.array_equal(got[k], want[k]), k
    assert len(got["bow_ids"]) > 20 and (want["weight"] == 0).any()


def test_worker_pool_stress_under_thread_sanitizer(tmp_path):
    """The context's worker pool (vslam_pool.h): alternating small/large parallel_for calls with stack lambdas;
    every index runs exactly once and ThreadSanitizer reports nothing (stale-index race)."""
    import subprocess
    exe = str(tmp_path / "pool_stress")
    src = os.path.join(ROOT, "tests", "cpp", "pool_stress.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", exe, src])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    env.pop("LD_PRELOAD", None)  # run_sanitizers.sh preloads ASan for the Python process; TSan cannot share a process with it
    r = subprocess.run([exe, "6000"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pool_stress ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr


@pytest.mark.parametrize("size,nlevels,scale", [((1241, 376), 8, 1.2), ((1920, 1080), 8, 1.2), ((752, 480), 8, 1.2),
                                                ((321, 203), 6, 1.2), ((640, 480), 5, 1.1), ((1241, 376), 12, 1.1),
                                                ((128, 128), 2, 1.2), ((500, 400), 4, 1.5)])
def test_fused_pyramid_plan_emulation_equals_oracle_cascade(H, size, nlevels, scale):
    """The tile plan of the fused pyramid kernel (k_pyramid_group: groups of levels computed in LDS with recomputed
    halos) evaluated on the CPU with the kernel's own indexing: every level equals the oracle's cv::resize cascade,
    no tile reads a byte that was neither staged nor computed, and the plan fits LDS."""
    w, h = size
    img = synth.make_frame(w, h, seed=w * 3 + nlevels)
    e = orbo.Extractor(500, scale=scale, nlevels=nlevels)
    e.pyramid_only(img)
    sizes = [e.level(l).shape for l in range(nlevels)]
    out = np.zeros(sum(a * b for a, b in sizes[1:]) + 16, np.uint8)
    lds, ntiles = C.c_int(), C.c_int()
    H.vslamh_pyramid_fused.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    ng = H.vslamh_pyramid_fused(_p(img), w, h, img.strides[0], 500, scale, nlevels, _p(out), C.byref(lds), C.byref(ntiles))
    assert ng == (0 if nlevels == 1 else 1 + max(0, (nlevels - 1 - 3 + 3) // 4)), ng
    assert 0 < lds.value <= 64 * 1024
    o = 0
    for l in range(1, nlevels):
        hh, ww = sizes[l]
        assert np.array_equal(out[o:o + hh * ww].reshape(hh, ww), e.level(l)), l
        o += hh * ww
