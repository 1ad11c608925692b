"""GPU (-m gpu): the grid FAST detector behind vi_slam::geometry::FAST::detect (src/geometry/fast_cuda.cpp:70-132 ->
vilib::FASTGPU) through the C ABI of include/vslam_fastgrid.h, bit-exact against oracle/fastgrid_oracle.cpp, the
committed golden grids and -- where oracle/_ref is present -- the reference's own CPU detector."""
import os

import numpy as np
import pytest

import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth
from vi_slam_amd.fastgrid import FASTGPU, MAX_THRESHOLD, SUM_OF_ABS_DIFF_ALL, SUM_OF_ABS_DIFF_ON_ARC

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _crops():
    z = np.load(os.path.join(GOLD, "fast_rosten.npz"))
    out = {}
    for name, key in (("lenna", "lenna_256x192_img"), ("hut", "hut_320x200_img")):
        img = z[key]
        h, w = img.shape
        out[name] = np.ascontiguousarray(img[:h & ~3, :w & ~3])
    return out


def _same(got, want):
    assert np.array_equal(got[1], want[1]), "score"
    assert np.array_equal(got[2], want[2]), "level"
    assert np.array_equal(got[0], want[0]), "pos"


@pytest.mark.parametrize("cfg", [
    dict(),                                                         # fast_cuda.cpp:24-39: one level, 32x32, th 10, arc 10
    dict(max_level=3), dict(max_level=3, tie_rule=1), dict(min_level=1, max_level=3, horizontal_border=8, vertical_border=5),
    dict(max_level=2, min_arc_length=9, score=MAX_THRESHOLD, threshold=20.0),
    dict(max_level=2, min_arc_length=12, score=SUM_OF_ABS_DIFF_ALL, threshold=10.5),
    dict(max_level=2, min_arc_length=11, threshold=7.5, horizontal_border=16, vertical_border=16),
    dict(max_level=3, cell_size_width=64, cell_size_height=64), dict(max_level=2, cell_size_width=64, cell_size_height=32),
    dict(max_level=3, threshold=0.0), dict(max_level=2, horizontal_border=200, vertical_border=3),
])
def test_grid_equals_oracle_on_reference_image_crops(cfg):
    for img in _crops().values():
        h, w = img.shape
        d = FASTGPU(w, h, **cfg)
        try:
            got = d.detect(img)
            want = orbo.fg_detect(img, (cfg.get("cell_size_width", 32), cfg.get("cell_size_height", 32)),
                                  cfg.get("min_level", 0), cfg.get("max_level", 1),
                                  (cfg.get("horizontal_border", 0), cfg.get("vertical_border", 0)), cfg.get("threshold", 10.0),
                                  cfg.get("min_arc_length", 10), cfg.get("score", SUM_OF_ABS_DIFF_ON_ARC), cfg.get("tie_rule", 0))
            _same(got, want)
            assert len(d.getPoints(*got)) == int((want[1] > 0).sum())
        finally:
            d.close()


def test_pyramid_and_response_images():
    """vilib::Frame's half-sampled pyramid (pyramid_gpu.cu:76-96) and DetectorBaseGPU::copyResponseTo."""
    img = _crops()["hut"]
    h, w = img.shape
    for score, th, arc in ((SUM_OF_ABS_DIFF_ON_ARC, 10.0, 10), (MAX_THRESHOLD, 15.0, 9), (SUM_OF_ABS_DIFF_ALL, 12.25, 12)):
        d = FASTGPU(w, h, max_level=3, threshold=th, min_arc_length=arc, score=score, horizontal_border=6)
        try:
            d.detect(img)
            cur = img
            for l in range(3):
                if l:
                    cur = orbo.fg_halfsample(cur)
                assert np.array_equal(d.level(0, l), cur)
                want = orbo.fg_response(cur, 5, 3, th, arc, score)  # detection border = max(3, border - 1)
                got = d.response(0, l)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        finally:
            d.close()


def test_golden_grids():
    crops = _crops()
    z = np.load(os.path.join(GOLD, "fastgrid.npz"))
    n = 0
    for key in z.files:
        if not key.endswith("_score"):
            continue
        name, cfg = key[:-6].split("__")
        lv0, lv1, hb, vb, arc, kind, tie, th10 = [int(v) for v in cfg.split("_")]
        img = crops[name]
        d = FASTGPU(img.shape[1], img.shape[0], 32, 32, lv0, lv1, hb, vb, th10 / 10.0, arc, kind, tie)
        try:
            _same(d.detect(img), (z[key[:-6] + "_pos"], z[key], z[key[:-6] + "_level"]))
        finally:
            d.close()
        n += 1
    assert n >= 12


def test_kitti_size_batch_device_inputs_and_the_references_cpu_points():
    import torch
    W, H, B = 1240, 376, 6  # KITTI width rounded down to a multiple of 4 (pyramid_pool.cpp:58-59 asserts divisibility)
    frames = [np.ascontiguousarray(synth.make_frame(1241, 376, step=s)[:, :W]) for s in range(B)]
    d = FASTGPU(W, H, max_level=3, max_batch=B)
    try:
        pos, sc, lv = d.detect_batch(frames)
        for s in range(B):
            _same((pos[s], sc[s], lv[s]), orbo.fg_detect(frames[s], (32, 32), 0, 3))
        one = d.detect(frames[2])
        _same(one, (pos[2], sc[2], lv[2]))
        dev = torch.zeros((B, H, 1280), dtype=torch.uint8, device="cuda")
        for s in range(B):
            dev[s, :, :W] = torch.from_numpy(frames[s]).cuda()
        torch.cuda.synchronize()
        p2, s2, l2 = d.detect_batch(dev_ptrs=[dev[s].data_ptr() for s in range(B)], pitch=1280)
        assert np.array_equal(p2, pos) and np.array_equal(s2, sc) and np.array_equal(l2, lv)
        assert (sc[0] > 0).mean() > 0.8
        # test_fast.cpp:212-245: every grid point is one of the reference CPU detector's points
        ref0 = orbo.ref_fast_detect_nonmax(frames[0], 10, 10, False)
        if ref0 is not None:
            cpu = {(int(x), int(y)) for x, y, _ in ref0}
            for c in np.nonzero((sc[0] > 0) & (lv[0] == 0))[0]:
                assert (int(pos[0, c, 0]), int(pos[0, c, 1])) in cpu
    finally:
        d.close()


def test_equal_maxima_follow_the_cuda_launch_geometry_on_the_device():
    def dots(pts):
        img = np.full((96, 96), 50, np.uint8)
        for x, y in pts:
            img[y, x] = 200
        return img
    cases = [[(40, 37), (48, 37)], [(33, 41), (62, 41)], [(36, 42), (36, 38)], [(36, 37), (52, 40)], [(39, 35), (39, 44)],
             [(35 + i, 33 + (7 * i) % 29) for i in range(0, 28, 3)]]
    for tie in (0, 1):
        d = FASTGPU(96, 96, tie_rule=tie)
        try:
            for pts in cases:
                img = dots(pts)
                _same(d.detect(img), orbo.fg_detect(img, (32, 32), 0, 1, (0, 0), 10.0, 10, 1, tie))
        finally:
            d.close()
    d = FASTGPU(96, 96)
    try:
        pos, sc, lv = d.detect(dots(cases[0]))
        assert tuple(pos[4]) == (48.0, 37.0) and sc[4] == 2240.0  # lane 16 beats lane 8
    finally:
        d.close()


def test_flat_and_saturated_images_and_bad_parameters():
    d = FASTGPU(128, 64, max_level=2)
    try:
        for v in (0, 128, 255):
            pos, sc, lv = d.detect(np.full((64, 128), v, np.uint8))
            assert np.all(sc == 0) and np.all(lv == -1) and np.all(pos == 0)
        rng = np.random.default_rng(0)
        noise = (rng.integers(0, 2, (64, 128)) * 255).astype(np.uint8)
        _same(d.detect(noise), orbo.fg_detect(noise, (32, 32), 0, 2))
    finally:
        d.close()
    for bad in (dict(cell_size_width=48), dict(min_arc_length=8), dict(max_level=0), dict(max_level=9), dict(score=3),
                dict(max_level=3, image_width=130)):
        kw = dict(image_width=128, image_height=64)
        kw.update(bad)
        with pytest.raises(Exception):
            FASTGPU(**kw)
