"""CPU: the restatement of the grid FAST detector behind vi_slam::geometry::FAST::detect (oracle/fastgrid_oracle.cpp:
vilib K2 fast_gpu_cuda_tools.cu:244-420, K3 detector_base_gpu_cuda_tools.cu:700-878, K5 pyramid_gpu.cu:76-96) against
the reference's own CPU detector (rosten::fastN_detect_nonmax, compiled from the reference into oracle/_ref), the
subset property the reference's test asserts (test/src/feature_detection/test_fast.cpp:212-245), hand-made tie cases
and the committed golden grids."""
import os

import numpy as np
import pytest

from oracle import orbo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def crops():
    z = np.load(os.path.join(GOLD, "fast_rosten.npz"))
    return {"lenna": z["lenna_256x192_img"], "hut": z["hut_320x200_img"]}


def _survivors(resp):
    h, w = resp.shape
    pad = np.pad(resp, 1)
    nb = np.max([pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy], axis=0)
    ys, xs = np.nonzero((resp > nb) & (resp > 0))
    return np.stack([xs, ys, resp[ys, xs].astype(np.int32)], 1)


needs_ref = pytest.mark.skipif(orbo.ref_fast_detect_nonmax(np.zeros((16, 16), np.uint8), 10) is None,
                               reason="oracle/_ref (the reference's Rosten FAST) not built")


@needs_ref
@pytest.mark.parametrize("arc", [9, 10, 11, 12])
def test_response_and_nms_equal_the_references_cpu_detector(crops, arc):
    """K2's corner test and both Rosten scores, then 3x3 strict NMS == rosten::fastN_detect_nonmax<false|true>
    (what rosten::FASTCPU dispatches to, fast_cpu.cpp:68-92), points, order and scores, on three pyramid levels."""
    total = 0
    for img in crops.values():
        for kind, new in ((orbo.FG_SCORE["SUM_OF_ABS_DIFF_ON_ARC"], False), (orbo.FG_SCORE["MAX_THRESHOLD"], True)):
            for th in (10, 20, 35):
                lv = img
                for level in range(3):
                    if level:
                        lv = orbo.fg_halfsample(lv)
                    mine = _survivors(orbo.fg_response(lv, 3, 3, th, arc, kind))
                    ref = orbo.ref_fast_detect_nonmax(lv, th, arc, new)
                    assert mine.shape == ref.shape and np.array_equal(mine, ref), (arc, kind, th, level)
                    total += len(ref)
    assert total > 3000


def test_halfsample_is_a_truncating_box_average():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 50), dtype=np.uint8)
    want = ((img[0:36:2, 0:50:2].astype(np.int32) + img[0:36:2, 1:50:2] + img[1:36:2, 0:50:2] + img[1:36:2, 1:50:2]) >> 2)
    assert np.array_equal(orbo.fg_halfsample(img), want.astype(np.uint8))


@needs_ref
@pytest.mark.parametrize("levels,border", [((0, 1), (0, 0)), ((0, 3), (0, 0)), ((1, 3), (8, 5)), ((0, 2), (20, 40))])
def test_grid_is_the_per_cell_maximum_of_the_cpu_points(crops, levels, border):
    """test_fast.cpp:212-245 (every GPU grid point is a CPU point) and more: a cell's score is the largest score of
    the reference CPU detector's points that fall into it, over the searched levels."""
    for img in crops.values():
        h, w = img.shape
        w, h = w & ~3, h & ~3
        img = np.ascontiguousarray(img[:h, :w])
        for tie in (0, 1):
            pos, sc, lv = orbo.fg_detect(img, (32, 32), levels[0], levels[1], border, 10.0, 10, 1, tie)
            nc = (w + 31) // 32
            hb, vb = max(3, border[0]), max(3, border[1])
            best = {}
            cpu = set()
            cur = img
            for level in range(levels[1]):
                if level:
                    cur = orbo.fg_halfsample(cur)
                if level < levels[0]:
                    continue
                lh, lw = cur.shape
                for x, y, s in orbo.ref_fast_detect_nonmax(cur, 10, 10, False):
                    if x < hb or y < vb or x >= lw - hb or y >= lh - vb:  # fast_cpu.cpp:112-119
                        continue
                    cpu.add((x << level, y << level, level))
                    cell = (y // (32 >> level)) * nc + x // (32 >> level)
                    best[cell] = max(best.get(cell, 0), s)
            for c in range(len(sc)):
                assert sc[c] == best.get(c, 0), (c, sc[c], best.get(c))
                if sc[c] > 0:
                    assert (int(pos[c, 0]), int(pos[c, 1]), int(lv[c])) in cpu
                    # DetectorBase::addFeaturePoint's cell index of the reported position is this cell
                    assert (int(pos[c, 1]) // 32) * nc + int(pos[c, 0]) // 32 == c
                else:
                    assert lv[c] == -1 and pos[c, 0] == 0 and pos[c, 1] == 0
            assert (sc > 0).sum() > 10


def _dots(w, h, pts, bg=50, fg=200):
    img = np.full((h, w), bg, np.uint8)
    for x, y in pts:
        img[y, x] = fg
    return img


def test_equal_maxima_follow_the_cuda_launch_geometry():
    """An isolated bright pixel is one corner with score 16 * (150 - 10) = 2240.  Two of them in one 32x32 cell tie;
    K3 (detector_base_gpu_cuda_tools.cu:700-878, block 32x4 on level 0) keeps: within a thread (column, row % 4) the
    topmost; within a warp the lane whose index is smallest bit-reversed; the lower warp; raster order never enters."""
    W, H = 96, 96
    def run(pts, tie):
        pos, sc, lv = orbo.fg_detect(_dots(W, H, pts), (32, 32), 0, 1, (0, 0), 10.0, 10, 1, tie)
        c = 1 * 3 + 1  # the middle cell: x0 = y0 = 32, away from the borders
        assert sc[c] == 2240.0 and lv[c] == 0
        return int(pos[c, 0]) - 32, int(pos[c, 1]) - 32
    # same row phase (warp 1: rows 5, 9, ...), lanes 8 and 16: bit-reversed 2 and 1 -> lane 16
    assert run([(32 + 8, 32 + 5), (32 + 16, 32 + 5)], 0) == (16, 5)
    assert run([(32 + 8, 32 + 5), (32 + 16, 32 + 5)], 1) == (8, 5)
    # lanes 1 and 30 (bit-reversed 16 and 15) -> lane 30
    assert run([(32 + 1, 32 + 9), (32 + 30, 32 + 9)], 0) == (30, 9)
    # same thread (column 4, rows 6 and 10 are both row phase 2): the topmost
    assert run([(32 + 4, 32 + 10), (32 + 4, 32 + 6)], 0) == (4, 6)
    # different warps: row 5 is warp 1, row 8 is warp 0 -> warp 0 although raster order would take row 5
    assert run([(32 + 4, 32 + 5), (32 + 20, 32 + 8)], 0) == (20, 8)
    assert run([(32 + 4, 32 + 5), (32 + 20, 32 + 8)], 1) == (4, 5)
    # same lane in different warps and a later row of the winning warp
    assert run([(32 + 7, 32 + 3), (32 + 7, 32 + 12)], 0) == (7, 12)  # rows 3 -> warp 3, 12 -> warp 0


def test_levels_merge_with_strict_greater():
    """Levels are merged in ascending order with `d_score < max_resp` (:871-876): the finer level keeps a tie, a
    coarser level needs a strictly larger score.  A 2x2 bright block is suppressed on level 0 (four equal
    neighbours) and is a single bright pixel on level 1."""
    c = 1 * 4 + 1  # cell x 32..63, y 32..63
    img = _dots(128, 128, [(40, 40)])
    pos, sc, lv = orbo.fg_detect(img, (32, 32), 0, 2, (0, 0), 10.0, 10, 1, 0)
    assert sc[c] == 2240.0 and lv[c] == 0 and tuple(pos[c]) == (40.0, 40.0)
    block = [(50, 50), (51, 50), (50, 51), (51, 51)]
    img = _dots(128, 128, block)
    assert _survivors(orbo.fg_response(img, 3, 3, 10.0, 10, 1)).shape[0] == 0
    pos, sc, lv = orbo.fg_detect(img, (32, 32), 0, 2, (0, 0), 10.0, 10, 1, 0)
    assert sc[c] == 2240.0 and lv[c] == 1 and tuple(pos[c]) == (50.0, 50.0)
    img = _dots(128, 128, [(40, 40)] + block)  # 2240 on both levels: level 0 stays
    pos, sc, lv = orbo.fg_detect(img, (32, 32), 0, 2, (0, 0), 10.0, 10, 1, 0)
    assert sc[c] == 2240.0 and lv[c] == 0 and tuple(pos[c]) == (40.0, 40.0)
    for x, y in block:
        img[y, x] = 250  # 16 * (200 - 10) = 3040 on level 1
    pos, sc, lv = orbo.fg_detect(img, (32, 32), 0, 2, (0, 0), 10.0, 10, 1, 0)
    assert sc[c] == 3040.0 and lv[c] == 1 and tuple(pos[c]) == (50.0, 50.0)


def test_golden_grids(crops):
    """tests/golden/fastgrid.npz (made by tests/golden/make_golden.py from the reference's own test images)."""
    z = np.load(os.path.join(GOLD, "fastgrid.npz"))
    n = 0
    for key in z.files:
        if not key.endswith("_score"):
            continue
        name, cfg = key[:-6].split("__")
        lv0, lv1, hb, vb, arc, kind, tie, th10 = [int(v) for v in cfg.split("_")]
        img = crops[name]
        h, w = img.shape
        img = np.ascontiguousarray(img[:h & ~3, :w & ~3])
        pos, sc, lv = orbo.fg_detect(img, (32, 32), lv0, lv1, (hb, vb), th10 / 10.0, arc, kind, tie)
        assert np.array_equal(sc, z[key]) and np.array_equal(pos, z[key[:-6] + "_pos"]) and np.array_equal(lv, z[key[:-6] + "_level"])
        n += 1
    assert n >= 8
