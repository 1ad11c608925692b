"""CPU: settle the oracle's OpenCV knobs against REAL OpenCV 4.2 output, IF a maintainer has produced it
(tools/dump_opencv_primitives.cpp -> tests/golden/opencv/out_*.bin; OpenCV is not in this image).
Without those files the OpenCV comparisons are skipped and parity stays "HIP == oracle, oracle unpinned for OpenCV
arithmetic".  The consumer itself is always exercised: a dump in the tool's format is synthesised from the ORACLE in a
temporary directory and pushed through the same checks, so the day a real dump arrives the only thing that can fail
is the arithmetic."""
import glob
import os
import struct

import numpy as np
import pytest

from oracle import orbo

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "golden", "opencv")
HAVE = bool(glob.glob(os.path.join(DIR, "out_*.bin")))
need = pytest.mark.skipif(not HAVE, reason="no OpenCV dump under tests/golden/opencv (see tools/dump_opencv_primitives.cpp)")
TAPS_42 = (18, 34, 48, 56, 48, 34, 18)   # OpenCV 4.2 (error-diffused, sum 256): the product default
TAPS_OLD = (18, 34, 49, 55, 49, 34, 18)
INPUTS = [("hut", 320, 240), ("lenna", 256, 192)]


def read_blob(d, name):
    raw = open(os.path.join(d, name), "rb").read()
    assert raw[:4] == b"VSLD"
    kind, nd = struct.unpack_from("<II", raw, 4)
    dims = struct.unpack_from("<%dI" % nd, raw, 12)
    return kind, dims, raw[12 + 4 * nd:]


def write_blob(d, name, kind, arr):
    arr = np.ascontiguousarray(arr)
    with open(os.path.join(d, name), "wb") as f:
        f.write(b"VSLD" + struct.pack("<II", kind, arr.ndim) + struct.pack("<%dI" % arr.ndim, *arr.shape) + arr.tobytes())


def _img(d, name):
    _, (h, w), p = read_blob(d, name)
    return np.frombuffer(p, np.uint8).reshape(h, w)


def _atan2(yx, fma):
    return np.array([orbo.fast_atan2(y, x, fma) for y, x in yx], np.float32)


# ---- the checks (d = directory holding out_*.bin)
def check_resize(d, name):
    e = orbo.Extractor(500)
    e.pyramid_only(_img(d, "out_resize_%s_l0.bin" % name))
    for l in range(1, 8):
        assert np.array_equal(e.level(l), _img(d, "out_resize_%s_l%d.bin" % (name, l))), (name, l)


def check_blur(d, name):
    """exactly ONE of the two candidate tap sets must reproduce the dumped 7x7 sigma-2 fixed-point blur on every level"""
    ok = {taps: all(np.array_equal(orbo.blur7(_img(d, "out_resize_%s_l%d.bin" % (name, l)), taps),
                                   _img(d, "out_blur_%s_l%d.bin" % (name, l))) for l in range(8))
          for taps in (TAPS_42, TAPS_OLD)}
    assert sum(ok.values()) == 1, ok
    assert ok[TAPS_42], "the product default (vslam_fe_params.gauss_taps all zero) must be the matching set"


def check_fast(d, name, th):
    _, (n, _), p = read_blob(d, "out_fast_%s_th%d.bin" % (name, th))
    want = np.frombuffer(p, np.int32).reshape(n, 3)
    k = orbo.fast_detect(_img(d, "out_resize_%s_l0.bin" % name), th)
    got = np.stack([k["x"], k["y"], k["response"]], 1).astype(np.int32)
    assert np.array_equal(got, want)


def check_atan2(d):
    _, (n, _), p = read_blob(d, "out_atan2_in.bin")
    yx = np.frombuffer(p, np.float32).reshape(n, 2)[:20000]
    want = np.frombuffer(read_blob(d, "out_atan2.bin")[2], np.float32)[:20000]
    eq = {fma: bool(np.array_equal(_atan2(yx, fma), want)) for fma in (0, 1)}
    assert eq[0] or eq[1], "neither the separate mul/add nor the FMA Horner form reproduces cv::fastAtan2"
    return eq


def check_cvround(d):
    x = np.frombuffer(read_blob(d, "out_cvround_in.bin")[2], np.float32)
    want = np.frombuffer(read_blob(d, "out_cvround.bin")[2], np.int32)
    assert np.array_equal(np.rint(x).astype(np.int32), want)


# ---- always: tool present, consumer logic exercised on an oracle-made dump
def test_dump_tool_and_exporter_are_present_and_fixture_inputs_exist():
    root = os.path.dirname(HERE)
    assert os.path.exists(os.path.join(root, "tools", "dump_opencv_primitives.cpp"))
    assert os.path.exists(os.path.join(HERE, "golden", "export_opencv_inputs.py"))
    for name, w, h in INPUTS:
        assert os.path.getsize(os.path.join(DIR, "in_%s_%dx%d.gray" % (name, w, h))) == w * h


def test_consumer_on_a_dump_synthesised_from_the_oracle(tmp_path):
    d = str(tmp_path)
    for name, w, h in INPUTS:
        img = np.fromfile(os.path.join(DIR, "in_%s_%dx%d.gray" % (name, w, h)), np.uint8).reshape(h, w)
        e = orbo.Extractor(500)
        e.pyramid_only(img)
        for l in range(8):
            lv = e.level(l)
            write_blob(d, "out_resize_%s_l%d.bin" % (name, l), 1, lv)
            write_blob(d, "out_blur_%s_l%d.bin" % (name, l), 2, orbo.blur7(lv, TAPS_42))
        for th in (20, 7):
            k = orbo.fast_detect(img, th)
            write_blob(d, "out_fast_%s_th%d.bin" % (name, th), 3, np.stack([k["x"], k["y"], k["response"]], 1).astype(np.int32))
    rng = np.random.default_rng(5)
    yx = rng.integers(-200000, 200001, (3000, 2)).astype(np.float32)
    write_blob(d, "out_atan2_in.bin", 4, yx)
    write_blob(d, "out_atan2.bin", 5, _atan2(yx, 0))
    x = (np.arange(-2000, 2001) * 0.25).astype(np.float32)
    write_blob(d, "out_cvround_in.bin", 6, x)
    write_blob(d, "out_cvround.bin", 7, np.rint(x).astype(np.int32))
    for name, _, _ in INPUTS:
        check_resize(d, name)
        check_blur(d, name)
        for th in (20, 7):
            check_fast(d, name, th)
    assert check_atan2(d)[0]
    check_cvround(d)
    # and the checks do discriminate: a blur made with the OTHER tap set must be rejected
    for l in range(8):
        write_blob(d, "out_blur_hut_l%d.bin" % l, 2, orbo.blur7(_img(d, "out_resize_hut_l%d.bin" % l), TAPS_OLD))
    with pytest.raises(AssertionError):
        check_blur(d, "hut")


# ---- with a real OpenCV dump
@need
@pytest.mark.parametrize("name", ["hut", "lenna"])
def test_resize_cascade_equals_opencv(name):
    check_resize(DIR, name)


@need
@pytest.mark.parametrize("name", ["hut", "lenna"])
def test_gaussian_taps_knob_settled_by_opencv(name):
    check_blur(DIR, name)


@need
@pytest.mark.parametrize("name", ["hut", "lenna"])
@pytest.mark.parametrize("th", [20, 7])
def test_fast_equals_opencv(name, th):
    check_fast(DIR, name, th)


@need
def test_fast_atan2_knob_settled_by_opencv():
    eq = check_atan2(DIR)
    print("cv::fastAtan2 matches: separate mul/add=%s, FMA=%s (VSLAM_FLAG_ATAN_FMA)" % (eq[0], eq[1]))


@need
def test_cvround_is_round_half_even():
    check_cvround(DIR)
