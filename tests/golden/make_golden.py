#!/usr/bin/env python3
"""Generate tests/golden/*.npz (run in the build container, where /root/reference exists).

Inputs: small gray crops of the reference's own PNG test images (thirdparty/vilib/visual_lib/test/
images; gray = OpenCV's fixed-point BGR2GRAY formula).  Expected outputs:
  * FAST corners+scores from the REFERENCE's Rosten FAST compiled into oracle/_ref (real reference
    output: pins the FAST stage),
  * full-pipeline outputs of the CPU oracle (keypoints, descriptors, stereo, init matches) -- these pin
    the oracle against regressions; they are NOT reference outputs (the reference needs OpenCV 4.2).
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orbo  # noqa: E402

IMG = "/root/reference/thirdparty/vilib/visual_lib/test/images"
OUT = os.path.dirname(os.path.abspath(__file__))


def gray(path):
    rgb = np.array(Image.open(path).convert("RGB")).astype(np.int64)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def main():
    lenna = gray(os.path.join(IMG, "lenna.png"))
    hutL = gray(os.path.join(IMG, "scenery/hut_stereo/01.png"))
    hutR = gray(os.path.join(IMG, "scenery/hut_stereo/02.png"))
    crops = {
        "lenna_256x192": lenna[160:352, 128:384],
        "hut_320x200": hutL[140:340, 200:520],
    }
    # ---- FAST golden from the reference's Rosten code
    fast = {}
    for name, im in crops.items():
        fast[name + "_img"] = im
        for th in (7, 20):
            r = orbo.ref_fast9(im, th)
            assert r is not None, "build oracle/_ref first (make -C oracle)"
            fast["%s_th%d" % (name, th)] = r
    np.savez_compressed(os.path.join(OUT, "fast_rosten.npz"), **fast)

    # ---- oracle pipeline golden on a stereo crop pair (320x240 keeps 8 levels >= 40 px)
    L = hutL[100:340, 160:480].copy()
    R = hutR[100:340, 160:480].copy()
    eL, eR = orbo.Extractor(500), orbo.Extractor(500)
    kL, dL, mL = eL.compute(L)
    kR, dR, mR = eR.compute(R)
    u, dep, bi, bs = orbo.stereo(eL, eR, kL, dL, kR, dR, 40.0, 400.0)
    eM = orbo.Extractor(500)
    kM, dM, mM = eM.compute(L, lap=(0, 1000))
    nm, m12, pm = orbo.search_for_initialization(kL, dL, kR, dR, 320, 240, window=100)
    # ---- tracking matchers on the same real-image pair (frame 1 = left image, frame 2 = the third image of the
    # hut_stereo set cropped the same way): stereo points of frame 1 as MapPoints, a small sideways motion
    hut3 = gray(os.path.join(IMG, "scenery/hut_stereo/03.png"))
    C = hut3[100:340, 160:480].copy()
    eC = orbo.Extractor(500)
    kC, dC, _ = eC.compute(C)
    T0 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
    cam = (400.0, 400.0, 160.0, 120.0, 40.0, 0.1)
    z = np.where(dep > 0, dep, 5.0).astype(np.float32)
    x3, has = orbo.unproject_stereo(kL, z, T0, cam[2], cam[3], 1.0 / cam[0], 1.0 / cam[1])
    Tcw = np.hstack([np.eye(3), np.array([[0.02], [0.0], [0.0]])]).astype(np.float32)
    flags = (has * 3).astype(np.uint8)
    flags[::7] = 1  # some temporal points without observations
    sbp_n, sbp_m, sbp_dir = orbo.search_by_projection_frame(Tcw, T0, cam, 15, kL, flags, x3, dL, kC, dC,
                                                            np.full(len(kC), -1, np.float32), eL.tables()["scale"],
                                                            320, 240)
    mps = np.zeros(len(kL), orbo.MP_TRACK_DTYPE)
    mps["proj_x"], mps["proj_y"] = kL["x"] + 2.0, kL["y"] + 0.5
    mps["proj_xr"] = mps["proj_x"] - 40.0 / z
    mps["view_cos"] = np.where(np.arange(len(kL)) % 3 == 0, 0.9, 0.999).astype(np.float32)
    mps["level"] = kL["octave"]
    mps["flags"] = 3
    occ = (sbp_m >= 0).astype(np.uint8)
    mp_n, mp_m = orbo.search_by_projection_mappoints(mps, dL, kC, dC, np.full(len(kC), -1, np.float32),
                                                     eL.tables()["scale"], 320, 240, 3.0, 0.8, occ)
    sizes = [1, 2, 3, 5, 9, 20]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dd = dL[:off[-1]]
    np.savez_compressed(os.path.join(OUT, "tracking_hut_320x240.npz"), C=C, kC=kC, dC=dC, x3=x3, flags=flags, Tcw=Tcw,
                        cam=np.asarray(cam, np.float32), sbp_nmatches=np.int32(sbp_n), sbp_match=sbp_m, mps=mps, occ=occ,
                        mp_nmatches=np.int32(mp_n), mp_match=mp_m, dist_off=off, dist_desc=dd,
                        dist_best=orbo.distinctive_descriptors(dd, off))
    np.savez_compressed(os.path.join(OUT, "pipeline_hut_320x240.npz"), L=L, R=R, kL=kL, dL=dL, kR=kR, dR=dR,
                        uRight=u, depth=dep, kM=kM, dM=dM, monoIndex=np.int32(mM), init_matches=m12,
                        init_nmatches=np.int32(nm), lvl3=eL.level(3), lvl3_blur=eL.level(3, blurred=True))
    print("wrote golden fixtures:", sorted(os.listdir(OUT)))


def make_fastgrid():
    """tests/golden/fastgrid.npz: the grid detector behind geometry::FAST::detect on the crops of fast_rosten.npz
    (cut from the reference's own test images): the oracle's feature grids for several configurations, and the
    output of the REFERENCE's rosten::fast10_detect_nonmax<false> / <true> (oracle/_ref) per pyramid level."""
    z = np.load(os.path.join(OUT, "fast_rosten.npz"))
    out = {}
    cfgs = [(0, 1, 0, 0, 10, 1, 0, 100), (0, 3, 0, 0, 10, 1, 0, 100), (0, 3, 0, 0, 10, 1, 1, 100), (1, 3, 8, 5, 9, 2, 0, 200),
            (0, 2, 0, 0, 12, 0, 0, 105), (0, 2, 16, 16, 11, 1, 0, 75)]
    for name, key in (("lenna", "lenna_256x192_img"), ("hut", "hut_320x200_img")):
        img = z[key]
        h, w = img.shape
        img = np.ascontiguousarray(img[:h & ~3, :w & ~3])
        for c in cfgs:
            pos, sc, lv = orbo.fg_detect(img, (32, 32), c[0], c[1], (c[2], c[3]), c[7] / 10.0, c[4], c[5], c[6])
            k = "%s__%s" % (name, "_".join(str(v) for v in c))
            out[k + "_pos"], out[k + "_score"], out[k + "_level"] = pos, sc, lv
        cur = img
        for level in range(3):
            if level:
                cur = orbo.fg_halfsample(cur)
            for new in (0, 1):
                r = orbo.ref_fast_detect_nonmax(cur, 10, 10, bool(new))
                if r is not None:
                    out["ref_%s_l%d_fast10_%s" % (name, level, "new" if new else "old")] = r
    np.savez_compressed(os.path.join(OUT, "fastgrid.npz"), **out)
    print("wrote fastgrid.npz:", len(out), "arrays")


REAL_STEREO = (822.5 * 0.4, 822.5)  # (bf, fx): hut_stereo.json's fx; the baseline is a test parameter, not a calibration
REAL_CASES = [  # (image, nfeatures, lapping area): Frame's stereo ctor passes {0,0}, the mono ctor {0,1000} (frame.cpp:107,289)
    ("hut1", 1200, (0, 0)), ("hut2", 1200, (0, 0)), ("hut3", 2000, (0, 0)), ("hut4", 2000, (0, 0)), ("hut5", 2000, (0, 0)),
    ("lenna", 1200, (0, 0)), ("lenna", 2000, (0, 1000)), ("chess", 1200, (0, 0)), ("chess", 2000, (0, 1000)),
    ("hut1", 5000, (0, 1000)), ("hut2", 5000, (0, 1000)), ("hut4", 5000, (0, 1000)), ("hut5", 5000, (0, 1000)),
]
REAL_STEREO_PAIRS = [("hut1", "hut2", 1200), ("hut3", "hut4", 2000), ("hut4", "hut5", 2000)]
REAL_INIT_PAIRS = [("hut1", "hut2", 5000), ("hut4", "hut5", 5000)]  # mono initialisation extracts 5 x nFeatures (tracking.cpp:1093)


def real_case_key(name, nf, lap):
    return "%s_n%d_lap%d_%d" % (name, nf, lap[0], lap[1])


def real_images():
    """The reference's own test images at FULL size as gray arrays (fixed-point BGR2GRAY): hut_stereo/01-05.png 752x480,
    lenna.png 512x512, chessboard_798_798.png (the tie-heavy one)."""
    ims = {"hut%d" % i: gray(os.path.join(IMG, "scenery/hut_stereo/%02d.png" % i)) for i in range(1, 6)}
    ims["lenna"] = gray(os.path.join(IMG, "lenna.png"))
    ims["chess"] = gray("/root/reference/test/images/chessboard_798_798.png")
    return ims


def make_real_fullsize():
    """tests/golden/real_images.npz (inputs) + real_expected.npz (oracle outputs: keypoints in full, descriptors as
    SHA-256, stereo uRight/depth and init matches in full).  These pin the oracle on real pixels at full size; they are
    oracle outputs, not reference outputs (the reference needs OpenCV 4.2)."""
    import hashlib
    ims = real_images()
    np.savez_compressed(os.path.join(OUT, "real_images.npz"), **ims)
    out = {}
    for name, nf, lap in REAL_CASES:
        e = orbo.Extractor(nf)
        k, d, m = e.compute(ims[name], lap=lap)
        key = real_case_key(name, nf, lap)
        out[key + "_kps"] = k
        out[key + "_desc_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(d).tobytes()).digest(), np.uint8)
        out[key + "_mono"] = np.int32(m)
        out[key + "_ncand"] = np.asarray([len(e.candidates(l)) for l in range(8)], np.int32)
    for a, b, nf in REAL_STEREO_PAIRS:
        eL, eR = orbo.Extractor(nf), orbo.Extractor(nf)
        kL, dL, _ = eL.compute(ims[a])
        kR, dR, _ = eR.compute(ims[b])
        u, dep, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, *REAL_STEREO)
        out["stereo_%s_%s_n%d_uRight" % (a, b, nf)] = u
        out["stereo_%s_%s_n%d_depth" % (a, b, nf)] = dep
    for a, b, nf in REAL_INIT_PAIRS:
        e = orbo.Extractor(nf)
        k1, d1, _ = e.compute(ims[a], lap=(0, 1000))
        k2, d2, _ = e.compute(ims[b], lap=(0, 1000))
        h, w = ims[a].shape
        nm, m12, pm = orbo.search_for_initialization(k1, d1, k2, d2, w, h, window=100)
        out["init_%s_%s_n%d_matches" % (a, b, nf)] = m12
        out["init_%s_%s_n%d_nmatches" % (a, b, nf)] = np.int32(nm)
    np.savez_compressed(os.path.join(OUT, "real_expected.npz"), **out)
    print("wrote real_images.npz (%d images), real_expected.npz (%d arrays)" % (len(ims), len(out)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fastgrid":
        make_fastgrid()
    elif len(sys.argv) > 1 and sys.argv[1] == "real":
        make_real_fullsize()
    else:
        main()
        make_fastgrid()
        make_real_fullsize()
