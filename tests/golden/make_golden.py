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
    np.savez_compressed(os.path.join(OUT, "pipeline_hut_320x240.npz"), L=L, R=R, kL=kL, dL=dL, kR=kR, dR=dR,
                        uRight=u, depth=dep, kM=kM, dM=dM, monoIndex=np.int32(mM), init_matches=m12,
                        init_nmatches=np.int32(nm), lvl3=eL.level(3), lvl3_blur=eL.level(3, blurred=True))
    print("wrote golden fixtures:", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
