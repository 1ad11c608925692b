#!/usr/bin/env python3
"""Write the raw gray crops tools/dump_opencv_primitives.cpp reads (tests/golden/opencv/in_*.gray) from the committed
fixtures.  No reference access needed: the crops are data already in tests/golden/*.npz."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(HERE, "opencv")
os.makedirs(out, exist_ok=True)
hut = np.load(os.path.join(HERE, "pipeline_hut_320x240.npz"))["L"]
lenna = np.load(os.path.join(HERE, "fast_rosten.npz"))["lenna_256x192_img"]
for name, im in (("hut", hut), ("lenna", lenna)):
    h, w = im.shape
    im.tofile(os.path.join(out, "in_%s_%dx%d.gray" % (name, w, h)))
    print("wrote in_%s_%dx%d.gray" % (name, w, h))
