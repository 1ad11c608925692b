#!/usr/bin/env python3
"""debug: staged-upload sequence per transport route, every pass compared with the oracle"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import vi_slam_amd as V
from vi_slam_amd import synth
from oracle import orbo
W, H, NF, B = 1241, 376, 1000, 4
imgs = [synth.make_frame(W, H, seed=77, step=s) for s in range(B)]
e = orbo.Extractor(NF)
ref = [e.compute(im, lap=(0, 1000)) for im in imgs]
def same(res, tag):
    bad = []
    for s in range(B):
        k, d, m = res[s]
        ko, do, mo = ref[s]
        ok = len(k) == len(ko) and all(np.array_equal(k[f], ko[f]) for f in k.dtype.names) and np.array_equal(d, do)
        if not ok: bad.append(s)
    print("   %-28s %s" % (tag, "ok" if not bad else "DIFF slots %s" % bad))
for order in (0, 1):
  for tuning in ([None, {"h2d_route": 1}, {"graphs": 0}, {"h2d_route": 1, "graphs": 0}, None] if order == 0 else [{"h2d_route": 1}, None]):
    print("tuning", tuning)
    fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B, tuning=tuning)
    pin = V.PinnedImages(B, H, W, W)
    for s in range(B):
        pin.array[s][:] = imgs[s]
    pitch = (W + 127) & ~127
    dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
    for s in range(B):
        dev[s, :, :W] = torch.from_numpy(imgs[s]).cuda()
    torch.cuda.synchronize()
    for rep in range(2):
        fe.compute_batch_async([dev[s].data_ptr() for s in range(B)], pitch, (0, 1000))
        same(fe.wait(copy=True), "rep%d device" % rep)
        fe.stage_images_async(pin.ptrs, W, V.IMGS_PINNED)
        fe.compute_batch_async(pin.ptrs, W, (0, 1000), where=V.IMGS_STAGED)
        fe.stage_images_async(pin.ptrs, W, V.IMGS_PINNED)
        same(fe.wait(copy=True), "rep%d staged A" % rep)
        fe.compute_batch_async(pin.ptrs, W, (0, 1000), where=V.IMGS_STAGED)
        same(fe.wait(copy=True), "rep%d staged B" % rep)
        fe.compute_batch_async(pin.ptrs, W, (0, 1000), where=V.IMGS_PINNED)
        same(fe.wait(copy=True), "rep%d pinned" % rep)
    pin.close(); fe.close()
