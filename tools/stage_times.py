#!/usr/bin/env python3
"""Single-context stage times (HIP events on the context's stream, nothing else on the GPU) of the extraction pass:
    stage_times.py [W H NF B passes]      -> one JSON line {pyramid, fast, blur, describe, octree} in ms per pass
Environment knobs of the library (VSLAM_*) are read by the library itself, so A/B sweeps are
    VSLAM_PYR_ROWS=14 python tools/stage_times.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import vi_slam_amd as V  # noqa: E402
from vi_slam_amd import synth  # noqa: E402

W, H, NF, B, passes = (int(a) for a in (sys.argv[1:6] + ["1241", "376", "1000", "32", "40"][len(sys.argv) - 1:]))
fe = V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
pitch = (W + 127) & ~127
dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
nd = min(B, 8 if W * H > 1000000 else B)
fr = [synth.make_frame(W, H, step=s) for s in range(nd)]
for s in range(B):
    dev[s, :, :W] = torch.from_numpy(fr[s % nd]).cuda()
ptrs = [dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
for i in range(2 * passes):
    if i == passes:
        fe.set_profiling(True)
    fe.compute_batch_async(ptrs, pitch, (0, 1000), to_host=False)
    fe.wait()
p = fe.get_profile()
nb = max(p["batches"], 1)
print(json.dumps({k[:-3]: round(p[k] / nb, 4) for k in ("pyramid_ms", "fast_ms", "blur_ms", "describe_ms", "octree_ms")}
                 | {"env": {k: v for k, v in os.environ.items() if k.startswith("VSLAM_")}}))
fe.close()
