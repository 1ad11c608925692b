"""Target for `rocprofv3 --kernel-trace`: 60 single-image extractions (the replayed HIP graph of the latency path); the last
kernels of the trace are one steady-state iteration."""
import sys
sys.path.insert(0, "/root/repo")
import vi_slam_amd as V
from vi_slam_amd import synth
fe = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
img = synth.make_frame(1241, 376)
for _ in range(60):
    fe.compute(img, (0, 1000))
fe.close()
