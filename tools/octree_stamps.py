import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import vi_slam_amd as V
from vi_slam_amd import synth
W,H,NF,B=1241,376,2000,16
fe=V.FExtractor(NF,1.2,8,20,7,W,H,max_batch=B)
dev=torch.zeros((B,H,1280),dtype=torch.uint8,device="cuda")
for s in range(B): dev[s,:,:W]=torch.from_numpy(synth.make_frame(W,H,step=s)).cuda()
ptrs=[dev[s].data_ptr() for s in range(B)]
torch.cuda.synchronize()
for _ in range(3):
    fe.compute_batch_async(ptrs,1280,(0,0),to_host=False); fe.wait()
out=(C.c_ulonglong*64)()
L=V.lib(); L.vslam_dbg_octree_stamps.argtypes=[C.c_void_p,C.c_void_p]
print(L.vslam_dbg_octree_stamps(fe._h,out))
n=out[63]; t=[out[i] for i in range(n)]
print("nstamps",n); print([round((t[i]-t[0])/100.0,1) for i in range(n)])
