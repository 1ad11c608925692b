#!/bin/bash
O=gpurun_out
rm -f $O/octree_stamps_ab.txt
for lib in libvslam_fe_stamps.so libvslam_fe_b16s.so; do
  echo "### $lib" >> $O/octree_stamps_ab.txt
  for cfg in "1241 376 1000 4" "1920 1080 4000 4" "1241 376 1000 1"; do
    VSLAM_FE_LIB=$PWD/vi_slam_amd/$lib VSLAM_OCT_DBG=1 timeout -k 10 120 python tools/octree_stamps.py $cfg 2>&1 | grep -v amdgpu >> $O/octree_stamps_ab.txt
  done
done
cat $O/octree_stamps_ab.txt
echo "== latency batch 1 (default lib)"; timeout -k 10 300 python tools/latency_batch1.py 2>&1 | grep -v amdgpu | tee $O/latency_batch1.txt
echo "== latency batch 1 (b16)"; VSLAM_FE_LIB=$PWD/vi_slam_amd/libvslam_fe_b16.so timeout -k 10 300 python tools/latency_batch1.py 2>&1 | grep -v amdgpu
M="--no-cpu-baseline --inputs device"
for lib in libvslam_fe.so libvslam_fe_b16.so libvslam_fe.so libvslam_fe_b16.so; do
  for wl in kitti00_mono_1241x376_n1000 synthetic_stereo_1920x1080_n4000; do
    echo "== $lib $wl"; VSLAM_FE_LIB=$PWD/vi_slam_amd/$lib timeout -k 10 300 python bench.py --workload $wl $M 2>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['spread'])"; python -c "
import json
d=json.loads(open('$O/ab.err').read().strip().split('\n')[-1])
print({k:round(v,4) for k,v in d['bench_detail'][0]['stage_ms_single_context'].items()})"
  done
done
