#!/bin/bash
set -o pipefail
O=gpurun_out
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -5 $O/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
rm -f $O/octree_stamps_walk.txt
for wk in 2 0 1; do
  echo "### oct_walk $wk" >> $O/octree_stamps_walk.txt
  for cfg in "1241 376 1000 4" "1241 376 2000 4" "1920 1080 4000 4" "1241 376 1000 1"; do
    VSLAM_OCT_WALK=$wk VSLAM_FE_LIB=$PWD/vi_slam_amd/libvslam_fe_stamps.so VSLAM_OCT_DBG=1 timeout -k 10 120 python tools/octree_stamps.py $cfg 2>&1 | grep -v amdgpu >> $O/octree_stamps_walk.txt
  done
done
cat $O/octree_stamps_walk.txt
M="--no-cpu-baseline --inputs device"
for wk in 2 0 2 0; do
  for wl in kitti00_mono_1241x376_n1000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
    echo "== walk $wk $wl"; VSLAM_OCT_WALK=$wk timeout -k 10 300 python bench.py --workload $wl $M 2>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['spread'])"; python -c "
import json
d=json.loads(open('$O/ab.err').read().strip().split('\n')[-1])
print({k:round(v,4) for k,v in d['bench_detail'][0]['stage_ms_single_context'].items()})"
  done
done
echo "== latency"; timeout -k 10 300 python tools/latency_batch1.py 2>&1 | grep -v amdgpu
