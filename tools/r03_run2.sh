#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -15 $O/gpu_tests.log
if [ $rc -ne 0 ]; then echo "tests failed; stopping"; exit 1; fi
echo "== real image stats"; timeout -k 10 300 python -m pytest tests/test_real_images.py -m gpu -x -q -s 2>&1 | grep quadtree > $O/real_stats.txt; cat $O/real_stats.txt
echo "== octree stamps"
for cfg in "1241 376 1000 4" "1241 376 2000 4" "1920 1080 4000 4" "1241 376 1000 1" "1241 376 2000 2"; do
  VSLAM_FE_LIB=$PWD/vi_slam_amd/libvslam_fe_stamps.so VSLAM_OCT_DBG=1 timeout -k 10 120 python tools/octree_stamps.py $cfg >> $O/octree_stamps_v4.txt 2>&1
done
grep -v amdgpu.ids $O/octree_stamps_v4.txt
echo "== bench"; timeout -k 10 600 python bench.py > $O/bench_v4.json 2> $O/bench_v4.err; echo rc=$?; cat $O/bench_v4.json
echo "== issue rate probe"; timeout -k 10 300 tools/bin/issue_rate_probe 1500 > $O/issue_rate_probe2.txt 2>&1; echo rc=$?
echo "== d2h pipeline probe"
R=$PWD
cd /tmp && export TMPDIR=/tmp
for args in "pipeline 4 default" "pipeline 4 d2h" "pipeline 1 default"; do
  tag=$(echo $args | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/d2hp_$tag -o t -- $R/tools/bin/d2h_route_probe $args > $R/$O/d2hp_$tag.txt 2>&1
  echo "-- $args rc=$?"; grep "pipeline" $R/$O/d2hp_$tag.txt
  find $R/$O/d2hp_$tag -name "*kernel_stats.csv" -exec head -5 {} \;
  find $R/$O/d2hp_$tag -name "*memory_copy_stats.csv" -exec head -5 {} \;
  find $R/$O/d2hp_$tag -name "*_trace.csv" -delete
done
echo done
