#!/bin/bash
# round 3, GPU call 1: probes + stamps + real-image tests
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out
echo "== issue rate probe"; timeout -k 10 300 tools/bin/issue_rate_probe 1000 > $O/issue_rate_probe.txt 2>&1; echo rc=$?
echo "== real image tests"; timeout -k 10 600 python -m pytest tests/test_real_images.py -m gpu -x -q -s > $O/real_tests.log 2>&1; echo rc=$?; tail -5 $O/real_tests.log
echo "== octree stamps"
for cfg in "1241 376 1000 4" "1241 376 2000 4" "1920 1080 4000 4" "1241 376 1000 1"; do
  VSLAM_FE_LIB=$PWD/vi_slam_amd/libvslam_fe_stamps.so VSLAM_OCT_DBG=1 timeout -k 10 120 python tools/octree_stamps.py $cfg >> $O/octree_stamps.txt 2>&1
done
cat $O/octree_stamps.txt
echo "== d2h route probe"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE=0" "HSA_FORCE_SDMA_SIZE=1" "DEBUG_CLR_LIMIT_BLIT_WG=4"; do
  tag=$(echo $v | tr '=' '_')
  if [ "$v" = default ]; then
    timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/$O/d2h_$tag -o t -- $R/tools/bin/d2h_route_probe > $R/$O/d2h_$tag.txt 2>&1
  else
    env_k=${v%%=*}; env_v=${v#*=}
    export $env_k=$env_v
    timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/$O/d2h_$tag -o t -- $R/tools/bin/d2h_route_probe > $R/$O/d2h_$tag.txt 2>&1
    unset $env_k
  fi
  echo "-- $v rc=$?"; grep "D2H" $R/$O/d2h_$tag.txt
  find $R/$O/d2h_$tag -name "*kernel_stats.csv" -exec cat {} \; | head -5
  find $R/$O/d2h_$tag -name "*memory_copy_stats.csv" -exec cat {} \; | head -5
done
timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/$O/d2h_with_h2d -o t -- $R/tools/bin/d2h_route_probe with_h2d > $R/$O/d2h_with_h2d.txt 2>&1
echo "-- with_h2d rc=$?"; grep "D2H" $R/$O/d2h_with_h2d.txt
find $R/$O/d2h_with_h2d -name "*kernel_stats.csv" -exec cat {} \; | head -5
find $R/$O/d2h_with_h2d -name "*memory_copy_stats.csv" -exec cat {} \; | head -5
# keep only the small csv summaries
find $R/$O -name "*_trace.csv" -size +2M -delete
echo done
