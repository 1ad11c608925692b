#!/bin/bash
# round 4: (1) GPU tests of the batched brute-force matcher, (2) dynamic block counts of k_fast_bands from a diagnostic build
set -o pipefail
O=gpurun_out/counts
mkdir -p $O
echo "== matcher tests"; timeout -k 10 600 python -m pytest tests/test_gpu_match.py -q -k "hamming" > $O/match_tests.log 2>&1; echo rc=$?; tail -5 $O/match_tests.log
echo "== diagnostic build"
touch vi_slam_amd/csrc/vslam_image_kernels.hip
make -C vi_slam_amd/csrc EXTRA_HIPFLAGS=-DVSLAM_FAST_COUNT > $O/build.log 2>&1; echo rc=$?
for cfg in "1241 376 1000 32" "1920 1080 4000 32" "752 480 1200 32"; do
  echo "$cfg: $(timeout -k 10 200 python tools/fast_band_counts.py $cfg 2>/dev/null | tail -1)"
done | tee $O/band_counts.txt
echo done
