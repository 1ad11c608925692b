#!/bin/bash
# round 4: orientation and descriptors as two launches (vslam_tuning.desc_split) against the fused kernel -- parity tests with
# the split (the default for batches), single-context stage times, and the four bench workloads per (split, keypoints per wave).
# usage (through gpurun): bash tools/r04_desc_split.sh [tests|times|bench ...]
set -o pipefail
O=gpurun_out/dsplit
mkdir -p $O
WHAT=${@:-tests times bench}
for w in $WHAT; do
case $w in
tests)
  echo "== gpu tests (split = library default)"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -4 $O/gpu_tests.log
  for k in 1 2 4; do
    echo "== extraction tests, split, kpw=$k"; VSLAM_DESC_SPLIT=1 VSLAM_DESC_KPW=$k timeout -k 10 600 python -m pytest tests/test_gpu_extract.py tests/test_real_images.py -m gpu -x -q > $O/gpu_tests_k$k.log 2>&1; echo rc=$?; tail -2 $O/gpu_tests_k$k.log
  done
  [ $rc -ne 0 ] && exit 1;;
times)
  for cfg in "1241 376 1000 32" "1241 376 2000 32" "1920 1080 4000 32" "752 480 1200 32"; do
    for sk in "0 -1" "0 1" "1 1" "1 2" "1 4"; do
      set -- $sk
      echo "split=$1 kpw=$2 $cfg: $(VSLAM_DESC_SPLIT=$1 VSLAM_DESC_KPW=$2 timeout -k 10 200 python tools/stage_times.py $cfg 30 2>/dev/null | tail -1)"
    done
  done | tee $O/stage_times.txt;;
bench)
  for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
    for sk in "0 -1" "1 -1" "1 1" "1 2" "1 4" "0 -1" "1 -1"; do
      set -- $sk
      VSLAM_DESC_SPLIT=$1 VSLAM_DESC_KPW=$2 timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device > $O/${wl}_s$1_k$2.json 2> $O/${wl}_s$1_k$2.err
      echo "$wl split=$1 kpw=$2 rc=$? $(python3 -c "
import json
d=json.load(open('$O/${wl}_s$1_k$2.json'))
print(d['value'], d['ms_per_step'], d.get('spread'))")"
    done
  done | tee $O/bench.txt;;
esac
done
echo done
