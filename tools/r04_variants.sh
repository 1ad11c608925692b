#!/bin/bash
# round 4: the optional paths behind vslam_tuning switches stay bit-exact -- the extraction / real-image / stereo tests under
# each switch a reviewer may flip.   usage (through gpurun): bash tools/r04_variants.sh
set -o pipefail
O=gpurun_out/variants
mkdir -p $O
T="tests/test_gpu_extract.py tests/test_real_images.py tests/test_gpu_round2.py"
i=0
for env in "VSLAM_FAST_KERNEL=3" "VSLAM_FAST_KERNEL=4 VSLAM_FAST_BAND_CELLS=2" "VSLAM_FAST_KERNEL=4 VSLAM_FAST_BAND_CELLS=1" "VSLAM_OCT_PRECOUNT=1" \
           "VSLAM_WAVE_PRIO=15" "VSLAM_DESC_KPW=1" "VSLAM_DESC_KPW=4" "VSLAM_OCT_THREADS=1024" "VSLAM_OCT_THREADS=256" "VSLAM_FAST_LDS_PAD=12288" \
           "VSLAM_PYR_ROWS=16" "VSLAM_STREAM_PRIORITY=2"; do
  i=$((i+1))
  env $env timeout -k 10 600 python -m pytest $T -m gpu -x -q > $O/v$i.log 2>&1; rc=$?
  echo "$env rc=$rc $(tail -1 $O/v$i.log)"
  [ $rc -ne 0 ] && tail -30 $O/v$i.log
done | tee $O/summary.txt
echo done
