#!/bin/bash
# round 4: k_fast_bands against k_fast_cells_v3 -- parity tests with the band kernel (the default), single-context stage
# times of both (VSLAM_FAST_KERNEL=3|4), SQ / traffic counters of both at KITTI batch 32.
# usage (through gpurun): bash tools/r04_bands_check.sh [tests|times|pmc ...]
set -o pipefail
O=gpurun_out/bands
R=$PWD
mkdir -p $O
WHAT=${@:-tests times pmc}
for w in $WHAT; do
case $w in
tests)
  echo "== gpu tests, band kernel"; timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests_bands.log 2>&1; rc=$?; echo rc=$rc; tail -15 $O/gpu_tests_bands.log
  echo "== gpu tests, two-cell bands"; VSLAM_FAST_BAND_CELLS=2 VSLAM_FAST_KERNEL=4 timeout -k 10 900 python -m pytest tests/test_gpu_extract.py tests/test_real_images.py -m gpu -q > $O/gpu_tests_bands2.log 2>&1; echo rc=$?; tail -4 $O/gpu_tests_bands2.log
  if [ $rc -ne 0 ]; then
    echo "== gpu tests, cell kernel"; VSLAM_FAST_KERNEL=3 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_cells.log 2>&1; echo rc=$?; tail -3 $O/gpu_tests_cells.log
  fi;;
times)
  for k in "3 4" "4 4"; do
    set -- $k
    for cfg in "1241 376 1000 32" "1920 1080 4000 32" "752 480 1200 32" "1241 376 1000 2"; do
      echo "kernel=$1 cells=$2 $cfg: $(VSLAM_FAST_KERNEL=$1 VSLAM_FAST_BAND_CELLS=$2 timeout -k 10 200 python tools/stage_times.py $cfg 40 2>/dev/null | tail -1)"
    done
  done | tee $O/stage_times.txt;;
pmc)
  for k in 4; do
    VSLAM_FAST_KERNEL=4 VSLAM_FAST_BAND_CELLS=$k timeout -k 10 600 bash tools/collect_pmc.sh $R/$O/pmc_k$k mono 32 1000 1241 376 > $O/pmc_k$k.log 2>&1; echo "pmc band cells=$k rc=$?"
    python3 - <<EOF
import json
d=json.load(open("$O/pmc_k$k/summary.json"))["kernels"]
for n,e in d.items():
    if n.startswith("k_fast"):
        print(n, {c: e.get(c) for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVES","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","fetch_bytes_raw","write_bytes","SQ_BUSY_CYCLES")})
EOF
  done;;
esac
done
find $O -name "*.csv" -size +2M -delete
echo done
