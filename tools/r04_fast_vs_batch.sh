#!/bin/bash
# round 4: single-context stage times against the number of images per launch -- does k_fast_bands' time step with the
# number of workgroup rounds (10336 workgroups of 20 KB on 2048 slots at 32 KITTI images)?
# usage (through gpurun): bash tools/r04_fast_vs_batch.sh
set -o pipefail
O=gpurun_out/fastb
mkdir -p $O
for b in 8 16 24 28 30 31 32 33 34 36 40 48 56 64; do
  echo "B=$b $(timeout -k 10 200 python tools/stage_times.py 1241 376 1000 $b 30 2>/dev/null | tail -1)"
done | tee $O/fast_vs_batch.txt
echo done
