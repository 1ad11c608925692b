#!/bin/bash
# round 4: images per step and contexts in flight -- does a larger batch per context amortise the per-launch fixed costs
# (profiles/r04_fast_vs_batch.txt) in the pipeline too?   usage (through gpurun): bash tools/r04_batch_ab.sh
set -o pipefail
O=gpurun_out/batchab
mkdir -p $O
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 hut_stereo_752x480_n1200_real; do
  for cfg in "32 4" "64 4" "64 3" "48 4" "64 2"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device --batch $1 --inflight $2 > $O/${wl}_b$1_i$2.json 2> $O/${wl}_b$1_i$2.err
    echo "$wl batch=$1 inflight=$2 rc=$? $(python3 -c "
import json,sys
d=json.load(open('$O/${wl}_b$1_i$2.json'))
print(d['value'], d['ms_per_step'], d.get('spread'))")"
  done
done | tee $O/summary.txt
echo done
