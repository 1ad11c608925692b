#!/bin/bash
# round 4: hardware queues (GPU_MAX_HW_QUEUES, the HIP runtime's limit of 4 by default) x contexts in flight.
# usage (through gpurun): bash tools/r04_hwq_ab.sh
set -o pipefail
O=gpurun_out/hwq
mkdir -p $O
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
  for cfg in "4 4" "8 4" "8 6" "8 8" "16 8" "4 4" "8 6"; do
    set -- $cfg
    GPU_MAX_HW_QUEUES=$1 timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device --inflight $2 > $O/${wl}_q$1_i$2.json 2> $O/${wl}_q$1_i$2.err
    echo "$wl hwq=$1 inflight=$2 rc=$? $(python3 -c "
import json
d=json.load(open('$O/${wl}_q$1_i$2.json'))
print(d['value'], d['ms_per_step'], d.get('spread'))")"
  done
done | tee $O/summary.txt
echo done
