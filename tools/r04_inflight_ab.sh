#!/bin/bash
# round 4: contexts in flight at 32 images per step.   usage (through gpurun): bash tools/r04_inflight_ab.sh
set -o pipefail
O=gpurun_out/inflight
mkdir -p $O
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
  for n in 3 4 5 6 8; do
    timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device --inflight $n > $O/${wl}_i$n.json 2> $O/${wl}_i$n.err
    echo "$wl inflight=$n rc=$? $(python3 -c "
import json
d=json.load(open('$O/${wl}_i$n.json'))
print(d['value'], d['ms_per_step'], d.get('spread'))")"
  done
done | tee $O/summary.txt
echo done
