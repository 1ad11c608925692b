#!/bin/bash
# round 4: bench.py with and without the FAST chain (vslam_fe_set_fast_gate) on the default line's workloads, both input modes
set -o pipefail
O=gpurun_out/chain
mkdir -p $O
WLS="kitti00_mono_1241x376_n1000 kitti00_mono_1241x376_n2000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real"
for rep in 1 2; do
for ch in chain nochain; do
  flag=""; [ $ch = nochain ] && flag="--no-fast-chain"
  for wl in $WLS; do
    timeout -k 10 240 python bench.py --workload $wl --no-cpu-baseline $flag > $O/b_${wl}_${ch}_r$rep.json 2> $O/b_${wl}_${ch}_r$rep.err
    echo "$ch rep=$rep $wl rc=$? $(python3 -c "import json; d=json.load(open('$O/b_${wl}_${ch}_r$rep.json')); print(d['value'], d['value_host_inputs'], d.get('spread'), d.get('spread_host_inputs'))" 2>/dev/null)"
  done
done
done | tee $O/chain_ab.txt
echo done
