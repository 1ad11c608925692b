#!/bin/bash
# round 4, VERDICT items 1 + 2 in the pipeline: A/B of vslam_tuning.fast_kernel (3 = one workgroup per cell, 4 = per band)
# and vslam_tuning.wave_prio (s_setprio 3 at entry of the chain kernels) on the four workloads, two repetitions each,
# plus one kernel trace per setting of the mono and stereo benches (rows of describe / quadtree under load).
# usage (through gpurun): bash tools/r04_prio_ab.sh [configs...]   config = k<fast_kernel>p<wave_prio>
set -o pipefail
O=gpurun_out/prio
R=$PWD
mkdir -p $O
CFGS=${@:-k3p0 k4p0 k4p1 k4p3 k4p7 k3p7}
WLS="kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real"
for rep in 1 2; do
for cfg in $CFGS; do
  k=${cfg:1:1}; prio=${cfg:3}
  for wl in $WLS; do
    timeout -k 10 240 python bench.py --workload $wl --inputs device --no-cpu-baseline --fast-kernel $k --wave-prio $prio > $O/b_${wl}_${cfg}_r${rep}.json 2> $O/b_${wl}_${cfg}_r${rep}.err
    echo "cfg=$cfg rep=$rep $wl rc=$? $(python3 -c "import json,sys; d=json.load(open('$O/b_${wl}_${cfg}_r${rep}.json')); print(d['value'], d.get('spread'))" 2>/dev/null)"
  done
done
done
cd /tmp && export TMPDIR=/tmp
for cfg in k4p0 k4p7; do
  k=${cfg:1:1}; prio=${cfg:3}
  for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${wl}_${cfg} -o t -- python3 $R/bench.py --workload $wl --inputs device --no-cpu-baseline --steps 20 --fast-kernel $k --wave-prio $prio > $R/$O/prof_${wl}_${cfg}.json 2> $R/$O/prof_${wl}_${cfg}.err
    echo "-- trace $wl $cfg rc=$?"
    find $R/$O/prof_${wl}_${cfg} -name "*_trace.csv" -delete
  done
done
cd $R
echo done
