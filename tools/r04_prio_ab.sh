#!/bin/bash
# round 4, VERDICT item 2: A/B of vslam_tuning.wave_prio (s_setprio 3 at entry of the chain kernels) on the four workloads,
# plus one kernel trace per setting of the mono and stereo benches (rows of describe / quadtree under load).
# usage (through gpurun): bash tools/r04_prio_ab.sh [quick]
set -o pipefail
O=gpurun_out/prio
R=$PWD
mkdir -p $O
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -3 $O/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
WLS="kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real"
for rep in 1 2; do
for prio in 0 1 3 7; do
  for wl in $WLS; do
    timeout -k 10 240 python bench.py --workload $wl --inputs device --no-cpu-baseline --wave-prio $prio > $O/b_${wl}_p${prio}_r${rep}.json 2> $O/b_${wl}_p${prio}_r${rep}.err
    echo "prio=$prio rep=$rep $wl rc=$? $(python3 -c "import json,sys; d=json.load(open('$O/b_${wl}_p${prio}_r${rep}.json')); print(d['value'], d.get('spread'))" 2>/dev/null)"
  done
done
done
cd /tmp && export TMPDIR=/tmp
for prio in 0 3 7; do
  for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${wl}_p${prio} -o t -- python3 $R/bench.py --workload $wl --inputs device --no-cpu-baseline --steps 20 --wave-prio $prio > $R/$O/prof_${wl}_p${prio}.json 2> $R/$O/prof_${wl}_p${prio}.err
    echo "-- trace $wl prio=$prio rc=$?"
    find $R/$O/prof_${wl}_p${prio} -name "*_trace.csv" -delete
  done
done
cd $R
echo done
