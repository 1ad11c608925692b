#!/bin/bash
# round 4: the bench under rocprofv3 (kernel trace + stats) for the two workloads tools/r04_final.sh does not trace.
# usage (through gpurun): bash tools/r04_prof_extra.sh
set -o pipefail
R=$PWD
O=gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real kitti00_mono_1241x376_n2000; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl -o t -- python3 $R/bench.py --workload $wl --inputs device --no-cpu-baseline --steps 20 > $R/$O/prof_$wl.json 2> $R/$O/prof_$wl.err
  echo "-- $wl rc=$?"
  find $R/$O/prof_$wl -name "*_trace.csv" -delete
done
cd $R
find $O -name "*.csv" -size +2M -delete
echo done
