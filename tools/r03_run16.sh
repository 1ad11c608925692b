#!/bin/bash
set -o pipefail
O=gpurun_out
R=$PWD
echo "== gpu tests (matcher)"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -4 $O/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000; do
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/prof_$wl -o t -- python3 $R/bench.py --workload $wl --inputs device --no-cpu-baseline --steps 20 > $R/$O/prof_$wl.json 2> $R/$O/prof_$wl.err
  echo "-- $wl rc=$?"; cut -c1-100 $R/$O/prof_$wl/t_kernel_stats.csv | head -16; cat $R/$O/prof_$wl/t_memory_copy_stats.csv
  find $R/$O/prof_$wl -name "*_trace.csv" -delete
done
cd $R
echo "== pmc traffic mono"; timeout -k 10 600 bash tools/collect_pmc.sh $R/$O/pmc_traffic_n1000 mono 32 1000 > $O/pmc_traffic_n1000.log 2>&1; echo rc=$?
find $O/pmc_traffic_n1000 -name "*.csv" -size +3M -delete
echo done
