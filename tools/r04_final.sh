#!/bin/bash
# round 4: the measurements the documents quote (run through gpurun): tests, smoke, the default bench line, the same bench
# under rocprofv3 (kernel trace), single-context kernel stats, PMC passes (traffic per geometry, matcher counters, the batched
# brute-force matcher), batch-1 latency, the multi-rank rehearsal with the exchange proof.
# usage: bash tools/r04_final.sh [part ...]   parts: tests bench prof pmc matcher latency multi
set -o pipefail
O=gpurun_out
R=$PWD
mkdir -p $O/final
PARTS=${@:-tests bench prof pmc matcher latency multi}
for part in $PARTS; do
case $part in
tests)
  echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/final/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -4 $O/final/gpu_tests.log
  if [ $rc -ne 0 ]; then exit 1; fi
  echo "== smoke"; timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -2;;
bench)
  echo "== bench default"; timeout -k 10 900 python bench.py > $O/final/default_bench.json 2> $O/final/default_bench.err; echo rc=$?; wc -c $O/final/default_bench.json
  python3 -c "
import json; d=json.load(open('$O/final/default_bench.json'))
print(d['value'], d['value_host_inputs'], d['spread'], d['roofline'].get('kernel'), d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'))
for e in d.get('extra_workloads', []): print(e['workload'], e['value'], e['value_host_inputs'], e.get('spread'))";;
prof)
  cd /tmp && export TMPDIR=/tmp
  for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000; do
    timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/final/prof_$wl -o t -- python3 $R/bench.py --workload $wl --inputs device --no-cpu-baseline --steps 20 > $R/$O/final/prof_$wl.json 2> $R/$O/final/prof_$wl.err
    echo "-- $wl rc=$?"
    find $R/$O/final/prof_$wl -name "*_trace.csv" -delete
  done
  echo "== single-context kernel stats (what roofline.avg_launch_ms is compared with)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/final/prof_single -o t -- python3 $R/tools/run_extract_loop.py mono 100 32 1000 1241 376 > $R/$O/final/prof_single.log 2>&1; echo "rc=$?"
  find $R/$O/final/prof_single -name "*_trace.csv" -delete
  cd $R;;
pmc)
  echo "== pmc traffic"; for cfg in "mono 32 1000 1241 376" "mono 32 2000 1241 376" "stereo 32 4000 1920 1080" "stereo 32 1200 752 480"; do set -- $cfg; timeout -k 10 600 bash tools/collect_pmc.sh $R/$O/final/pmc_${4}x${5}_n${3} $1 $2 $3 $4 $5 > $O/final/pmc_${4}x${5}_n${3}.log 2>&1; echo "$cfg rc=$?"; done;;
matcher)
  echo "== pmc matcher"; timeout -k 10 600 bash tools/collect_pmc_matcher.sh $R/$O/final/pmc_matcher > $O/final/pmc_matcher.log 2>&1; echo rc=$?
  echo "== pmc matcher batch"; timeout -k 10 600 bash tools/collect_pmc_matcher_batch.sh $R/$O/final/pmc_matcher_batch 16 2000 > $O/final/pmc_matcher_batch.log 2>&1; echo rc=$?; tail -30 $O/final/pmc_matcher_batch.log;;
latency)
  echo "== latency"; timeout -k 10 300 python tools/latency_batch1.py 2>&1 | grep -v amdgpu > $O/final/latency_batch1.txt; cat $O/final/latency_batch1.txt;;
multi)
  echo "== multi-rank rehearsal"; bash tools/rehearse_multi.sh > $O/final/rehearse_multi.txt 2>&1; tail -8 $O/final/rehearse_multi.txt;;
esac
done
find $O/final -name "*.csv" -size +2M -delete
echo done
