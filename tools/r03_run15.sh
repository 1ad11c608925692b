#!/bin/bash
set -o pipefail
O=gpurun_out
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -6 $O/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
echo "== bench default"; timeout -k 10 900 python bench.py > $O/bench_r03b.json 2> $O/bench_r03b.err; echo rc=$?; python -c "
import json; d=json.loads(open('$O/bench_r03b.json').read().strip().split('\n')[-1])
print('mono', d['value'], d['value_host_inputs'], d['spread'], d['spread_host_inputs'], d.get('roofline_pcie',{}).get('frac'))
for e in d['extra_workloads']: print(e['workload'], e['value'], e['value_host_inputs'], e.get('spread'), e.get('spread_host_inputs'), e.get('quadtree'))
"
echo "== matcher pmc"; timeout -k 10 600 bash tools/collect_pmc_matcher.sh $PWD/$O/pmc_matcher > $O/pmc_matcher.log 2>&1; echo rc=$?; tail -5 $O/pmc_matcher.log
find $O/pmc_matcher -name "*.csv" -size +3M -delete
echo done
