#!/bin/bash
set -o pipefail
O=gpurun_out
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo rc=$rc; tail -4 $O/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
echo "== bench default"; timeout -k 10 900 python bench.py > $O/bench_r03a.json 2> $O/bench_r03a.err; echo rc=$?; cat $O/bench_r03a.json
M="--workload kitti00_mono_1241x376_n1000 --no-cpu-baseline"
for i in 1 2 3; do
  echo "== mono chain run $i"; timeout -k 10 300 python bench.py $M 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_host_inputs'], d['spread_host_inputs'], d.get('roofline_pcie'))"
  echo "== mono no-chain run $i"; timeout -k 10 300 python bench.py $M --no-upload-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_host_inputs'], d['spread_host_inputs'])"
done
echo "== force collective"; timeout -k 10 300 python bench.py $M --force-collective 2>$O/fc.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_host_inputs'], d.get('exchange'), d['config']['sharding'])"; tail -3 $O/fc.err | cut -c1-300
echo "== gloo 2 ranks same gpu"; timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --same-gpu $M --inputs device 2>$O/gloo2.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d.get('exchange'))"; tail -3 $O/gloo2.err | cut -c1-300
echo done
