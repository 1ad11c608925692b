#!/bin/bash
O=gpurun_out
M="--workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs device"
for a in "" "--no-exchange-chain" "--exchange allgather"; do
  echo "== force collective $a"; timeout -k 10 300 python bench.py $M --force-collective $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('exchange'), d['spread'])"
done
echo "== plain"; timeout -k 10 300 python bench.py $M 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['spread'])"
