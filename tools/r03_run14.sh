#!/bin/bash
M="--workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs device"
for pr in 0 1 2; do
for mode in "" "--force-collective"; do
  echo "== priority $pr [$mode]"; VSLAM_STREAM_PRIORITY=$pr timeout -k 10 300 python bench.py $M $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['spread'])"
done; done
