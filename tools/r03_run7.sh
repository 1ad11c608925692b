#!/bin/bash
O=gpurun_out
M="--workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs pinned"
for rep in 1 2; do
for L in 0 1 2 3; do
  echo "== mono upload-chain $L (run $rep)"; timeout -k 10 300 python bench.py $M --upload-chain $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value_host_inputs'], d['spread_host_inputs'])"
done
done
S="--workload kitti00_stereo_1241x376_n2000 --no-cpu-baseline --inputs pinned"
for L in 0 1 2; do
  echo "== stereo upload-chain $L"; timeout -k 10 300 python bench.py $S --upload-chain $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value_host_inputs'], d['spread_host_inputs'])"
done
S="--workload synthetic_stereo_1920x1080_n4000 --no-cpu-baseline --inputs pinned"
for L in 0 1 2; do
  echo "== 1080p upload-chain $L"; timeout -k 10 300 python bench.py $S --upload-chain $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value_host_inputs'], d['spread_host_inputs'])"
done
echo "== force collective"; timeout -k 10 300 python bench.py --workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --force-collective > $O/fc.out 2>$O/fc.err; echo rc=$?; cat $O/fc.out | cut -c1-600
echo done
