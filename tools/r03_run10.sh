#!/bin/bash
cd _r02
M="--workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs device"
echo "== r02 code, force collective"; timeout -k 10 300 python bench.py $M --force-collective 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
echo "== r02 code, plain"; timeout -k 10 300 python bench.py $M 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
