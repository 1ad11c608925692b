#!/bin/bash
env | grep -E "HSA_|NCCL|RCCL|GPU_MAX" 
M="--workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs device"
echo "== current, force collective, HSA_ENABLE_IPC_MODE_LEGACY=1"; HSA_ENABLE_IPC_MODE_LEGACY=1 timeout -k 10 300 python bench.py $M --force-collective 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
echo "== current, force collective, env as is"; timeout -k 10 300 python bench.py $M --force-collective 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
