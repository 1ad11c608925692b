set -o pipefail
O=gpurun_out/dH; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t.log 2>&1; rc=$?; tail -4 $O/t.log; [ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python - <<P
import json; j=json.load(open("$O/b.json")); print(j["value"], j["value_host_inputs"]); [print(e["workload"], e["value"], e["value_host_inputs"]) for e in j["extra_workloads"]]
P
FUZZ_SEED=21 FUZZ_N=40 timeout -k 10 500 python tests/tools/fuzz_extract.py 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 300 python tools/latency_batch1.py 2>&1 | grep -v amdgpu | tail -2
