set -o pipefail
O=gpurun_out/d1; mkdir -p $O; R=$PWD
timeout -k 10 300 python -m pytest tests/test_gpu_match.py -x -q -k "deferred or init" > $O/t1.log 2>&1; rc=$?; tail -5 $O/t1.log; [ $rc -ne 0 ] && exit 1
for d in single separate single separate; do
timeout -k 10 300 python bench.py --workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --delivery $d > $O/mono_$d.json 2> $O/mono_$d.err; python - <<P
import json; j=json.load(open("$O/mono_$d.json")); print("$d", j["value"], j.get("value_host_inputs"))
P
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -4 $O/gpu_tests.log; [ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
wl=kitti00_mono_1241x376_n1000
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/prof_$wl -o t -- python3 $R/bench.py --workload $wl --inputs device --no-cpu-baseline --steps 20 > $R/$O/prof_$wl.json 2> $R/$O/prof_$wl.err
echo "-- $wl rc=$?"
find $R/$O/prof_$wl -name "*_trace.csv" -delete
