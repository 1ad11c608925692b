#!/bin/bash
# round 4: after a kernel change -- all GPU tests, SQ counters of the KITTI mono pass, single-context stage times, the four bench
# workloads (HBM-resident inputs).   usage (through gpurun): bash tools/r04_quick_check.sh
set -o pipefail
O=gpurun_out/quick
R=$PWD
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc $(tail -1 $O/gpu_tests.log)"
[ $rc -ne 0 ] && { grep -v "^$" $O/gpu_tests.log | tail -30; exit 1; }
for cfg in "1241 376 1000 32" "1920 1080 4000 32" "752 480 1200 32"; do
  echo "$cfg: $(timeout -k 10 200 python tools/stage_times.py $cfg 40 2>/dev/null | tail -1)"
done | tee $O/stage_times.txt
timeout -k 10 600 bash tools/collect_pmc.sh $R/$O/pmc mono 32 1000 1241 376 > $O/pmc.log 2>&1; echo "pmc rc=$?"
python3 - <<PY
import json
d=json.load(open("$O/pmc/summary.json"))["kernels"]
for n,e in d.items():
    print(n[:30].ljust(30), {c: (round(e[c]/1e6,3) if isinstance(e.get(c),(int,float)) else e.get(c)) for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_BUSY_CYCLES")})
PY
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device > $O/$wl.json 2> $O/$wl.err
  echo "$wl rc=$? $(python3 -c "
import json
d=json.load(open('$O/$wl.json'))
print(d['value'], d['ms_per_step'], d.get('spread'))")"
done | tee $O/bench.txt
find $O -name "*.csv" -size +2M -delete
echo done
