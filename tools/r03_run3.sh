#!/bin/bash
set -o pipefail
O=gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in default; do
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/mono_trace_$v -o t -- python3 $R/bench.py --workload kitti00_mono_1241x376_n1000 --inputs device --no-cpu-baseline --steps 20 > $R/$O/mono_trace_$v.json 2> $R/$O/mono_trace_$v.err
  echo "-- $v rc=$?"
  find $R/$O/mono_trace_$v -name "*kernel_stats.csv" -exec head -12 {} \; | cut -c1-120
  find $R/$O/mono_trace_$v -name "*memory_copy_stats.csv" -exec head -5 {} \;
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/$O/mono_trace_$v/*memory_copy_trace.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    print(rows[0].keys())
    c=collections.Counter()
    for r in rows:
        dur=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        c[r['Direction']]+=1
    print(c)
PY
  find $R/$O/mono_trace_$v -name "*_trace.csv" -size +20M -delete
done
echo done
