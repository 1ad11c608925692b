#!/bin/bash
O=gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/fc_trace -o t -- python3 $R/bench.py --workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs device --force-collective --steps 20 > $R/$O/fc_trace.json 2> $R/$O/fc_trace.err
echo rc=$?
head -16 $R/$O/fc_trace/*kernel_stats.csv | cut -c1-150
python3 - <<PY
import csv,glob
f=glob.glob("$R/$O/fc_trace/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
# last 400 rows: timeline per stream
rows=rows[-3000:]
t0=int(rows[0]['Start_Timestamp'])
import collections
for r in rows[-160:]:
    print(r.get('Stream_Id',r.get('Queue_Id')), r['Kernel_Name'][:40], (int(r['Start_Timestamp'])-t0)/1000, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
PY
find $R/$O/fc_trace -name "*_trace.csv" -delete
