#!/bin/bash
O=gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/fc_trace -o t -- python3 $R/bench.py --workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs device --force-collective --steps 20 > $R/$O/fc_trace.json 2> $R/$O/fc_trace.err
echo rc=$?
python3 - <<PY
import csv,glob
f=glob.glob("$R/$O/fc_trace/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)
mid=rows[n//2: n//2+140]
t0=int(mid[0]['Start_Timestamp'])
for r in mid:
    print("%3s %-34s %9.1f %7.1f" % (r['Stream_Id'], r['Kernel_Name'][:34], (int(r['Start_Timestamp'])-t0)/1000, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000))
PY
rm -rf $R/$O/fc_trace
