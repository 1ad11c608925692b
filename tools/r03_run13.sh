#!/bin/bash
O=gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in "--force-collective" ""; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/fc_trace -o t -- python3 $R/bench.py --workload kitti00_mono_1241x376_n1000 --no-cpu-baseline --inputs device $mode --steps 20 > $R/$O/fc_trace.json 2> $R/$O/fc_trace.err
echo "mode [$mode] rc=$?"
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/$O/fc_trace/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
c=collections.Counter()
for r in rows:
    nm=r['Kernel_Name'][:24]
    if nm.startswith('void k_fast') or nm.startswith('void rccl') or nm.startswith('k_si_replay') or nm.startswith('__amd_rocclr_copyB'):
        c[(nm, r['Queue_Id'], r['Stream_Id'])]+=1
for k,v in sorted(c.items()): print(k, v)
PY
rm -rf $R/$O/fc_trace
done
