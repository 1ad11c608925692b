#!/bin/bash
O=gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/d2hp_q8 -o t -- $R/tools/bin/d2h_route_probe pipeline 4 default > $R/$O/d2hp_q8.txt 2>&1
echo "-- probe with GPU_MAX_HW_QUEUES=8"; grep pipeline $R/$O/d2hp_q8.txt; cat $R/$O/d2hp_q8/*kernel_stats.csv | cut -c1-100; cat $R/$O/d2hp_q8/*memory_copy_stats.csv
unset GPU_MAX_HW_QUEUES
export GPU_MAX_HW_QUEUES=4
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/mono_trace_q4 -o t -- python3 $R/bench.py --workload kitti00_mono_1241x376_n1000 --inputs device --no-cpu-baseline --steps 20 > $R/$O/mono_trace_q4.json 2> $R/$O/mono_trace_q4.err
echo "-- bench with GPU_MAX_HW_QUEUES=4 rc=$?"
grep -E "copyBuffer|k_fast" $R/$O/mono_trace_q4/*kernel_stats.csv | cut -c1-140; cat $R/$O/mono_trace_q4/*memory_copy_stats.csv
find $R/$O -name "*_trace.csv" -delete
echo done
