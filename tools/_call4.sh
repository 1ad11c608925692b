set -o pipefail
O=gpurun_out/d4; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 4 5; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$n -o t -- python3 $R/bench.py --workload kitti00_mono_1241x376_n1000 --inputs device --no-cpu-baseline --steps 20 --min-seconds 0.15 --inflight $n > $R/$O/tr_$n.json 2> $R/$O/tr_$n.err || exit 1
python3 $R/tools/_queue_report.py $R/$O/tr_$n > $R/$O/queues_$n.txt
python3 $R/tools/trace_timeline.py $R/$O/tr_$n 0.5 > $R/$O/timeline_$n.txt
find $R/$O/tr_$n -name "*_trace.csv" -size +30M -delete
done
