#!/bin/bash
O=gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for args in "busy 4 same" "busy 4 xstream" "busy 1 same"; do
  tag=$(echo $args | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/d2hb_$tag -o t -- $R/tools/bin/d2h_route_probe $args > $R/$O/d2hb_$tag.txt 2>&1
  echo "-- $args rc=$?"; grep "busy" $R/$O/d2hb_$tag.txt
  cat $R/$O/d2hb_$tag/*kernel_stats.csv | cut -c1-110
  cat $R/$O/d2hb_$tag/*memory_copy_stats.csv
  find $R/$O/d2hb_$tag -name "*_trace.csv" -delete
done
