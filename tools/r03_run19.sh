#!/bin/bash
for rep in 1 2 3; do
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000; do
for sp in 0 1; do
  echo "== $wl split $sp (run $rep)"; timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs pinned --upload-split $sp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value_host_inputs'], d['spread_host_inputs'])"
done; done; done
