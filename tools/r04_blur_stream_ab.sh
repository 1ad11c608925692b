#!/bin/bash
# round 4: the blur of a pass on a second stream of its context (vslam_tuning.blur_stream) -- parity tests with it (the default for
# batches), then the bench workloads with and without.   usage (through gpurun): bash tools/r04_blur_stream_ab.sh
set -o pipefail
O=gpurun_out/blurstream
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc $(tail -1 $O/gpu_tests.log)"
[ $rc -ne 0 ] && { grep -v "^$" $O/gpu_tests.log | tail -30; exit 1; }
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
  for b in 0 1 0 1; do
    VSLAM_BLUR_STREAM=$b timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $O/${wl}_b$b.json 2> $O/${wl}_b$b.err
    echo "$wl blur_stream=$b rc=$? $(python3 -c "
import json
d=json.load(open('$O/${wl}_b$b.json'))
print(d['value'], d['value_host_inputs'], d['ms_per_step'], d.get('spread'))")"
  done
done | tee $O/summary.txt
echo done
