#!/bin/bash
# round 4: rows per blur task (6 halo rows are re-read and re-filtered per task).   usage (through gpurun): bash tools/r04_blur_rows.sh
set -o pipefail
O=gpurun_out/blurrows
mkdir -p $O
for r in 24 32 48 64 96; do
  for cfg in "1241 376 1000 32" "1920 1080 4000 32" "752 480 1200 32"; do
    echo "blur_rows=$r $cfg: $(VSLAM_BLUR_ROWS=$r timeout -k 10 200 python tools/stage_times.py $cfg 30 2>/dev/null | tail -1)"
  done
done | tee $O/stage_times.txt
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
  for r in 32 48 64 32 64; do
    VSLAM_BLUR_ROWS=$r timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device > $O/${wl}_r$r.json 2> $O/${wl}_r$r.err
    echo "$wl blur_rows=$r rc=$? $(python3 -c "
import json
d=json.load(open('$O/${wl}_r$r.json'))
print(d['value'], d['ms_per_step'], d.get('spread'))")"
  done
done | tee $O/bench.txt
echo done
