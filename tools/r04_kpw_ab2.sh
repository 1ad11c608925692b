#!/bin/bash
# round 4: keypoints per wave of the descriptor kernel again, now that its keypoint records are scalar (58 VGPRs at four per
# wave instead of 110).   usage (through gpurun): bash tools/r04_kpw_ab2.sh
set -o pipefail
O=gpurun_out/kpw2
mkdir -p $O
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
  for k in -1 1 2 4 -1 4; do
    VSLAM_DESC_KPW=$k timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device > $O/${wl}_k$k.json 2> $O/${wl}_k$k.err
    echo "$wl kpw=$k rc=$? $(python3 -c "
import json
d=json.load(open('$O/${wl}_k$k.json'))
print(d['value'], d['ms_per_step'], d.get('spread'))")"
  done
done | tee $O/summary.txt
echo done
