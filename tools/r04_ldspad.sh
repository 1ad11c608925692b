#!/bin/bash
# round 4: does the pipeline gain when k_fast_bands leaves LDS to the other contexts' kernels?  (8 bands of 20 KB take a CU's
# whole LDS; VSLAM_FAST_LDS_PAD makes them 7, 6 or 5 per CU)
set -o pipefail
O=gpurun_out/ldspad
mkdir -p $O
for pad in 0 1024 3584 7168 12288; do
  for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
    VSLAM_FAST_LDS_PAD=$pad timeout -k 10 240 python bench.py --workload $wl --inputs device --no-cpu-baseline > $O/b_${wl}_$pad.json 2> $O/b_${wl}_$pad.err
    echo "pad=$pad $wl rc=$? $(python3 -c "import json; d=json.load(open('$O/b_${wl}_$pad.json')); print(d['value'], d.get('spread'))" 2>/dev/null)"
  done
done | tee $O/ldspad.txt
echo done
