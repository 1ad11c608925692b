#!/bin/bash
set -o pipefail
O=gpurun_out/pyr
mkdir -p $O
for rows in 48 40 32; do
  for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real; do
    VSLAM_PYR_ROWS=$rows timeout -k 10 240 python bench.py --workload $wl --inputs device --no-cpu-baseline > $O/b_${wl}_$rows.json 2> $O/b_${wl}_$rows.err
    echo "rows=$rows $wl rc=$? $(python3 -c "import json; d=json.load(open('$O/b_${wl}_$rows.json')); print(d['value'], d.get('spread'))" 2>/dev/null)"
  done
done | tee $O/pyr_bench.txt
echo done
