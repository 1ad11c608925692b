#!/bin/bash
# what the blocks well under the median are -- the four workloads with the host clock of every step dumped,
# once with Python's garbage collector left on inside the timed region (round 3's behaviour) and once with it off (default now)
set -o pipefail
O=gpurun_out/slow
mkdir -p $O
WLS="kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000 synthetic_stereo_1920x1080_n4000 hut_stereo_752x480_n1200_real"
for gc in gcon gcoff; do
  flag=""; [ $gc = gcon ] && flag="--gc"
  for wl in $WLS; do
    timeout -k 10 240 python bench.py --workload $wl --inputs device --no-cpu-baseline $flag --stamp-dump $O/st_$gc > $O/b_${wl}_$gc.json 2> $O/b_${wl}_$gc.err
    echo "$gc $wl rc=$? $(python3 -c "import json; d=json.load(open('$O/b_${wl}_$gc.json')); print(d['value'], d.get('spread'))" 2>/dev/null)"
  done
done
python3 tools/slow_blocks.py $O/st_*.json | tee $O/slow_blocks.txt
echo done
