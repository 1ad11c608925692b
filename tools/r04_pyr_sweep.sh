#!/bin/bash
# round 4: the fused pyramid's single-context time against tile height and threads per tile (the stereo and 1080p steps are the sum
# of the wide kernels' single-context times)
set -o pipefail
O=gpurun_out/pyr
mkdir -p $O
for cfg in "1241 376 2000 32" "1920 1080 4000 32" "752 480 1200 32"; do
  for nt in 256 512; do
    for rows in 24 32 40 48 56 64; do
      echo "rows=$rows nt=$nt $cfg: $(VSLAM_PYR_ROWS=$rows VSLAM_PYR_NT=$nt timeout -k 10 200 python tools/stage_times.py $cfg 30 2>/dev/null | tail -1 | cut -c1-60)"
    done
  done
done | tee $O/pyr_sweep.txt
echo done
