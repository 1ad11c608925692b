#!/bin/bash
# round 4, VERDICT r3 item 6: the quadtree's counting walk as k_oct_count (vslam_tuning.oct_precount) -- parity tests with it on
# (default) and off, stage times of both, batch-1 latency of both
set -o pipefail
O=gpurun_out/precount
mkdir -p $O
echo "== gpu tests, precount on"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_on.log 2>&1; rc=$?; echo rc=$rc; tail -4 $O/tests_on.log
echo "== gpu tests, precount off"; VSLAM_OCT_PRECOUNT=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_off.log 2>&1; echo rc=$?; tail -2 $O/tests_off.log
if [ $rc -ne 0 ]; then exit 1; fi
for pc in 1 0; do
  for cfg in "1241 376 1000 32" "1241 376 2000 32" "1920 1080 4000 32" "752 480 1200 32" "1241 376 1000 2"; do
    echo "precount=$pc $cfg: $(VSLAM_OCT_PRECOUNT=$pc timeout -k 10 200 python tools/stage_times.py $cfg 40 2>/dev/null | tail -1)"
  done
done | tee $O/stage_times.txt
for pc in 1 0; do echo "== latency precount=$pc"; VSLAM_OCT_PRECOUNT=$pc timeout -k 10 300 python tools/latency_batch1.py 2>&1 | grep -v amdgpu | tee $O/latency_$pc.txt | cut -c1-400; done
echo done
