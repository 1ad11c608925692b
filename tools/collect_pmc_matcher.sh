#!/bin/bash
# Matcher counters north_star names (L2 hit rate, LDS bank-conflict share) + kernel durations, for the matcher kernels as
# they are NOW: tools/run_match_loop.py (8 stereo pairs + 7 init pairs + one brute-force top-2, KITTI size, 2000 features).
# Separate rocprofv3 passes: TCC counters, SQ/LDS counters, kernel trace (--pmc is never combined with tracing).
#   tools/collect_pmc_matcher.sh <outdir>
set -e
OUT=${1:-/root/repo/gpurun_out/pmc_matcher}
mkdir -p $OUT/tcc $OUT/sq $OUT/trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 /root/repo/tools/run_match_loop.py > $OUT/tcc/log.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/sq -- python3 /root/repo/tools/run_match_loop.py > $OUT/sq/log.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 /root/repo/tools/run_match_loop.py > $OUT/trace/log.txt 2>&1
python3 /root/repo/tools/summarize_pmc_matcher.py $OUT
