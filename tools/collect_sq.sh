#!/bin/bash
# SQ counters only (two passes) for the extraction kernels:  tools/collect_sq.sh <outdir> [mono|stereo B NF W H]
set -e
OUT=${1:-/root/repo/gpurun_out/pmc_sq}
WL=${2:-mono}
B=${3:-32}
NF=${4:-1000}
W=${5:-1241}
H=${6:-376}
mkdir -p $OUT/sq $OUT/sq2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d $OUT/sq -- python3 /root/repo/tools/run_extract_loop.py $WL 5 $B $NF $W $H > $OUT/sq/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/sq2 -- python3 /root/repo/tools/run_extract_loop.py $WL 5 $B $NF $W $H > $OUT/sq2/log.txt 2>&1
python3 /root/repo/tools/summarize_pmc.py $OUT $B $NF $W $H
