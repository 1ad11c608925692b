#!/bin/bash
# round 4: SQ / traffic counters of the stereo step (KITTI, 2000 features, 16 pairs per launch): what the four k_stereo_*
# kernels cost next to the extractor's.   usage (through gpurun): bash tools/r04_stereo_pmc.sh
set -o pipefail
O=gpurun_out/stereo_pmc
R=$PWD
mkdir -p $O
timeout -k 10 900 bash tools/collect_pmc.sh $R/$O stereo 32 2000 1241 376 > $O/log.txt 2>&1; echo "pmc rc=$?"
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))["kernels"]
for n,e in d.items():
    print(n[:36].ljust(36), {c: (round(e[c]/1e6,3) if isinstance(e.get(c),(int,float)) else e.get(c)) for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVES","SQ_BUSY_CYCLES","SQ_WAIT_INST_LDS","SQ_LDS_BANK_CONFLICT","fetch_bytes_raw","write_bytes")})
PY
find $O -name "*.csv" -size +2M -delete
echo done
