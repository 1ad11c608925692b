#!/bin/bash
# round 4: the oracle-checked fuzzers (tests/tools/) over the kernels rewritten this round: random extractor configurations
# with the band FAST kernel and four keypoints per wave forced (single-image contexts would take the per-cell kernel and one
# keypoint per wave), the library defaults, the grid detector, the init matcher.   usage (through gpurun): bash tests/tools/r04_fuzz.sh
set -o pipefail
O=gpurun_out/fuzz
mkdir -p $O
FUZZ_SEED=11 FUZZ_N=150 VSLAM_FAST_KERNEL=4 VSLAM_DESC_KPW=4 timeout -k 10 900 python tests/tools/fuzz_extract.py > $O/extract_bands_kpw4.txt 2>&1; echo "extract bands+kpw4 rc=$? $(tail -1 $O/extract_bands_kpw4.txt)"
FUZZ_SEED=12 FUZZ_N=100 VSLAM_FAST_KERNEL=4 VSLAM_FAST_BAND_CELLS=2 VSLAM_DESC_KPW=2 timeout -k 10 900 python tests/tools/fuzz_extract.py > $O/extract_bands2_kpw2.txt 2>&1; echo "extract bands(2 cells)+kpw2 rc=$? $(tail -1 $O/extract_bands2_kpw2.txt)"
FUZZ_SEED=13 FUZZ_N=100 timeout -k 10 900 python tests/tools/fuzz_extract.py > $O/extract_default.txt 2>&1; echo "extract defaults rc=$? $(tail -1 $O/extract_default.txt)"
FUZZ_SEED=14 FUZZ_N=200 timeout -k 10 900 python tests/tools/fuzz_fastgrid.py > $O/fastgrid.txt 2>&1; echo "fastgrid rc=$? $(tail -1 $O/fastgrid.txt)"
FUZZ_SEED=15 timeout -k 10 900 python tests/tools/fuzz_init_matcher.py > $O/init_matcher.txt 2>&1; echo "init matcher rc=$? $(tail -1 $O/init_matcher.txt)"
echo done
