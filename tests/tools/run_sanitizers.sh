#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (the oracle and the GPU-free host logic: tables, cells, resize
# tables, quadtree, frame grid, matcher replay).  GPU sanitizers are not available on the pool; this is the CPU
# build only.  Restores the normal libraries afterwards.
set -e
cd "$(dirname "$0")/../.."
cp vi_slam_amd/libvslam_host.so /tmp/libvslam_host.so.bak
cp oracle/liborb_oracle.so /tmp/liborb_oracle.so.bak
trap 'cp /tmp/libvslam_host.so.bak vi_slam_amd/libvslam_host.so; cp /tmp/liborb_oracle.so.bak oracle/liborb_oracle.so' EXIT
SAN="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -shared -o vi_slam_amd/libvslam_host.so vi_slam_amd/csrc/vslam_host.cpp -Iinclude
(cd oracle && g++ $SAN -fopenmp -pthread -shared -o liborb_oracle.so orb_oracle.cpp orb_oracle_c.cpp fastgrid_oracle.cpp -lm)
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests/test_oracle.py tests/test_fastgrid_oracle.py tests/test_host_logic.py -x -q -p no:cacheprovider
