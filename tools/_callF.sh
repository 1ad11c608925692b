set -o pipefail
O=gpurun_out/dF; mkdir -p $O
for v in "" q40 q60 ""; do
if [ -n "$v" ]; then export VSLAM_FE_LIB=$PWD/vi_slam_amd/libvslam_fe_$v.so; else unset VSLAM_FE_LIB; fi
timeout -k 10 200 python -m pytest tests/test_gpu_extract.py -x -q -k "stagewise or other_geometries" 2>&1 | tail -1
for wl in kitti00_mono_1241x376_n1000 kitti00_stereo_1241x376_n2000; do
timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --inputs device > $O/b.json 2> $O/b.err || exit 1
python - <<P
import json; j=json.load(open("$O/b.json")); print("variant=$v", "$wl"[:20], j["value"])
P
done; done
