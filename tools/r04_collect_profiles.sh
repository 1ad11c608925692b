#!/bin/bash
# copy what tools/r04_final.sh left under gpurun_out/final into profiles/ (the committed, judged copies)
F=gpurun_out/final
cp $F/default_bench.json profiles/r04_default_bench.json
cp $F/prof_kitti00_mono_1241x376_n1000/t_kernel_stats.csv profiles/r04_mono_bench_kernel_stats.csv
cp $F/prof_kitti00_stereo_1241x376_n2000/t_kernel_stats.csv profiles/r04_stereo_bench_kernel_stats.csv
cp $F/prof_kitti00_mono_1241x376_n1000/t_memory_copy_stats.csv profiles/r04_mono_bench_memory_copy_stats.csv 2>/dev/null
cp $F/prof_single/t_kernel_stats.csv profiles/r04_single_context_kernel_stats.csv
cp $F/prof_kitti00_mono_1241x376_n1000.json profiles/r04_mono_bench_under_rocprof.json
cp $F/prof_kitti00_stereo_1241x376_n2000.json profiles/r04_stereo_bench_under_rocprof.json
for g in 1241x376_n1000 1241x376_n2000 1920x1080_n4000 752x480_n1200; do cp $F/pmc_$g/summary.json profiles/r04_pmc_traffic_${g}_b32.json; done
cp $F/pmc_matcher/summary.json profiles/r04_pmc_matcher_kitti_n2000.json
cp $F/pmc_matcher_batch/summary.json profiles/r04_pmc_matcher_batch_16x2000.json
cp $F/latency_batch1.txt profiles/r04_latency_batch1.txt
cp $F/rehearse_multi.txt profiles/r04_multi_rank_rehearsal_one_gpu.txt
# tools/r04_prof_extra.sh: the other workloads under rocprofv3
for p in "synthetic_stereo_1920x1080_n4000 1080p" "hut_stereo_752x480_n1200_real hut" "kitti00_mono_1241x376_n2000 mono_n2000"; do set -- $p; [ -f $F/prof_$1/t_kernel_stats.csv ] && cp $F/prof_$1/t_kernel_stats.csv profiles/r04_$2_bench_kernel_stats.csv; done
