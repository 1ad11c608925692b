#!/bin/bash
# the brute-force matcher at a size where it can be ALU-bound -- 16 x 2000 x 2000 in one launch
# (vslam_hamming_top2_batch): kernel trace, TCC hit rate, SQ / LDS counters, words/s against the integer-ALU ceiling.
#   tools/collect_pmc_matcher_batch.sh <outdir> [P N]
set -e
OUT=${1:-/root/repo/gpurun_out/pmc_matcher_batch}
P=${2:-16}
N=${3:-2000}
mkdir -p $OUT/tcc $OUT/sq $OUT/trace
python3 /root/repo/tools/run_match_batch_loop.py $P $N 50 | tee $OUT/host_clock.txt
python3 /root/repo/tools/run_match_batch_loop.py 1 $N 50 | tee -a $OUT/host_clock.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 /root/repo/tools/run_match_batch_loop.py $P $N 5 > $OUT/tcc/log.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/sq -- python3 /root/repo/tools/run_match_batch_loop.py $P $N 5 > $OUT/sq/log.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 /root/repo/tools/run_match_batch_loop.py $P $N 20 > $OUT/trace/log.txt 2>&1
python3 - $OUT $P $N <<'PY'
import collections, csv, glob, json, sys
root, P, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = collections.defaultdict(dict)
def short(n):
    k = n.split("(")[0]
    return (k[5:] if k.startswith("void ") else k).split("<")[0]
for sub in ("tcc", "sq"):
    files = glob.glob("%s/%s/*/*counter_collection.csv" % (root, sub)) + glob.glob("%s/%s/*counter_collection.csv" % (root, sub))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        if k.startswith("k_hamming_top2"):
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        v = v[1:] if len(v) > 1 else v  # the first launch is the allocating synchronous pass
        out[k][c] = sum(v) / len(v)
for f in glob.glob("%s/trace/*kernel_stats.csv" % root):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith("k_hamming_top2"):
            out[k]["avg_us"] = float(r["AverageNs"]) / 1e3
            out[k]["launches_trace"] = int(r["Calls"])
k = out["k_hamming_top2_batch"]
if "TCC_HIT_sum" in k:
    k["l2_hit_rate"] = k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k.get("TCC_MISS_sum", 0.0), 1.0)
if k.get("SQ_LDS_IDX_ACTIVE"):
    k["lds_bank_conflict_share_of_lds_cycles"] = k.get("SQ_LDS_BANK_CONFLICT", 0.0) / k["SQ_LDS_IDX_ACTIVE"]
words = P * N * N * 8.0
if k.get("avg_us"):
    k["xor_popcount_words_per_s"] = words / (k["avg_us"] * 1e-6)
    peak = 1024 * 64 / 6.0 * 2.26e9  # SIMDs x lanes / (v_xor 2 + v_bcnt 4 cycles) x shader clock (profiles/r03_issue_rate_probe.txt)
    k["integer_alu_ceiling_words_per_s"] = peak
    k["frac_of_ceiling"] = k["xor_popcount_words_per_s"] / peak
    if k.get("SQ_INSTS_VALU"):
        k["valu_per_64_pairs"] = k["SQ_INSTS_VALU"] / (P * N * N / 64.0)
res = {"workload": "tools/run_match_batch_loop.py: %d independent %d x %d brute-force top-2 problems in ONE launch of k_hamming_top2_batch (+ merge)" % (P, N, N),
       "kernels": {n: out[n] for n in sorted(out)}}
json.dump(res, open("%s/summary.json" % root, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
