#!/usr/bin/env python3
"""Summarise tools/collect_pmc_matcher.sh into one JSON: per matcher kernel and per launch the L2 (TCC) hit rate, the LDS
bank-conflict share of LDS-active cycles, instruction counts and the average duration; for the all-pairs top-2 kernel the
XOR+popcount word rate against the integer-ALU ceiling measured by tools/issue_rate_probe.hip (v_xor_b32 2 cycles +
v_bcnt_u32_b32 4 cycles per wave64 instruction and SIMD)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
MATCHER = ("k_stereo_rows", "k_stereo_best", "k_stereo_refine", "k_stereo_median_cut", "k_si_topm", "k_si_replay",
           "k_hamming_top2_batch", "k_hamming_top2_merge_batch")


def short(name):
    k = name.split("(")[0]
    if k.startswith("void "):
        k = k[5:]
    return k.split("<")[0]


out = collections.defaultdict(dict)
for sub in ("tcc", "sq"):
    files = glob.glob("%s/%s/*/*counter_collection.csv" % (root, sub)) + glob.glob("%s/%s/*counter_collection.csv" % (root, sub))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        if k in MATCHER:
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
        out[k]["launches_" + sub] = len(v)
for f in glob.glob("%s/trace/*kernel_stats.csv" % root):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k in MATCHER:
            out[k]["avg_us"] = float(r["AverageNs"]) / 1e3
            out[k]["launches_trace"] = int(r["Calls"])
for k, d in out.items():
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0.0), 1.0)
    if d.get("SQ_ACTIVE_INST_LDS"):
        # rocprof's own "LDSBankConflict" metric: conflict cycles over SQ_ACTIVE_INST_LDS -- which counts QUAD-cycles on this
        # chip (MI355X_MICROARCH.md), so the ratio can exceed 1; kept for comparison with rounds 1-2
        d["lds_bank_conflict_over_active_inst_lds"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_ACTIVE_INST_LDS"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        # both in LDS-array cycles: the share of the LDS pipe's busy cycles that were conflict replays
        d["lds_bank_conflict_share_of_lds_cycles"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
NQ = NT = 2000  # run_match_loop.py: one brute-force top-2 of a KITTI frame pair at 2000 features (actual counts ~1990)
k = out.get("k_hamming_top2_batch")
if k and k.get("avg_us"):
    words = NQ * NT * 8.0
    k["xor_popcount_words_per_s"] = words / (k["avg_us"] * 1e-6)
    peak = 1024 * 64 / 6.0 * 2.26e9  # SIMDs x lanes / (2 + 4 cycles) x shader clock
    k["integer_alu_ceiling_words_per_s"] = peak
    k["frac_of_ceiling"] = k["xor_popcount_words_per_s"] / peak
    k["note"] = "%d x %d descriptors x 8 words; a launch this small (%.0f us) is latency- and launch-bound, not ALU-bound" % (NQ, NT, k["avg_us"])
res = {"workload": "tools/run_match_loop.py: 1241x376, 2000 features, per iteration 8 stereo pairs (k_stereo_*), 7 frame pairs of "
                   "SearchForInitialization (k_si_*), one brute-force top-2 (k_hamming_top2_*); 5 iterations; per-launch averages",
       "kernels": {k: out[k] for k in sorted(out)}}
json.dump(res, open("%s/summary.json" % root, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True)[:4000])
