#!/usr/bin/env python3
"""Table of an A/B run of tools/r04_prio_ab.sh: gpurun_out/prio/b_<workload>_<cfg>_r<rep>.json -> value and spread per
configuration (cfg = k<fast_kernel>p<wave_prio>), plus the rows of the chain kernels from the kernel traces.
    summarize_ab.py [dir] > profiles/r04_fast_kernel_wave_prio_ab.txt"""
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prio"
runs = {}
for f in sorted(glob.glob(os.path.join(root, "b_*_r*.json"))):
    m = re.match(r"b_(.+)_(k\dp\d+)_r(\d+)\.json", os.path.basename(f))
    if not m:
        continue
    try:
        d = json.load(open(f))
    except Exception:  # noqa: BLE001
        continue
    runs.setdefault(m.group(1), {}).setdefault(m.group(2), []).append(d)
print("vslam_tuning.fast_kernel (k3 = k_fast_cells_v3, k4 = k_fast_bands) x vslam_tuning.wave_prio (bit 1 quadtree + output order,")
print("2 descriptors, 4 matchers), bench.py --workload W --inputs device --no-cpu-baseline, HBM-resident frames, frames/s;")
print("every cell: the runs' values, then min / median / max over the blocks of 20 steps of the first run\n")
for wl, cfgs in runs.items():
    print(wl)
    base = None
    for cfg in sorted(cfgs):
        vals = [d["value"] for d in cfgs[cfg]]
        sp = cfgs[cfg][0].get("spread") or {}
        mean = sum(vals) / len(vals)
        if cfg == "k3p0":
            base = mean
        rel = " (%+.1f %% vs k3p0)" % (100 * (mean / base - 1)) if base and cfg != "k3p0" else ""
        print("  %-6s %s   blocks %s / %s / %s%s" % (cfg, "  ".join("%9.0f" % v for v in vals), sp.get("min"), sp.get("median"), sp.get("max"), rel))
    print()
print("kernel traces of the same command (rocprofv3 --kernel-trace --stats; under load, four contexts), average launch in us:")
for d in sorted(glob.glob(os.path.join(root, "prof_*"))):
    if not os.path.isdir(d):
        continue
    st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    rows = {r["Name"].split("(")[0].replace("void ", "").split("<")[0]: r for r in csv.DictReader(open(st[0]))}
    want = ["k_fast_bands", "k_fast_cells_v3", "k_octree_v4", "k_orient_describe_dev", "k_assign_out", "k_si_topm", "k_si_replay",
            "k_stereo_rows", "k_stereo_best", "k_stereo_refine", "k_stereo_median_cut", "k_blur7_v2", "k_pyramid_group"]
    print("  " + os.path.basename(d).replace("prof_", ""))
    print("    " + "  ".join("%s %.1f" % (k.replace("k_", ""), float(rows[k]["AverageNs"]) / 1e3) for k in want if k in rows))
