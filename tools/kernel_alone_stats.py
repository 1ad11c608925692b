#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 --kernel-trace csv of a SINGLE-context loop (tools/run_extract_loop.py):
nothing else is on the GPU, so these are the kernels' own times.  Kernels are keyed by (name, grid size) so the two
launches of k_pyramid_group show up separately.   usage: kernel_alone_stats.py <dir> [skip_fraction]"""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        g = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], g,
                     r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("SGPR_Count", "")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
rows = [r for r in rows if r[0] >= t0 + (t1 - t0) * skip]
acc = defaultdict(list)
for s, e, k, g, lds, vg, sg in rows:
    acc[(k, g, lds, vg, sg)].append((e - s) / 1e3)
out = []
for (k, g, lds, vg, sg), v in acc.items():
    v.sort()
    out.append({"kernel": k, "grid": g, "lds": lds, "vgpr": vg, "sgpr": sg, "n": len(v), "avg_us": round(sum(v) / len(v), 2),
                "min_us": round(v[0], 2), "med_us": round(v[len(v) // 2], 2)})
out.sort(key=lambda e: -e["avg_us"] * e["n"])
for e in out:
    print(json.dumps(e))
