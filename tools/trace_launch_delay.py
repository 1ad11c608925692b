#!/usr/bin/env python3
"""rocprofv3 --hip-trace --kernel-trace csv: for every kernel the delay between its launch call on the host and its
start on the GPU, by kernel name.   usage: trace_launch_delay.py <dir> [skip_fraction]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
api = {}
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "Launch" in r["Function"]:
                api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:26], r["Correlation_Id"], r.get("Queue_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo = t0 + (t1 - t0) * skip
by = defaultdict(list)
for s, e, k, cid, q in rows:
    if s >= lo and cid in api:
        by[k].append((s - api[cid][0]) / 1000.0)
print("kernel: launch call -> start on the GPU, us (mean / p50 / p90 / n)")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    v.sort()
    print("  %-28s %8.1f %8.1f %8.1f %6d" % (k, sum(v) / len(v), v[len(v) // 2], v[int(len(v) * 0.9)], len(v)))
