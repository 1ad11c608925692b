#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/collect_pmc.sh) into one JSON: per kernel and per launch,
FETCH_SIZE / WRITE_SIZE in bytes (raw counter x 1024; the gfx950 factor-2 under-count of FETCH_SIZE for wide
coalesced streams, MI355X_MICROARCH.md 'HBM', is NOT applied here -- it is recorded as a separate field) and
the SQ counters."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
out = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq", "sq2"):
    files = glob.glob("%s/%s/*/*counter_collection.csv" % (root, sub))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):  # template instantiations are printed with their return type
            k = k[5:]
        k = k.split("<")[0]
        if k.startswith("k_"):
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
        out[k]["launches_" + sub] = len(v)
B = sys.argv[2] if len(sys.argv) > 2 else "16"
NF = sys.argv[3] if len(sys.argv) > 3 else "2000"
W = sys.argv[4] if len(sys.argv) > 4 else "1241"
H = sys.argv[5] if len(sys.argv) > 5 else "376"
res = {"workload": "run_extract_loop.py: %sx%s, %s features, %s images per launch, 5 launches (single context)" % (W, H, NF, B),
       "kernels": {}}
for k, d in sorted(out.items()):
    e = dict(d)
    if "FETCH_SIZE" in d:
        e["fetch_bytes_raw"] = d["FETCH_SIZE"] * 1024
        e["fetch_bytes_x2_wide_stream_correction"] = d["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in d:
        e["write_bytes"] = d["WRITE_SIZE"] * 1024
    res["kernels"][k] = e
json.dump(res, open("%s/summary.json" % root, "w"), indent=1, sort_keys=True)
print(json.dumps(res["kernels"], indent=1, sort_keys=True)[:3000])
