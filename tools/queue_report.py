#!/usr/bin/env python3
"""Which hardware queues the kernels of a rocprofv3 kernel trace ran on: per Queue_Id the kernel count, and per stream
(Stream_Id if present) its queue ids.  usage: _queue_report.py <dir>"""
import csv, glob, sys
from collections import Counter, defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            rows.append(r)
print("columns:", list(rows[0].keys()))
q = Counter(r.get("Queue_Id") for r in rows)
print("queues:", q)
if "Stream_Id" in rows[0]:
    m = defaultdict(Counter)
    for r in rows:
        m[r["Stream_Id"]][r["Queue_Id"]] += 1
    for s, c in sorted(m.items()):
        print("stream", s, dict(c))
names = defaultdict(Counter)
for r in rows:
    names[r["Queue_Id"]][r["Kernel_Name"].split("(")[0][:30]] += 1
for k, c in names.items():
    print("queue", k, c.most_common(4))
