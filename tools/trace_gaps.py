#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 --kernel-trace csv: for every HIP stream (hardware queue) the share of time a kernel
of that queue is running, and the gap between the end of one kernel and the start of the next one on the same queue,
by successor kernel.   usage: trace_gaps.py <dir with *_kernel_trace.csv> [skip_fraction]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
byq = defaultdict(list)
for s, e, k, q in rows:
    byq[q].append((s, e, k))
gap_by = defaultdict(list)
for q, L in sorted(byq.items()):
    span = L[-1][1] - L[0][0]
    run = sum(e - s for s, e, _ in L)
    gaps = []
    for (s0, e0, k0), (s1, e1, k1) in zip(L[:-1], L[1:]):
        g = (s1 - e0) / 1000.0
        gaps.append(g)
        gap_by[(k0[:22], k1[:22])].append(g)
    gaps.sort()
    print("queue %s: %d kernels, running %.0f %% of its span, gap mean %.1f us p50 %.1f p90 %.1f" %
          (q, len(L), 100.0 * run / span, sum(gaps) / len(gaps), gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)]))
print("gap before a kernel (by predecessor -> successor), us: mean / p50 / n")
for (k0, k1), g in sorted(gap_by.items(), key=lambda kv: -sum(kv[1])):
    g.sort()
    print("  %-22s -> %-22s %7.1f %7.1f %6d" % (k0, k1, sum(g) / len(g), g[len(g) // 2], len(g)))
