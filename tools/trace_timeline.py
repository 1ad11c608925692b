#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace csv: how busy the GPU is and how kernels overlap.

usage: trace_timeline.py <dir with *_kernel_trace.csv> [skip_fraction]
Prints: per kernel count / avg / total; the union of all kernel intervals (GPU "something is running" time),
the average number of kernels in flight, and the time share during which each kernel is the ONLY one running.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + (t1 - t0) * skip  # drop start-up / warm-up
    rows = [r for r in rows if r[0] >= lo]
    span = max(r[1] for r in rows) - rows[0][0]
    per = defaultdict(lambda: [0, 0])
    ev = []
    for s, e, k in rows:
        per[k][0] += 1
        per[k][1] += e - s
        ev.append((s, 1, k))
        ev.append((e, -1, k))
    ev.sort()
    active = defaultdict(int)
    nact = 0
    busy = 0
    conc = 0
    alone = defaultdict(int)
    last = ev[0][0]
    for t, dlt, k in ev:
        dt = t - last
        if nact > 0:
            busy += dt
            conc += dt * nact
            if nact == 1:
                only = [kk for kk, v in active.items() if v > 0][0]
                alone[only] += dt
        active[k] += dlt
        nact += dlt
        last = t
    out = {"span_ms": span / 1e6, "busy_frac": busy / span, "avg_kernels_in_flight_when_busy": conc / max(busy, 1),
           "kernels": {k: {"n": v[0], "avg_us": v[1] / v[0] / 1e3, "total_ms": v[1] / 1e6,
                           "alone_ms": alone[k] / 1e6} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
