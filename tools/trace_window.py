#!/usr/bin/env python3
"""Print a time window of a rocprofv3 --kernel-trace [--hip-trace] csv, one line per kernel (one column per queue) and,
if the API trace is there, one line per host synchronisation call and per first launch after it.
usage: trace_window.py <dir> [start_fraction] [window_us]"""
import csv
import glob
import sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
win = float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
a = t0 + (t1 - t0) * frac
qs = sorted({r[2] for r in rows})
out = []
for s, e, q, k in rows:
    if a <= s <= a + win * 1000:
        out.append((s, "%8.1f %8.1f  %s%-26s %6.1f" % ((s - a) / 1000, (e - a) / 1000, " " * (28 * qs.index(q)), k, (e - s) / 1000)))
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    prev_sync = False
    with open(f) as fh:
        for r in csv.DictReader(fh):
            s, e, fn = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]
            is_sync = "Synchronize" in fn or "StreamQuery" in fn
            if a <= s <= a + win * 1000 and (is_sync or (prev_sync and "Launch" in fn)):
                out.append((s, "%8.1f %8.1f  HOST %s (%.1f us)" % ((s - a) / 1000, (e - a) / 1000, fn, (e - s) / 1000)))
            if "Launch" in fn:
                prev_sync = False
            if is_sync:
                prev_sync = True
out.sort()
for _, line in out:
    print(line)
