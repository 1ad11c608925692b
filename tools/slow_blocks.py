#!/usr/bin/env python3
"""what the blocks well under the median are.  Input: bench.py --stamp-dump PREFIX files.
For every block of `steps` consecutive steps: its rate, the largest delivery-to-delivery gap inside it, and where the host
thread was during that gap (inside enqueue, inside the wait, or between calls = descheduled / collecting garbage)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    K = d["steps"]
    ev = d["events"]
    waits = [e for e in ev if e[0] == "w"]
    enq = {e[1]: e for e in ev if e[0] == "e"}
    deliv = [w[3] for w in waits]
    gaps = [deliv[i] - deliv[i - 1] for i in range(1, len(deliv))]
    gaps_sorted = sorted(gaps)
    med = gaps_sorted[len(gaps) // 2]
    blocks = []
    for r in range(1, len(deliv) // K):
        lo, hi = r * K - 1, (r + 1) * K - 1
        dur = deliv[hi] - deliv[lo]
        g = [(gaps[i - 1], i) for i in range(lo + 1, hi + 1)]
        gmax, imax = max(g)
        w = waits[imax]
        e = enq.get(w[1] + 3)  # the enqueue that ran just before this wait (NCTX - 1 = 3 steps ahead)
        blocks.append(dict(block=r, ms=dur * 1e3, largest_gap_us=gmax * 1e6, wait_us=(w[3] - w[2]) * 1e6,
                           enqueue_before_us=(e[3] - e[2]) * 1e6 if e else None,
                           between_calls_us=(w[2] - e[3]) * 1e6 if e else None))
    bs = sorted(blocks, key=lambda b: -b["ms"])
    bmed = sorted(b["ms"] for b in blocks)[len(blocks) // 2]
    print("%s: %d steps, median step gap %.1f us, p99 %.1f us, max %.1f us; blocks of %d: median %.3f ms, slowest %.3f ms (%.1f %% below the median rate)"
          % (path, len(deliv), med * 1e6, gaps_sorted[int(len(gaps) * 0.99)] * 1e6, gaps_sorted[-1] * 1e6, K, bmed, bs[0]["ms"],
             100 * (1 - bmed / bs[0]["ms"])))
    print("  gc during the timed region: %s" % d.get("gc"))
    for b in bs[:6]:
        print("  block %4d: %.3f ms; largest gap %.0f us (median %.0f): host waited %.0f us in the wait, %s us in the enqueue before it, %s us between the two calls"
              % (b["block"], b["ms"], b["largest_gap_us"], med * 1e6, b["wait_us"],
                 "%.0f" % b["enqueue_before_us"] if b["enqueue_before_us"] is not None else "-",
                 "%.0f" % b["between_calls_us"] if b["between_calls_us"] is not None else "-"))
