#!/usr/bin/env python3
"""Round counts of SearchForInitialization replay schemes, simulated on the oracle's data (design aid for k_si_replay).
usage: python tests/tools/replay_sim.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import orbo
from vi_slam_amd import synth

TH_LOW, INF = 50, 1 << 30


def lists_for(k1, d1, k2, d2, W, H, window=100):
    """per octave-0 query of frame 1: list of (dist, slot) sorted by (dist, GetFeaturesInArea order)"""
    q = [i for i in range(len(k1)) if k1["octave"][i] == 0]
    L = []
    for i in q:
        idx = orbo.grid_query(k2, W, H, float(k1["x"][i]), float(k1["y"][i]), float(window), 0, 0)
        ent = []
        for pos, j in enumerate(idx):
            dist = int(np.unpackbits(d1[i] ^ d2[j]).sum())
            ent.append((dist, pos, int(j)))
        ent.sort()
        L.append([(d, j) for d, _, j in ent])
    return q, L


def decide(lst, od, nnratio):
    """the reference's loop body: -> (accept, s1, d1, s2, d2) from the first two non-skipped entries"""
    first = second = None
    for d, s in lst:
        if od.get(s, INF) <= d:
            continue
        if first is None:
            first = (d, s)
        elif second is None:
            second = (d, s)
            break
    if first is None:
        return False, None, None, None, None
    d2 = second[0] if second else INF
    acc = first[0] <= TH_LOW and float(np.float32(first[0])) < float(np.float32(d2) * np.float32(nnratio))
    return acc, first[1], first[0], (second[1] if second else None), (second[0] if second else None)


def sequential(L, nnratio=0.9):
    od, log = {}, []
    for qi, lst in enumerate(L):
        acc, s1, d1, _, _ = decide(lst, od, nnratio)
        if acc:
            od[s1] = d1
            log.append((qi, s1))
    return od, log


def rounds_prefix64(L, nnratio=0.9):
    """the kernel as it is: windows of 64 queries, commit the prefix in front of the first lane that must wait"""
    od, log, rounds = {}, [], 0
    for qb in range(0, len(L), 64):
        pend = list(range(qb, min(qb + 64, len(L))))
        while pend:
            rounds += 1
            dec = {g: decide(L[g], od, nnratio) for g in pend}
            mark, markd = {}, {}
            for g in pend:
                acc, s1, d1, _, _ = dec[g]
                if acc:
                    mark[s1] = min(mark.get(s1, INF), g)
                    markd[s1] = min(markd.get(s1, INF), d1)
            f = None
            for g in pend:
                acc, s1, d1, s2, d2 = dec[g]
                if s1 is None:
                    continue
                stop = mark.get(s1, INF) < g and markd.get(s1, INF) <= d1
                if s2 is not None and mark.get(s2, INF) < g and markd.get(s2, INF) <= d2:
                    stop = True
                if stop:
                    f = g
                    break
            com = [g for g in pend if f is None or g < f]
            for g in com:
                acc, s1, d1, _, _ = dec[g]
                if acc:
                    od[s1] = min(od.get(s1, INF), d1)
                    log.append((g, s1))
            pend = [g for g in pend if g not in set(com)]
    return od, log, rounds


def rounds_listers(L, M=8, nnratio=0.9, window=None, th_filter=True):
    """all pending queries at once (or windows of `window`): a query commits when no earlier PENDING query lists
    (non-skipped, within its first M, distance <= TH_LOW if th_filter) its best or second-best slot; queries whose
    list is full with the M-th distance <= TH_LOW are wildcards (everything behind them waits)"""
    od, log, rounds = {}, [], 0
    n = len(L)
    allq = list(range(n))
    wsize = window or n
    for qb in range(0, n, wsize):
        pend = allq[qb:qb + wsize]
        while pend:
            rounds += 1
            lister, lister_all, wild = {}, {}, INF
            dec = {}
            for g in pend:
                dec[g] = decide(L[g], od, nnratio)
                top = L[g][:M]
                if len(L[g]) > M and top[-1][0] <= TH_LOW:
                    wild = min(wild, g)
                for d, s in top:
                    if od.get(s, INF) <= d:
                        continue
                    lister_all[s] = min(lister_all.get(s, INF), g)
                    if th_filter and d > TH_LOW:
                        continue
                    lister[s] = min(lister.get(s, INF), g)
            com = []
            for g in pend:
                if g > wild:
                    break
                acc, s1, d1, s2, d2 = dec[g]
                ok = True
                if s1 is not None and lister.get(s1, INF) < g:
                    ok = False
                if s2 is not None and lister.get(s2, INF) < g:
                    ok = False
                if acc and lister_all.get(s1, INF) < g:  # an earlier pending query still reads this slot
                    ok = False
                if ok:
                    com.append(g)
            for g in com:
                acc, s1, d1, _, _ = dec[g]
                if acc:
                    od[s1] = min(od.get(s1, INF), d1)
                    log.append((g, s1))
            cs = set(com)
            pend = [g for g in pend if g not in cs]
    return od, log, rounds


def rounds_kernel_scheme(L, M=8, nnratio=0.9, slot_capacity=4):
    """k_si_replay as built: every undecided query per round against the acceptances of EARLIER queries only (per-slot
    (writer, distance) entries), sorted prefixes of length M, re-scan of the whole window when the prefix runs out (only as
    the smallest undecided query), wildcard blocking, slot entries merged when full.  -> (od, log, rounds, rescans)"""
    n = len(L)
    slots = {}  # slot -> list of (writer or -1 for merged, d)
    pend = list(range(n))
    log, rounds, rescans = [], 0, 0

    def view(q, s):
        return min([d for w, d in slots.get(s, []) if w < q], default=INF)

    while pend:
        rounds += 1
        minp = pend[0]
        lister, wild_min, dec = {}, INF, {}
        for q in pend:
            top = L[q][:M]
            full = len(top) == M
            first = second = None
            for d, s in top:
                if view(q, s) <= d:
                    continue
                if d <= TH_LOW:
                    lister[s] = min(lister.get(s, INF), q)
                if first is None:
                    first = (d, s)
                elif second is None:
                    second = (d, s)
            dlast = top[-1][0] if top else 0
            wild = full and dlast <= TH_LOW
            if wild:
                wild_min = min(wild_min, q)
            accept = scan = False
            if first is None:
                scan = wild
            elif first[0] <= TH_LOW:
                d2 = second[0] if second else INF
                if second is None and full:
                    if float(np.float32(first[0])) < float(np.float32(dlast) * np.float32(nnratio)):
                        d2 = dlast
                    else:
                        scan = True
                if not scan:
                    accept = float(np.float32(first[0])) < float(np.float32(d2) * np.float32(nnratio))
            dec[q] = (accept, scan, first, second)
        done = set()
        scanq = None
        for q in pend:
            accept, scan, first, second = dec[q]
            if scan:
                if q == minp:
                    scanq = q
                continue
            blocked = q > wild_min
            if first is not None and lister.get(first[1], INF) < q:
                blocked = True
            if second is not None and lister.get(second[1], INF) < q:
                blocked = True
            if blocked:
                continue
            if accept:
                ent = slots.setdefault(first[1], [])
                if len(ent) >= slot_capacity:
                    if q != minp:
                        continue
                    assert all(w < q for w, _ in ent)
                    ent[:] = [(-1, min(d for _, d in ent))]
                ent.append((q, first[0]))
                log.append((q, first[1]))
            done.add(q)
        if scanq is not None:  # the reference's loop body over the WHOLE window, as the smallest undecided query
            rescans += 1
            q = scanq
            od_q = {s: view(q, s) for _, s in L[q]}
            acc, s1, d1, _, _ = decide(L[q], od_q, nnratio)
            if acc:
                ent = slots.setdefault(s1, [])
                if len(ent) >= slot_capacity:
                    assert all(w < q for w, _ in ent)
                    ent[:] = [(-1, min(d for _, d in ent))]
                ent.append((q, d1))
                log.append((q, s1))
            done.add(q)
        assert minp in done  # the smallest undecided query always commits
        pend = [q for q in pend if q not in done]
    od = {s: min(d for _, d in e) for s, e in slots.items()}
    return od, log, rounds, rescans


def owners(log):
    o = {}
    for g, s in sorted(log):
        o[s] = g
    return o


if __name__ == "__main__":
    for nf in (1000, 2000):
        W, H = 1241, 376
        e = orbo.Extractor(nf)
        fr = [e.compute(synth.make_frame(W, H, seed=5, step=s), lap=(0, 1000)) for s in range(4)]
        for s in range(1, 4):
            k1, d1, _ = fr[s - 1]
            k2, d2, _ = fr[s]
            q, L = lists_for(k1, d1, k2, d2, W, H)
            od0, log0 = sequential(L)
            od1, log1, r1 = rounds_prefix64(L)
            assert owners(log0) == owners(log1) and od0 == od1
            out = ["N=%d pair %d: %d queries, avg list %.1f, accepted %d; prefix64 rounds %d" % (nf, s, len(L), np.mean([len(l) for l in L]), len(log0), r1)]
            for M in (8, 2):
                od3, log3, r3, rs3 = rounds_kernel_scheme(L, M=M)
                ok = owners(log0) == owners(log3) and od0 == od3
                out.append("kernel scheme M=%d: %d rounds, %d re-scans%s" % (M, r3, rs3, "" if ok else " MISMATCH"))
            for M in (8, 16):
                for win in (None, 64):
                    od2, log2, r2 = rounds_listers(L, M=M, window=win)
                    ok = owners(log0) == owners(log2) and od0 == od2
                    nw = sum(1 for l in L if len(l) > M and l[M - 1][0] <= TH_LOW)
                    out.append("listers M=%d win=%s: %d rounds%s (wild %d)" % (M, win, r2, "" if ok else " MISMATCH", nw))
            print("; ".join(out))
