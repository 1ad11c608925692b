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
            for M in (8, 16):
                for win in (None, 64):
                    od2, log2, r2 = rounds_listers(L, M=M, window=win)
                    ok = owners(log0) == owners(log2) and od0 == od2
                    nw = sum(1 for l in L if len(l) > M and l[M - 1][0] <= TH_LOW)
                    out.append("listers M=%d win=%s: %d rounds%s (wild %d)" % (M, win, r2, "" if ok else " MISMATCH", nw))
            print("; ".join(out))
