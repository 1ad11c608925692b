"""The round scheme of k_si_replay (vslam_init_kernel.hip), modelled on the CPU (tests/tools/replay_sim.py): every
undecided query per round, acceptances visible to LATER queries only, commit when no earlier undecided query can still
take the best / second-best slot, re-scan and wildcard rules, four entries per slot with merging.  It must reproduce the
sequential loop of fmatcher.cpp:1003-1049 (ownership and vMatchedDistance) on extractor output and on hand-made crowded
frames; the GPU tests compare the kernel itself with the oracle."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import orbo
from vi_slam_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("replay_sim", os.path.join(HERE, "tools", "replay_sim.py"))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


def _crowded_lists(rng, n1, n2, nproto, q_flips, s_flips):
    """every query sees every slot (one window): lists straight from descriptor distances, slot order = index"""
    def flip(d, nb):
        d = d.copy()
        for b in rng.choice(256, nb, replace=False):
            d[b >> 3] ^= 1 << (b & 7)
        return d
    proto = rng.integers(0, 256, (nproto, 32), dtype=np.uint8)
    d2 = [flip(proto[rng.integers(nproto)], int(rng.integers(0, s_flips + 1))) for _ in range(n2)]
    d1 = [flip(proto[rng.integers(nproto)], int(rng.integers(0, q_flips + 1))) for _ in range(n1)]
    L = []
    for a in d1:
        ent = sorted((int(np.unpackbits(a ^ b).sum()), j) for j, b in enumerate(d2))
        L.append(ent)
    return L


@pytest.mark.parametrize("M", [8, 2, 1, 16])
def test_scheme_equals_the_sequential_loop_on_crowded_frames(M):
    rng = np.random.default_rng(100 + M)
    total_rounds = total_q = 0
    for case in range(60):
        L = _crowded_lists(rng, int(rng.integers(5, 120)), int(rng.integers(2, 60)), int(rng.integers(1, 6)),
                           int(rng.integers(0, 50)), int(rng.integers(0, 30)))
        od0, log0 = sim.sequential(L)
        od1, log1, rounds, rescans = sim.rounds_kernel_scheme(L, M=M)
        assert sim.owners(log0) == sim.owners(log1) and od0 == od1, (M, case)
        total_rounds += rounds
        total_q += len(L)
    assert total_rounds <= total_q  # never worse than one query per round


def test_scheme_on_a_steal_chain_overflows_the_slot_entries():
    # every query a little closer to slot 0 than the one before: 44 acceptances of one slot, four entries per slot
    L = [[(max(45 - i, 1), 0), (200, 1), (210, 2)] for i in range(60)]
    od0, log0 = sim.sequential(L)
    od1, log1, rounds, _ = sim.rounds_kernel_scheme(L, M=8)
    assert sim.owners(log0) == sim.owners(log1) and od0 == od1 and len(log0) >= 44
    assert rounds >= 44  # fully dependent: one per round


def test_scheme_on_extractor_output_takes_few_rounds():
    W, H = 640, 360
    e = orbo.Extractor(500)
    fr = [e.compute(synth.make_frame(W, H, seed=5, step=s), lap=(0, 1000)) for s in range(3)]
    for s in (1, 2):
        q, L = sim.lists_for(fr[s - 1][0], fr[s - 1][1], fr[s][0], fr[s][1], W, H)
        od0, log0 = sim.sequential(L)
        od1, log1, r1 = sim.rounds_prefix64(L)
        od2, log2, r2, _ = sim.rounds_kernel_scheme(L, M=8)
        assert sim.owners(log0) == sim.owners(log1) == sim.owners(log2) and od0 == od1 == od2
        assert len(L) > 50 and r2 <= r1 and r2 <= 12
