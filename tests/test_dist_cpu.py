"""CPU: the N>1 path (frame sharding + the one exchange of packed result slots) on gloo, world_size 2/3, driving the SAME
object bench.py's RCCL branch drives (vi_slam_amd.dist.SlotExchange) with REAL packed slots: every rank extracts its
frames with the oracle, packs them in the k_pack_slots layout, exchanges, matches each frame against its predecessor
out of the received buffer, and the matches must equal a single process walking the whole sequence."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vi_slam_amd import dist as vd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, NF, CAP = 320, 240, 300, 344


def test_predecessor_is_video_order():
    for world in (1, 2, 4, 8):
        for batch in (1, 3, 16):
            frames = {}
            for r in range(world):
                for s in range(batch):
                    frames[vd.global_frame(r, s, world, batch)] = (r, s)
            assert sorted(frames) == list(range(world * batch))
            for g in range(world * batch):
                r, s = frames[g]
                pr, ps, prev_step = vd.predecessor(r, s, world, batch)
                if g == 0:
                    assert prev_step and (pr, ps) == frames[world * batch - 1]
                else:
                    assert not prev_step and (pr, ps) == frames[g - 1]


def _extract(step_frame):
    from oracle import orbo
    from vi_slam_amd import synth
    k, d, mono = orbo.Extractor(NF).compute(synth.make_frame(W, H, seed=77, step=step_frame), lap=(0, 1000))
    return k, d, mono


def _match(prev, cur):
    from oracle import orbo
    n, m12, _ = orbo.search_for_initialization(prev[0], prev[1], cur[0], cur[1], W, H, window=100, nnratio=0.9)
    return n, m12


def _single_process_reference(world, batch, steps):
    feats = [_extract(g) for g in range(world * batch * steps)]
    return {g: _match(feats[g - 1], feats[g]) for g in range(1, len(feats))}


def _worker(rank, world, port, batch, steps, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sb = vd.slot_bytes_for(CAP)
        x = vd.SlotExchange.create(rank, world, 0, mode=mode, transport="gloo")
        out = {}
        prev_recv = None
        for step in range(steps):  # two steps: slot 0 of rank 0 needs the previous step's buffer
            own = []
            send = torch.zeros(sb, dtype=torch.uint8)  # the right neighbour needs this rank's LAST frame only
            for s in range(batch):
                g = step * world * batch + vd.global_frame(rank, s, world, batch)
                k, d, mono = _extract(g)
                own.append((k, d))
                if s == batch - 1:
                    send[:] = torch.from_numpy(vd.pack_slot_host(k, d, mono, CAP, sb))
            recv = torch.zeros(sb * (world if mode == "allgather" else 1), dtype=torch.uint8)
            x.exchange(None, send, recv)
            for s in range(batch):
                g = step * world * batch + vd.global_frame(rank, s, world, batch)
                pr, ps, from_prev = vd.predecessor(rank, s, world, batch)
                if s > 0:
                    assert (pr, ps, from_prev) == (rank, s - 1, False)  # own previous slot
                    prev = own[ps]
                else:
                    assert pr == (rank - 1) % world and ps == batch - 1  # the left neighbour's last frame
                    if from_prev:
                        if prev_recv is None:
                            continue
                        blk = x.left_block(prev_recv)
                    else:
                        blk = x.left_block(recv)
                    pk, pd, _ = vd.unpack_slot_host(vd.slot_view(blk, 0, sb).numpy())
                    prev = (pk, pd)
                n, m12 = _match(prev, own[s])
                out[g] = (n, m12.tolist())
            prev_recv = recv
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "ring"), (3, "ring"), (2, "allgather")])
def test_exchange_delivers_predecessors_and_matches_equal_single_process(world, mode):
    batch, steps = 2, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, steps, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        _, out = q.get(timeout=240)
        got.update(out)
    for p in procs:
        p.join(60)
    want = _single_process_reference(world, batch, steps)
    assert sorted(got) == sorted(want)  # every frame but the very first found its predecessor
    for g in want:
        assert got[g][0] == want[g][0] and got[g][1] == want[g][1].tolist(), g
    assert sum(v[0] for v in want.values()) > 20 * len(want)


def _verify_worker(rank, world, port, batch, mode, q):
    """the proof bench.py --verify-exchange makes on the GPU, with the oracle as the extractor: what arrives through the
    exchange equals the slot this rank produces ITSELF from the left neighbour's frame"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sb = vd.slot_bytes_for(CAP)
        x = vd.SlotExchange.create(rank, world, 0, mode=mode, transport="gloo")
        k, d, mono = _extract(vd.global_frame(rank, batch - 1, world, batch))
        send = torch.from_numpy(vd.pack_slot_host(k, d, mono, CAP, sb))
        recv = torch.zeros(sb * (world if mode == "allgather" else 1), dtype=torch.uint8)
        x.exchange(None, send, recv)
        lr, ls, lg = vd.left_last_frame(rank, world, batch)
        assert (lr, ls) == ((rank - 1) % world, batch - 1)
        lk, ld, lmono = _extract(lg)
        want = vd.pack_slot_host(lk, ld, lmono, CAP, sb)
        got = vd.slot_view(x.left_block(recv), 0, sb).numpy()
        good = vd.compare_packed_slots(got, want)
        bad = got.copy()
        bad[16 + 28 * 3 + 5] ^= 1  # one bit of keypoint 3
        bad2 = got.copy()
        bad2[16 + CAP * 28 + 32 * (good["n"] - 1)] ^= 0x80  # one bit of the last descriptor
        q.put((rank, good, vd.compare_packed_slots(bad, want), vd.compare_packed_slots(bad2, want),
               vd.compare_packed_slots(want, vd.pack_slot_host(k, d, mono, CAP, sb))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "ring"), (2, "allgather")])
def test_verify_exchange_compares_the_arrived_slot_with_a_local_extraction(world, mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_verify_worker, args=(r, world, port, 2, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, good, bad, bad2, other in res:
        assert good["ok"] and good["n"] > 100, (rank, good)
        assert not bad["ok"] and "keypoint 3" in bad["reason"]
        assert not bad2["ok"] and "descriptor %d" % (good["n"] - 1) in bad2["reason"]
        assert not other["ok"]  # the rank's own last frame is not the left neighbour's


def test_world1_exchange_is_a_copy_without_process_group():
    x = vd.SlotExchange.create(0, 1, 0, mode="ring", transport="gloo")
    a = torch.arange(64, dtype=torch.uint8)
    b = torch.zeros(64, dtype=torch.uint8)
    x.exchange(None, a, b)
    assert torch.equal(a, x.left_block(b))


def test_pack_unpack_slot_roundtrip():
    k, d, mono = _extract(0)
    buf = vd.pack_slot_host(k, d, mono, CAP)
    k2, d2, m2 = vd.unpack_slot_host(buf)
    assert m2 == mono and np.array_equal(d, d2) and all(np.array_equal(k[f], k2[f]) for f in k.dtype.names)


def test_bench_self_launches_n_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no launcher spawns torch.distributed.run itself;
    --launch-check makes the spawned ranks only rendezvous (gloo) and all-reduce, so this runs without a GPU."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"launch_check": "ok", "world": 2, "sum_of_ranks_plus_1": 3.0}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--print-launch"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "--nproc-per-node=4" in r.stdout and "torch.distributed.run" in r.stdout


def test_bench_self_launch_propagates_failure():
    """without a GPU the spawned ranks fail: the parent must exit non-zero, not hang and not print a number"""
    import torch as _t
    if _t.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]


_DEAD_PEER_RANK = r'''
import os, sys, time
sys.path.insert(0, sys.argv[4])
import torch.distributed as dist
from vi_slam_amd import dist as vd
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
dist.init_process_group("gloo", rank=rank, world_size=world)
if rank == world - 1:
    os._exit(0)            # this rank dies before the set-up vote
time.sleep(1.0)
vd.SlotExchange.create(rank, world, 0, mode="ring", transport="gloo", setup_timeout=10.0)
print("rank %d: create returned" % rank)
'''


def test_a_rank_that_dies_before_the_vote_makes_the_others_exit_nonzero(tmp_path):
    """nothing may wait for ever in SlotExchange.create.  Three gloo ranks; the last one exits right
    after the rendezvous; the other two must leave with a non-zero code well inside the timeout (either gloo reports the
    lost peer in the vote's all-reduce, or the set-up deadline fires: vi_slam_amd.dist.EXIT_SETUP_TIMEOUT)."""
    import time
    script = tmp_path / "dead_peer_rank.py"
    script.write_text(_DEAD_PEER_RANK)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 3
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), ROOT], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    codes, outs = [], []
    for p in procs:
        try:
            o, e = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
            o += "\n<<killed by the test after 90 s>>"
        codes.append(p.returncode)
        outs.append((o + e)[-500:])
    assert codes[world - 1] == 0
    for r in range(world - 1):
        assert codes[r] not in (0, None) and codes[r] > 0, (r, codes, outs[r])
        assert "create returned" not in outs[r]
    assert time.time() - t0 < 80, codes


def test_deadline_is_disarmed_when_the_setup_finishes():
    with vd._Deadline(0.3, "a block that finishes in time", 0):
        pass
    import time
    time.sleep(0.6)  # the watcher must not fire afterwards
