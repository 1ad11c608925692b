"""CPU: the N>1 path of bench.py (frame sharding + the one ring shift of result slots) on gloo, world_size 2/3."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vi_slam_amd import dist as vd


def test_predecessor_is_video_order():
    for world in (1, 2, 4, 8):
        for batch in (1, 3, 16):
            frames = {}
            for r in range(world):
                for s in range(batch):
                    frames[vd.global_frame(r, s, world)] = (r, s)
            assert sorted(frames) == list(range(world * batch))
            for g in range(world * batch):
                r, s = frames[g]
                pr, ps, prev_step = vd.predecessor(r, s, world, batch)
                if g == 0:
                    assert prev_step and (pr, ps) == frames[world * batch - 1]
                else:
                    assert not prev_step and (pr, ps) == frames[g - 1]


def _worker(rank, world, port, batch, slot_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        prev_left = None
        for step in range(2):  # two steps: slot 0 of rank 0 needs the previous step's buffer
            local = torch.zeros(batch * slot_bytes, dtype=torch.uint8)
            for s in range(batch):  # stamp every slot with its global frame id and a payload derived from it
                g = step * world * batch + vd.global_frame(rank, s, world)
                v = vd.slot_view(local, s, slot_bytes)
                v[:4] = torch.from_numpy(np.array([g], np.int32).view(np.uint8))
                v[4:] = (g * 7 + 3) % 251
            left = torch.zeros(batch * slot_bytes, dtype=torch.uint8)
            vd.shift_slots(local, left)
            for s in range(batch):
                pr, ps, prev = vd.predecessor(rank, s, world, batch)
                assert pr == (rank - 1) % world  # every predecessor lives on the left neighbour
                if prev:
                    if prev_left is None:
                        continue
                    v = vd.slot_view(prev_left, ps, slot_bytes)
                else:
                    v = vd.slot_view(left, ps, slot_bytes)
                g = int(v[:4].numpy().view(np.int32)[0])
                ok &= g == step * world * batch + vd.global_frame(rank, s, world) - 1
                ok &= bool((v[4:] == (g * 7 + 3) % 251).all())
            prev_left = left
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_shift_delivers_predecessors_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 3, 512, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(world)]


def test_world1_is_a_copy_without_process_group():
    a = torch.arange(64, dtype=torch.uint8)
    b = torch.zeros(64, dtype=torch.uint8)
    vd.shift_slots(a, b)
    assert torch.equal(a, b)
