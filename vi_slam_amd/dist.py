"""Frame sharding across GPUs and the one exchange step of the path.

Extraction and stereo matching are independent per frame.  Cross-frame matching (mono initialisation, frame.cpp:289 +
fmatcher.cpp:983) needs the predecessor frame's keypoints and descriptors, so the frames of a step are dealt in BLOCKS:
rank r holds the `batch` consecutive frames r*batch .. (r+1)*batch - 1 of the step's world*batch.  Every predecessor
but one is then the rank's own previous slot; only slot 0 needs a frame from elsewhere -- the LAST slot of the left
neighbour (rank - 1 mod world; for rank 0 that is rank world-1's last frame of the PREVIOUS step).  The exchange is
therefore a ring shift of ONE packed result slot per rank and step (62-125 KB): every rank sends its last slot to
rank + 1 and receives rank - 1's.  (Round-robin dealing -- frame g on rank g % world -- put EVERY predecessor on the left
neighbour: the same ring shift with batch times the payload, a 2-4 MB pack kernel in front of it, and 25 % of the
single-GPU rate gone at world size 1; an all-gather, north_star's literal wording, moves world times as much again and is
kept as `mode="allgather"`.)  No other collective exists on this path.

`SlotExchange` is the ONE object bench.py, the GPU tests and the gloo CPU tests drive:
  transport "rccl": the library's own communicator (include/vslam_fe.h vslam_comm_*): vslam_exchange_ring =
                    ncclSend/ncclRecv in one group, ENQUEUED on the extractor context's stream (no host sync);
  transport "gloo": host-staged isend/irecv pair (CPU tests here; one-GPU rehearsals of the N>1 path).
Which slot of which buffer holds a frame's predecessor (`predecessor`, `left_block`, `slot_view`) is the same
code for both transports.
"""
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

#: exit code of a rank that gave up waiting for its peers while the exchange was being set up (see `_Deadline`)
EXIT_SETUP_TIMEOUT = 70


class _Deadline:
    """Bounded wait for the collective set-up.  ncclCommInitRank, the probe's synchronize and the vote's all-reduce all
    block for ever if a peer died before it got there, and none of them can be cancelled from Python; what can be bounded
    is the PROCESS: if the guarded block has not finished after `seconds`, the rank says why and exits with
    EXIT_SETUP_TIMEOUT (os._exit: no re-exec, no second attempt -- the launcher sees a non-zero rank and tears the job
    down).  Disarmed on normal exit of the block."""

    def __init__(self, seconds, what, rank):
        self.seconds, self.what, self.rank = float(seconds), what, rank
        self._done = threading.Event()
        self._lock = threading.Lock()
        self._expires = time.monotonic() + self.seconds

    def rearm(self, what=None):
        """A step of the guarded block has finished (e.g. one lane's communicator is up and voted on): the next step
        gets the full `seconds` again, so that a large world whose every lane is slow but healthy is not shot for the sum."""
        with self._lock:
            self._expires = time.monotonic() + self.seconds
            if what:
                self.what = what

    def __enter__(self):
        def watch():
            while True:
                with self._lock:
                    left = self._expires - time.monotonic()
                if left <= 0:
                    break
                if self._done.wait(min(left, 1.0)):
                    return
            sys.stderr.write("vi_slam_amd.dist: rank %d gave up after %.0f s in %s -- a peer did not arrive; exiting %d\n"
                             % (self.rank, self.seconds, self.what, EXIT_SETUP_TIMEOUT))
            sys.stderr.flush()
            os._exit(EXIT_SETUP_TIMEOUT)
        if self.seconds > 0:
            self.rearm()
            threading.Thread(target=watch, daemon=True).start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False


def global_frame(rank, slot, world, batch):
    """Index, inside one step, of the frame held by (rank, slot)."""
    return rank * batch + slot


def predecessor(rank, slot, world, batch):
    """(rank, slot, from_previous_step) of the frame preceding (rank, slot) in video order.  slot > 0: the rank's own
    previous slot.  slot 0: the last slot of the left neighbour, which arrives through the exchange (also at world
    size 1, where the left neighbour is the rank itself one step earlier)."""
    if slot > 0:
        return rank, slot - 1, False
    return (rank - 1) % world, batch - 1, rank == 0


def slot_view(buf, slot, slot_bytes):
    """Slot `slot` of a packed buffer (own slots or the left neighbour's)."""
    return buf[slot * slot_bytes:(slot + 1) * slot_bytes]


# ---- the packed result slot (vslam_fe_pack_slots): int32 n, mono, cap, 0 | vslam_kp[cap] | desc[cap][32]
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def slot_bytes_for(cap):
    return (16 + cap * 60 + 255) & ~255


def pack_slot_host(kps, desc, mono, cap, slot_bytes=None):
    """Host-side twin of k_pack_slots (the GPU test test_pack_slots_layout_* pins the device layout to this one)."""
    sb = slot_bytes or slot_bytes_for(cap)
    n = len(kps)
    assert n <= cap
    out = np.zeros(sb, np.uint8)
    out[:16].view(np.int32)[:] = (n, mono, cap, 0)
    out[16:16 + n * 28] = np.ascontiguousarray(kps).view(np.uint8).reshape(-1)
    off = 16 + cap * 28
    out[off:off + n * 32] = np.ascontiguousarray(desc, np.uint8).reshape(-1)
    return out


def unpack_slot_host(buf):
    """-> (keypoints, descriptors, monoIndex) of one packed slot (numpy uint8 array)."""
    buf = np.ascontiguousarray(buf)
    n, mono, cap, _ = (int(v) for v in buf[:16].view(np.int32))
    kps = buf[16:16 + n * 28].view(KP_DTYPE).copy()
    off = 16 + cap * 28
    return kps, buf[off:off + n * 32].reshape(n, 32).copy(), mono


def left_last_frame(rank, world, batch):
    """(rank, slot, global frame index inside a step) of the frame whose packed slot arrives through the exchange: the
    LAST frame of the left neighbour."""
    lr = (rank - 1) % world
    return lr, batch - 1, global_frame(lr, batch - 1, world, batch)


def compare_packed_slots(got, want):
    """Byte-for-byte comparison of two packed result slots (k_pack_slots layout): header (n, monoIndex, cap), the n
    keypoints and the n descriptors -- what lies behind entry n is never written by the packer and not compared.
    -> dict(ok, n, reason).  Used by bench.py --verify-exchange and its tests: `got` is what arrived through the exchange,
    `want` what this rank produced ITSELF from the left neighbour's frame (frames are synthesised from their global index)."""
    a = np.ascontiguousarray(got).reshape(-1).view(np.uint8)
    b = np.ascontiguousarray(want).reshape(-1).view(np.uint8)
    if a.size < 16 or b.size < 16:
        return dict(ok=False, n=0, reason="slot shorter than its header")
    ha, hb = a[:16].view(np.int32), b[:16].view(np.int32)
    n, cap = int(hb[0]), int(hb[2])
    if not np.array_equal(ha[:3], hb[:3]):
        return dict(ok=False, n=n, reason="header differs: got n=%d mono=%d cap=%d, want n=%d mono=%d cap=%d"
                    % (ha[0], ha[1], ha[2], hb[0], hb[1], hb[2]))
    if n < 0 or n > cap or 16 + cap * 60 > min(a.size, b.size):
        return dict(ok=False, n=n, reason="inconsistent header (n=%d cap=%d, %d bytes)" % (n, cap, min(a.size, b.size)))
    if not np.array_equal(a[16:16 + n * 28], b[16:16 + n * 28]):
        bad = int(np.flatnonzero(a[16:16 + n * 28] != b[16:16 + n * 28])[0]) // 28
        return dict(ok=False, n=n, reason="keypoint %d of %d differs" % (bad, n))
    off = 16 + cap * 28
    if not np.array_equal(a[off:off + n * 32], b[off:off + n * 32]):
        bad = int(np.flatnonzero(a[off:off + n * 32] != b[off:off + n * 32])[0]) // 32
        return dict(ok=False, n=n, reason="descriptor %d of %d differs" % (bad, n))
    return dict(ok=True, n=n, reason="")


class SlotExchange:
    """The exchange step.  exchange(fe, send, recv): `send` = what this rank's right neighbour needs (its last packed
    slot); afterwards left_block(recv) holds what the left neighbour sent.  mode "ring": recv has send's size;
    "allgather": world x that."""

    def __init__(self, rank, world, mode="ring", transport="gloo", comms=None):
        assert mode in ("ring", "allgather") and transport in ("rccl", "gloo", "local")
        self.rank, self.world, self.mode, self.transport = rank, world, mode, transport
        self.comms = list(comms or [])  # rccl: one communicator per lane (extractor context in flight)

    @classmethod
    def create(cls, rank, world, device, mode="ring", transport="rccl", lanes=1, setup_timeout=180.0, probe_limit=30.0):
        """Collective constructor (every rank calls it).  transport "rccl": rank 0 makes the ncclUniqueIds, they travel
        through torch.distributed's broadcast, every rank builds the library's communicators, and ONE probe exchange
        per communicator decides -- by an all-reduce every rank takes part in -- whether the transport works; a failed
        probe of the FIRST communicator is fatal (no collective is ever switched mid-run); a later lane that fails or
        whose probe takes longer than `probe_limit` seconds on any rank ends the lane set-up and the exchange runs on the
        lanes that passed (down to one).  The whole set-up is guarded by `setup_timeout` seconds (`_Deadline`): a rank
        whose peers never arrive exits with EXIT_SETUP_TIMEOUT instead of sitting in ncclCommInitRank for ever.
        lanes: communicators to build.  RCCL orders the operations of ONE communicator across streams (each launch
        waits for the communicator's previous one), which couples extractor contexts that are otherwise independent:
        with one communicator for four contexts in flight the mono workload lost a quarter of its rate at world size 1.
        exchange(..., lane=k) uses communicator k % lanes; every rank must use the same lane for the same step.
        Several communicators with operations in flight on different streams need all those kernels to be able to
        co-reside on the GPU (RCCL's documented condition): the exchanges here are ONE send/recv pair of <= 125 KB per
        lane, one workgroup each, next to kernels that never fill more than their own wave slots."""
        with _Deadline(setup_timeout, "SlotExchange.create (%s, world %d)" % (transport, world), rank) as deadline:
            if transport != "rccl":
                x = cls(rank, world, mode, "gloo" if dist.is_initialized() else "local")
                if world > 1 and dist.is_initialized():  # every rank is here and its group works: one vote, as for rccl
                    # the vote travels on the process group's own backend: an nccl group reduces device tensors only
                    t = torch.tensor([1.0], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return x
            import vi_slam_amd as V
            ctl = "cuda" if dist.get_backend() == "nccl" else "cpu"
            lanes = max(1, int(lanes))
            idt = torch.zeros(lanes * V.COMM_ID_BYTES, dtype=torch.uint8, device=ctl)
            if rank == 0:
                ids = b"".join(bytes(V.Comm.unique_id()) for _ in range(lanes))
                idt.copy_(torch.frombuffer(bytearray(ids), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(idt, 0)
            raw = bytes(idt.cpu().numpy().tobytes())
            x = cls(rank, world, mode, "rccl", [])
            fe = V.FExtractor(100, 1.2, 2, 20, 7, 128, 128, device=device, max_batch=1)  # ONE scratch stream context for all probes
            try:
                for k in range(lanes):  # one communicator at a time, each followed by a vote: all ranks end up with the same number
                    deadline.rearm("SlotExchange.create (rccl, world %d, lane %d of %d)" % (world, k + 1, lanes))  # `setup_timeout` per lane
                    ok, err, comm = 1.0, "", None
                    t0 = time.perf_counter()
                    try:
                        comm = V.Comm(device, rank, world, raw[k * V.COMM_ID_BYTES:(k + 1) * V.COMM_ID_BYTES])
                        x.comms.append(comm)
                        ok = 1.0 if x._probe(fe, k) else 0.0
                    except Exception as e:  # noqa: BLE001 - reported below, after every rank has voted
                        ok, err = 0.0, str(e)
                    took = time.perf_counter() - t0
                    if ok and k > 0 and took > probe_limit:
                        ok, err = 0.0, "lane %d took %.1f s to set up (limit %.0f s)" % (k, took, probe_limit)
                    if world > 1:
                        t = torch.tensor([ok], dtype=torch.float64, device=ctl)
                        dist.all_reduce(t, op=dist.ReduceOp.MIN)
                        ok = float(t.item())
                    if not ok:
                        if comm is not None:
                            x.comms.pop()
                            comm.close()
                        if k == 0:
                            raise RuntimeError("rank %d: the RCCL exchange probe failed on at least one rank (%s)" % (rank, err or "another rank"))
                        if rank == 0:
                            print("SlotExchange: communicator %d of %d could not be set up on every rank (%s); continuing with %d"
                                  % (k + 1, lanes, err or "another rank", k), file=sys.stderr, flush=True)
                        break
            finally:
                fe.close()
            return x

    def _probe(self, fe, lane=0):
        """One exchange of a stamped buffer on a scratch stream context: the left neighbour's stamp must arrive."""
        n = 4096
        send = torch.full((n,), self.rank + 1, dtype=torch.uint8, device="cuda")
        recv = torch.zeros(n * (self.world if self.mode == "allgather" else 1), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        self.exchange(fe, send, recv, lane)
        torch.cuda.synchronize()  # the probe is the only place that waits for an exchange on the host
        want = (self.rank - 1) % self.world + 1
        return bool((self.left_block(recv) == want).all().item())

    # ------------------------------------------------------------------------------------------ the exchange
    def exchange(self, fe, send, recv, lane=0):
        n = send.numel()
        if self.transport == "rccl":
            # enqueue-only, on fe's stream: ordered behind k_pack_slots, ahead of the matcher
            comm = self.comms[lane % len(self.comms)]
            if self.mode == "ring":
                comm.ring(fe, send.data_ptr(), recv.data_ptr(), n)
            else:
                comm.allgather(fe, send.data_ptr(), recv.data_ptr(), n)
            return recv
        if fe is not None and send.is_cuda:
            torch.cuda.synchronize()  # rehearsal path: gloo moves host memory only
        if self.transport == "local" or self.world == 1:
            self.left_block(recv).copy_(send)
            return recv
        src = send.cpu() if send.is_cuda else send
        if self.mode == "ring":
            host = torch.empty(n, dtype=send.dtype)
            ops = [dist.P2POp(dist.isend, src, (self.rank + 1) % self.world),
                   dist.P2POp(dist.irecv, host, (self.rank - 1) % self.world)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        else:
            host = torch.empty(n * self.world, dtype=send.dtype)
            dist.all_gather_into_tensor(host, src)
        recv.copy_(host)
        if recv.is_cuda:
            torch.cuda.synchronize()
        return recv

    def left_block(self, recv):
        """The left neighbour's packed slots inside a receive buffer."""
        if self.mode == "ring":
            return recv
        n = recv.numel() // self.world
        lo = ((self.rank - 1) % self.world) * n
        return recv[lo:lo + n]

    def close(self):
        for c in self.comms:
            c.close()
        self.comms = []
