"""Frame sharding across GPUs and the one exchange step of the path.

Extraction and stereo matching are independent per frame, so frames are dealt round-robin: global frame
g of a step lives on rank g % world, slot g // world.  Cross-frame matching (mono initialisation,
frame.cpp:289 + fmatcher.cpp:983) needs the predecessor frame's keypoints and descriptors.  With this dealing
every predecessor lives on the LEFT neighbour (rank - 1 mod world: same slot, or for rank 0 the slot before),
so the exchange is a ring shift of the fixed-size packed result slots: every rank sends its slots to
rank + 1 and receives rank - 1's.  An all-gather (north_star's literal wording) moves world times as much for
nothing; it is kept as `mode="allgather"`.  No other collective exists on this path.

`SlotExchange` is the ONE object bench.py, the GPU tests and the gloo CPU tests drive:
  transport "rccl": the library's own communicator (include/vslam_fe.h vslam_comm_*): vslam_exchange_ring =
                    ncclSend/ncclRecv in one group, ENQUEUED on the extractor context's stream (no host sync);
  transport "gloo": host-staged isend/irecv pair (CPU tests here; one-GPU rehearsals of the N>1 path).
Which slot of which buffer holds a frame's predecessor (`predecessor`, `left_block`, `slot_view`) is the same
code for both transports.
"""
import numpy as np
import torch
import torch.distributed as dist


def global_frame(rank, slot, world):
    """Index, inside one step, of the frame held by (rank, slot)."""
    return slot * world + rank


def predecessor(rank, slot, world, batch):
    """(rank, slot, from_previous_step) of the frame preceding (rank, slot) in video order."""
    g = global_frame(rank, slot, world)
    if g == 0:
        return world - 1, batch - 1, True
    g -= 1
    return g % world, g // world, False


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


class SlotExchange:
    """The exchange step.  exchange(fe, send, recv): `send` = this rank's packed slots; afterwards left_block(recv)
    holds the left neighbour's.  mode "ring": recv has send's size; "allgather": world x that."""

    def __init__(self, rank, world, mode="ring", transport="gloo", comm=None):
        assert mode in ("ring", "allgather") and transport in ("rccl", "gloo", "local")
        self.rank, self.world, self.mode, self.transport, self.comm = rank, world, mode, transport, comm

    @classmethod
    def create(cls, rank, world, device, mode="ring", transport="rccl"):
        """Collective constructor (every rank calls it).  transport "rccl": rank 0 makes the ncclUniqueId, it travels
        through torch.distributed's broadcast, every rank builds the library's communicator, and ONE probe exchange
        decides -- by an all-reduce every rank takes part in -- whether the transport works; a failed probe is fatal
        (no collective is ever switched mid-run)."""
        if transport != "rccl":
            return cls(rank, world, mode, "gloo" if dist.is_initialized() else "local")
        import vi_slam_amd as V
        ctl = "cuda" if dist.get_backend() == "nccl" else "cpu"
        idt = torch.zeros(V.COMM_ID_BYTES, dtype=torch.uint8, device=ctl)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(V.Comm.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(idt, 0)
        ok, comm, err = 1.0, None, ""
        try:
            comm = V.Comm(device, rank, world, bytes(idt.cpu().numpy().tobytes()))
        except Exception as e:  # noqa: BLE001 - reported below, after every rank has voted
            ok, err = 0.0, str(e)
        x = cls(rank, world, mode, "rccl", comm)
        if ok:
            try:
                ok = 1.0 if x._probe(device) else 0.0
            except Exception as e:  # noqa: BLE001
                ok, err = 0.0, str(e)
        if world > 1:
            t = torch.tensor([ok], dtype=torch.float64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = float(t.item())
        if not ok:
            raise RuntimeError("rank %d: the RCCL exchange probe failed on at least one rank (%s)" % (rank, err or "another rank"))
        return x

    def _probe(self, device):
        """One exchange of a stamped buffer on a scratch stream context: the left neighbour's stamp must arrive."""
        import vi_slam_amd as V
        fe = V.FExtractor(100, 1.2, 2, 20, 7, 128, 128, device=device, max_batch=1)
        try:
            n = 4096
            send = torch.full((n,), self.rank + 1, dtype=torch.uint8, device="cuda")
            recv = torch.zeros(n * (self.world if self.mode == "allgather" else 1), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            self.exchange(fe, send, recv)
            torch.cuda.synchronize()  # the probe is the only place that waits for an exchange on the host
            want = (self.rank - 1) % self.world + 1
            return bool((self.left_block(recv) == want).all().item())
        finally:
            fe.close()

    # ------------------------------------------------------------------------------------------ the exchange
    def exchange(self, fe, send, recv):
        n = send.numel()
        if self.transport == "rccl":
            # enqueue-only, on fe's stream: ordered behind k_pack_slots, ahead of the matcher
            if self.mode == "ring":
                self.comm.ring(fe, send.data_ptr(), recv.data_ptr(), n)
            else:
                self.comm.allgather(fe, send.data_ptr(), recv.data_ptr(), n)
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
        if self.comm is not None:
            self.comm.close()
            self.comm = None
