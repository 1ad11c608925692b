"""Frame sharding across GPUs and the one collective of the path.

Extraction and stereo matching are independent per frame, so frames are dealt round-robin: global frame
g of a step lives on rank g % world, slot g // world.  Cross-frame matching (mono initialisation,
frame.cpp:289 + fmatcher.cpp:983) needs the predecessor frame's keypoints and descriptors, which sit on
the neighbouring rank: ONE all-gather of fixed-size packed result slots per step (RCCL over xGMI when the
tensors are on GPUs; gloo in the CPU tests).  No other collective exists on this path.
"""
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


def exchange_slots(local_packed, gathered, group=None):
    """All-gather every rank's packed slots.  local_packed: uint8 [batch*slot_bytes]; gathered: uint8
    [world*batch*slot_bytes] (rank-major).  World 1: a plain copy, no collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        gathered.copy_(local_packed)
        return gathered
    if dist.get_backend(group) == "gloo" and local_packed.is_cuda:
        # rehearsal path (tests / one-GPU dry runs): gloo moves host memory only
        host = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(host, local_packed.cpu(), group=group)
        gathered.copy_(host)
        return gathered
    dist.all_gather_into_tensor(gathered, local_packed, group=group)
    return gathered


def slot_view(gathered, rank, slot, batch, slot_bytes):
    off = (rank * batch + slot) * slot_bytes
    return gathered[off:off + slot_bytes]
