"""Frame sharding across GPUs and the one exchange step of the path.

Extraction and stereo matching are independent per frame, so frames are dealt round-robin: global frame
g of a step lives on rank g % world, slot g // world.  Cross-frame matching (mono initialisation,
frame.cpp:289 + fmatcher.cpp:983) needs the predecessor frame's keypoints and descriptors.  With this dealing
every predecessor lives on the LEFT neighbour (rank - 1 mod world: same slot, or for rank 0 the slot before),
so the exchange is a ring shift of the fixed-size packed result slots: every rank sends its slots to
rank + 1 and receives rank - 1's -- ONE torch.distributed.all_to_all_single per step whose split lists
have a single non-empty entry (RCCL over xGMI executes it as one send/recv pair; gloo in the CPU tests).  An
all-gather would move world times as much for nothing.  No other collective exists on this path.
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


def shift_slots(local_packed, from_left, group=None):
    """Ring shift: send this rank's packed slots to rank+1, receive rank-1's into from_left (both uint8
    [batch*slot_bytes]).  World 1 without a process group: a plain copy (the only frame's predecessor is local)."""
    if not dist.is_initialized():
        from_left.copy_(local_packed)
        return from_left
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = local_packed.numel()
    ins, outs = [0] * world, [0] * world
    ins[(rank + 1) % world] = n   # everything goes to the right neighbour
    outs[(rank - 1) % world] = n  # everything comes from the left neighbour
    if dist.get_backend(group) == "gloo" and local_packed.is_cuda:
        # rehearsal path (tests / one-GPU dry runs): gloo moves host memory only
        host = torch.empty(n, dtype=from_left.dtype)
        dist.all_to_all_single(host, local_packed.cpu(), output_split_sizes=outs, input_split_sizes=ins, group=group)
        from_left.copy_(host)
        return from_left
    dist.all_to_all_single(from_left, local_packed, output_split_sizes=outs, input_split_sizes=ins, group=group)
    return from_left


def slot_view(buf, slot, slot_bytes):
    """Slot `slot` of a packed buffer (own slots or the left neighbour's)."""
    return buf[slot * slot_bytes:(slot + 1) * slot_bytes]
