"""Time-axis sharding of a block timeline over the GPUs of one node.

Each 0.1 s block depends only on its descriptor; the one piece of state the reference loop
hands from block to block is the carrier phase (gps.c:2821), and in the fixed-point model
it has the exact prefix  p_k = p_0 + sum_{j<k} nsamp*step_j (mod 2^59).  So rank r takes
the contiguous blocks [r*B/N, (r+1)*B/N) of the timeline, seeded from that prefix; the
data path needs no collective.  The host side shards too: a rank refreshes and quantises only
its own blocks and learns its carrier seed from 32 bytes per channel that every rank publishes
about its own range (one all-gather at set-up, gpsiq_shard_carry / gpsiq_shard_seed).
"""
import numpy as np

from . import quantize_blocks, shard_carry, shard_range, shard_seed  # noqa: F401
from .abi import SHARD_CARRY_DTYPE


def shard_descriptors(desc_all, fs, nsamp, rank, world):
    """Quantise the WHOLE timeline on this rank and return its slice (the simple recipe: host cost is
    microseconds per block, but every rank repeats it)."""
    q_all, _ = quantize_blocks(desc_all, fs, nsamp)
    b0, b1 = shard_range(len(desc_all), rank, world)
    return np.ascontiguousarray(q_all[b0:b1]), (b0, b1)


def quantize_own_shard(desc_own, fs, nsamp, rank, world, all_gather_bytes, history=None):
    """Quantise ONLY this rank's blocks (desc_own = its rows of the timeline) and seed its carrier from
    the ranks before it.  all_gather_bytes(bytes) -> [bytes of rank 0, ..., bytes of rank world-1] is the
    one exchange needed (32 bytes per channel and rank; torch.distributed, MPI, a pipe -- anything).
    The result equals shard_descriptors() of the whole timeline.  history: see seed_own_shard."""
    q, _ = quantize_blocks(desc_own, fs, nsamp)               # block 0 seeded from its own carr_phase
    return seed_own_shard(q, nsamp, rank, world, all_gather_bytes, history)


def seed_own_shard(q, nsamp, rank, world, all_gather_bytes, history=None):
    """The second half of quantize_own_shard for rows that are quantised already (self-seeded: quantize_blocks, or
    RunAhead.descriptors_quantized): exchange the 32-byte carries and add the exact prefix.  In place; returns q.

    history: a list the caller keeps between calls when the timeline goes on in rounds (round m gives rank
    r the blocks after those of rank r-1 of round m and after everything of round m-1): the carries of all
    earlier ranges, in timeline order; this call appends its round's.  The host side of round m+1 can then
    run while the GPUs are busy with round m."""
    from .abi import QCHAN_DTYPE
    if not (isinstance(q, np.ndarray) and q.dtype == QCHAN_DTYPE and q.flags.c_contiguous and q.flags.writeable):
        # "in place" is the contract (a preallocated upload buffer keeps being seeded): a copy would silently leave the
        # caller's rows self-seeded, i.e. a carrier phase step at every rank boundary
        raise ValueError("seed_own_shard needs a writable C-contiguous QCHAN_DTYPE array (it seeds in place)")
    mine = shard_carry(q, nsamp)
    parts = all_gather_bytes(mine.tobytes())
    assert len(parts) == world
    now = [np.frombuffer(p, dtype=SHARD_CARRY_DTYPE) for p in parts]
    before = [] if history is None else history
    seeded = shard_seed(q, nsamp, np.stack(before + now), len(before) + rank)
    if history is not None:
        history.extend(now)
    return seeded


def torch_all_gather_bytes(dist, device="cpu", group=None):
    """all_gather_bytes over a torch.distributed process group (gloo on CPU tensors, RCCL on GPU tensors).
    A run that prepares its next round while the GPUs are busy wants a gloo group here: an RCCL collective
    queues behind the kernels already launched on the device."""
    import torch

    def gather(b):
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return [b]
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(device)
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t, group=group)
        return [bytes(o.cpu().numpy().tobytes()) for o in outs]
    return gather


def max_over_ranks(seconds, dist=None, device=None):
    """bench.py's timing rule: the job takes as long as its slowest rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
