"""Time-axis sharding of a block timeline over the GPUs of one node.

Each 0.1 s block depends only on its descriptor; the one piece of state the reference loop
hands from block to block is the carrier phase (gps.c:2821), and in the fixed-point model
it has the exact prefix  p_k = p_0 + sum_{j<k} nsamp*step_j (mod 2^59).  So rank r takes
the contiguous blocks [r*B/N, (r+1)*B/N) of the timeline, seeded from that prefix; the
data path needs no collective (RCCL is only used for the bench's barrier / max-time).
"""
import numpy as np

from . import quantize_blocks, shard_range  # noqa: F401  (shard_range: the C-ABI's contiguous balanced split)


def shard_descriptors(desc_all, fs, nsamp, rank, world):
    """Quantise the whole timeline (host cost: microseconds per block) and return this
    rank's slice, so every shard starts from the exact carried carrier phase."""
    q_all, _ = quantize_blocks(desc_all, fs, nsamp)
    b0, b1 = shard_range(len(desc_all), rank, world)
    return np.ascontiguousarray(q_all[b0:b1]), (b0, b1)


def max_over_ranks(seconds, dist=None, device=None):
    """bench.py's timing rule: the job takes as long as its slowest rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
