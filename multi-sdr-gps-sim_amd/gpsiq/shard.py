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


def gather_ragged(all_gather_bytes, payload):
    """all_gather_bytes for payloads whose length differs from rank to rank (the first nblocks % world ranks own one
    block more; the channels of the chain do not divide evenly either): lengths first, then the payloads padded."""
    lens = [int.from_bytes(b, "little") for b in all_gather_bytes(len(payload).to_bytes(8, "little"))]
    width = max(lens)
    parts = all_gather_bytes(payload + bytes(width - len(payload)))
    return [p[:n] for p, n in zip(parts, lens)]


def reference_chain_by_time(cin_own, fs, nsamp, rank, world, all_gather_bytes, ctx=None, max_stretches=0):
    """The carrier chain of GPSIQ_NCO_REFERENCE sharded by TIME: this rank's rows of gpsiq_reference_chain over the whole
    timeline -- (carr_start_own[nblocks_own][nchan], carr_end[nchan], last_prn[nchan] after the LAST block of the timeline) --
    from the chain inputs (gpsiq_chain_inputs) of its own blocks only.  The expensive part, the certified map of every block
    (gpsiq_chain_maps; on the GPU when ctx is given), needs nothing but an ESTIMATE of where the range starts:
        1. every rank summarises its range (exact phase advance), all-gather, fold the ranks before: the phase its range starts at;
        2. the same again with the modelled rounding drift, which wants that absolute phase: two exchanges of 56 bytes per slot;
        3. level 1 of its own blocks, all ranks at once;
        4. the true states: every rank composes the maps of its range into ONE map per slot (gpsiq_chain_range: 160 bytes),
           all-gather, and every rank folds the ranges before its own (gpsiq_chain_range_fold) -- the accumulator its range is
           entered with; it links its own blocks from there, and one more all-gather of the end states (12 bytes per slot)
           confirms that every range ended where the fold said.  Four exchanges whatever the world size.  Where a range cannot
           be composed (a block whose map does not apply: its true walk needs the true entry state) the ranks relay one after
           the other as before: rank r links its range from the accumulator rank r - 1 ended on."""
    from . import chain_fold, chain_link, chain_maps, chain_range, chain_range_fold, chain_summary
    from .abi import CHAIN_EST_DTYPE, CHAIN_IN_DTYPE, CHAIN_RANGE_DTYPE
    cin_own = np.ascontiguousarray(cin_own, dtype=CHAIN_IN_DTYPE)
    nchan = cin_own.shape[1]

    def gathered(est):
        return np.stack([np.frombuffer(b, dtype=CHAIN_EST_DTYPE) for b in all_gather_bytes(est.tobytes())])

    phase = gathered(chain_summary(cin_own, fs, nsamp))
    drift = gathered(chain_summary(cin_own, fs, nsamp, start=chain_fold(phase[:rank])))
    maps = chain_maps(cin_own, fs, nsamp, start=chain_fold(drift[:rank]), max_stretches=max_stretches, ctx=ctx)[0]
    if world > 2:
        ranges = np.stack([np.frombuffer(b, dtype=CHAIN_RANGE_DTYPE) for b in all_gather_bytes(chain_range(cin_own, maps, fs, nsamp).tobytes())])
        true_end, true_prn, true_known = np.zeros((world, nchan)), np.zeros((world, nchan), dtype=np.int32), np.zeros(world, dtype=np.uint8)
        mine = None
        for _ in range(world + 1):                            # one round when every range composes; one more per range in a row that does not
            _, carr_at, prn_at, known = chain_range_fold(ranges, true_end, true_prn, true_known)
            if mine is None and known[rank].all():
                mine = chain_link(cin_own, maps, fs, nsamp, carr_at[rank] if rank else None, prn_at[rank] if rank else None)
            blob = bytes(1 + 12 * nchan) if mine is None else b"\x01" + mine[1].tobytes() + mine[2].astype(np.int32).tobytes()
            for r, got in enumerate(all_gather_bytes(blob)):
                if got[0]:
                    true_known[r] = 1
                    true_end[r] = np.frombuffer(got[1:1 + 8 * nchan], dtype=np.float64)
                    true_prn[r] = np.frombuffer(got[1 + 8 * nchan:], dtype=np.int32)
            if true_known.all():
                _, carr_at, prn_at, _ = chain_range_fold(ranges, true_end, true_prn, true_known)
                return mine[0], carr_at[world].copy(), prn_at[world].copy()
        raise RuntimeError("reference_chain_by_time: the ranks did not all link their ranges")
    start, carr, prn = None, None, None
    for r in range(world):
        blob = bytes(12 * nchan)
        if r == rank:
            start, end, last = chain_link(cin_own, maps, fs, nsamp, carr, prn)
            blob = end.tobytes() + last.tobytes()
        got = all_gather_bytes(blob)[r]
        carr = np.frombuffer(got[:8 * nchan], dtype=np.float64).copy()
        prn = np.frombuffer(got[8 * nchan:], dtype=np.int32).copy()
    return start, carr, prn


def reference_own_shard(desc_own, fs, nsamp, rank, world, all_gather_bytes, by_time=True, ctx=None):
    """GPSIQ_NCO_REFERENCE time-sharded over processes: this rank's rows of gpsiq_reference_batch over the WHOLE timeline,
    from its own blocks' descriptors -> (q_own, patches_own with block indices counted from its first block, carr_end[nchan]
    and last_prn[nchan] after the last block of the whole timeline; carr_end of a slot whose last_prn is 0 is 0.0: the slot is
    unused there, a caller that continues the timeline takes the next descriptor's own phase).
    by_time (default): the chain is sharded by time too (reference_chain_by_time above: every rank walks only its own blocks,
    on its GPU when ctx is given), then every rank evaluates its own blocks from their start states (gpsiq_reference_seeded).
    by_time False, the recipe of round 4: the chain sharded by CHANNEL -- all-gather the chain inputs of everybody's blocks
    (24 bytes per channel and block), gpsiq_reference_chain over this rank's channels of the whole timeline, all-gather the
    start states (8 bytes per channel and block).  Host-side exchanges at set-up either way; nothing on the data path."""
    from . import chain_inputs, reference_chain, reference_seeded, shard_range
    from .abi import CHAIN_IN_DTYPE
    desc_own = np.ascontiguousarray(desc_own)
    nchan = desc_own.shape[1]
    if by_time:
        start_own, carr_end, last_prn = reference_chain_by_time(chain_inputs(desc_own), fs, nsamp, rank, world, all_gather_bytes, ctx=ctx)
        q, patches = reference_seeded(desc_own, fs, nsamp, start_own)
        return q, patches, carr_end, last_prn
    parts = gather_ragged(all_gather_bytes, chain_inputs(desc_own).tobytes())
    cin = np.concatenate([np.frombuffer(p, dtype=CHAIN_IN_DTYPE).reshape(-1, nchan) for p in parts])       # timeline order = rank order
    first = sum(len(p) // (CHAIN_IN_DTYPE.itemsize * nchan) for p in parts[:rank])
    c0, c1 = shard_range(nchan, rank, world)
    if c1 > c0:
        st, end, last = reference_chain(np.ascontiguousarray(cin[:, c0:c1]), fs, nsamp)
    else:
        st, end, last = np.zeros((len(cin), 0)), np.zeros(0), np.zeros(0, dtype=np.int32)
    cols = gather_ragged(all_gather_bytes, st.tobytes() + end.tobytes() + last.astype(np.int32).tobytes())
    start = np.zeros((len(cin), nchan))
    carr_end = np.zeros(nchan)
    last_prn = np.zeros(nchan, dtype=np.int32)
    for r, blob in enumerate(cols):
        a0, a1 = shard_range(nchan, r, world)
        w = a1 - a0
        if w == 0:
            continue
        n_st = len(cin) * w * 8
        start[:, a0:a1] = np.frombuffer(blob[:n_st], dtype=np.float64).reshape(len(cin), w)
        carr_end[a0:a1] = np.frombuffer(blob[n_st:n_st + 8 * w], dtype=np.float64)
        last_prn[a0:a1] = np.frombuffer(blob[n_st + 8 * w:n_st + 12 * w], dtype=np.int32)
    q, patches = reference_seeded(desc_own, fs, nsamp, start[first:first + len(desc_own)])
    return q, patches, carr_end, last_prn


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
