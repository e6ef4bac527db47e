"""The N>1 path on CPU: two gloo ranks shard one timeline; the shards, generated
independently from the exact carrier prefix, must concatenate to exactly what one process
generates for the whole timeline (checked with the oracle standing in for the GPU)."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FS, NSAMP, NBLOCKS, NCHAN = 2.6e6, 20000, 7, 6


def _timeline():
    from gpsiq.scenario import synth_blocks
    d = synth_blocks(NBLOCKS, NCHAN, seed=17)
    d["prn"][3:, 1] = 0          # a satellite sets ...
    d["prn"][5:, 1] = 29         # ... and the slot is re-allocated (carrier re-seeded)
    d["carr_phase"][5:, 1] = 0.3125
    return d


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import _oracle
    from gpsiq.abi import SC16
    import numpy as np
    from gpsiq.shard import max_over_ranks, quantize_own_shard, shard_descriptors, shard_range, torch_all_gather_bytes
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    q, (b0, b1) = shard_descriptors(_timeline(), FS, NSAMP, rank, world)
    assert (b0, b1) == shard_range(NBLOCKS, rank, world) and len(q) == b1 - b0
    # the host side sharded as well: this rank sees ONLY its own rows and gets its carrier seed from the
    # 32 bytes per channel every rank publishes (one all-gather); same descriptors as the whole-timeline recipe
    own = quantize_own_shard(_timeline()[b0:b1], FS, NSAMP, rank, world, torch_all_gather_bytes(dist))
    assert np.array_equal(own, q)
    orc = _oracle.load_oracle()
    digests = [hashlib.sha256(orc.block_fixed(q[i], NSAMP, SC16, seq=True).tobytes()).hexdigest() for i in range(len(q))]
    dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, (b0, digests))
    t = max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        out.put((gathered, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_time_sharding_is_seamless(oracle):
    from gpsiq.abi import SC16
    from gpsiq.shard import shard_range
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, t = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0                                       # max over ranks
    q_all = oracle.quantize_blocks(_timeline(), FS, NSAMP)
    want = [hashlib.sha256(oracle.block_fixed(q_all[b], NSAMP, SC16, seq=True).tobytes()).hexdigest() for b in range(NBLOCKS)]
    got = {}
    for b0, digs in gathered:
        for i, d in enumerate(digs):
            got[b0 + i] = d
    assert [got[b] for b in range(NBLOCKS)] == want
    assert shard_range(7, 0, 2) == (0, 4) and shard_range(7, 1, 2) == (4, 7)
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


# ---- GPSIQ_NCO_REFERENCE over processes ---------------------------------------------------------------------------
RFS, RNS, RNB, RNCH = 10.0e6, 200000, 11, 7


def _ref_timeline():
    """Phases a hair off LUT / chip boundaries and whole numbers of samples per LUT step and chip on some channels (plenty of
    candidates and patches), a slot going out of view and one re-allocated."""
    from gpsiq.scenario import synth_blocks
    rng = np.random.default_rng(5)
    d = synth_blocks(RNB, RNCH, seed=23)
    d["carr_phase"][:] = (rng.integers(0, 512, (RNB, RNCH)) + 1e-12) / 512.0
    d["code_phase"][:] = rng.integers(0, 1023, (RNB, RNCH)) + 1e-9
    for c in (0, 3):
        d["f_carr"][:, c] = RFS / 512.0 / (5 + c)
        d["f_code"][:, c] = RFS / 25.0
        d["carr_phase"][:, c] = rng.integers(0, 512, RNB) / 512.0 + 2.0 ** -12 - 2.0 ** -50
        d["code_phase"][:, c] = rng.integers(1, 7, RNB) - 2.0 ** -41
    d["prn"][4:7, 2] = 0
    d["prn"][8:, 5] = 29
    return d


def _ref_worker(rank, world, port, out, by_time):
    sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
    import torch.distributed as dist
    import gpsiq
    from gpsiq.shard import reference_own_shard, shard_range, torch_all_gather_bytes
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    b0, b1 = shard_range(RNB, rank, world)
    # this rank sees ONLY its own rows of the timeline
    q, patches, carr_end, last_prn = reference_own_shard(_ref_timeline()[b0:b1], RFS, RNS, rank, world, torch_all_gather_bytes(dist), by_time=by_time)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b0, b1, q.tobytes(), patches.tobytes(), carr_end.tobytes(), last_prn.tobytes()))
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,by_time", [(2, True), (2, False), (3, True), (3, False), (8, True)])
def test_reference_nco_shards_equal_the_whole_timeline(world, by_time):
    """Two (three) gloo ranks, each holding only its own blocks' descriptors.  by_time: the carrier chain sharded by TIME
    (gpsiq/shard.py::reference_chain_by_time: every rank walks the certified maps of its own blocks, the true states are
    relayed rank to rank); else by channel (gpsiq_reference_chain over each rank's channels of the whole timeline).
    Everything else by time (gpsiq_reference_seeded over each rank's blocks) -- descriptors, patches and the carried phase
    equal gpsiq_reference_batch over the whole timeline in one process."""
    import gpsiq
    from gpsiq.abi import PATCH_DTYPE, QCHAN_DTYPE
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_ref_worker, args=(r, world, port, out, by_time)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = _ref_timeline()
    q, patches, carr = gpsiq.reference_blocks(d, RFS, RNS)
    assert len(patches) > 20
    seen = 0
    for b0, b1, qb, pb, cb, lb in gathered:
        assert np.frombuffer(qb, dtype=QCHAN_DTYPE).reshape(-1, RNCH).tobytes() == q[b0:b1].tobytes(), (b0, b1)
        want = patches[(patches["block"] >= b0) & (patches["block"] < b1)].copy()
        want["block"] -= b0
        assert np.frombuffer(pb, dtype=PATCH_DTYPE).tobytes() == want.tobytes(), (b0, b1)
        seen += len(want)
        act = d[-1]["prn"] > 0
        assert np.array_equal(np.frombuffer(cb)[act], carr[act])              # every rank knows the state after the whole timeline
        assert np.array_equal(np.frombuffer(lb, dtype=np.int32), np.where(act, d[-1]["prn"], 0))
    assert seen == len(patches)


# ---- the carrier chain alone, sharded by time ---------------------------------------------------------------------------
CFS, CNS, CNB, CNCH = 2.6e6, 260000, 90, 9


def _chain_timeline():
    from test_chain_parallel import timeline
    return timeline(41, CNB, CNCH)


def _chain_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import gpsiq
    from gpsiq.shard import reference_chain_by_time, torch_all_gather_bytes
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # uneven ranges, one rank without blocks (world 3 and 8), a one-block range (world 8)
    cuts = {2: [0, 37, CNB], 3: [0, 31, 31, CNB], 8: [0, 5, 19, 19, 20, 44, 61, 78, CNB]}[world]
    before = gpsiq.chain_stats()
    start, end, prn = reference_chain_by_time(_chain_timeline()[cuts[rank]:cuts[rank + 1]], CFS, CNS, rank, world, torch_all_gather_bytes(dist))
    linked, walked = (a - b for a, b in zip(gpsiq.chain_stats(), before))
    gathered = [None] * world
    dist.all_gather_object(gathered, (cuts[rank], start.tobytes(), end.tobytes(), prn.tobytes(), linked, walked))
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_time_sharded_chain_equals_the_serial_chain_over_the_whole_timeline(world):
    """gpsiq_reference_chain over 90 blocks x 9 slots in one process == the ranks' own rows of the chain sharded by time,
    bit for bit, and every rank ends up with the state after the whole timeline; most blocks go through their maps.
    World 8 (one node's worth of ranks): uneven ranges, a rank without blocks, a rank with one block."""
    import gpsiq
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_chain_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_start, want_end, want_prn = gpsiq.reference_chain(_chain_timeline(), CFS, CNS)
    rows = b"".join(g[1] for g in sorted(gathered, key=lambda g: g[0]))           # (a rank without blocks contributes b"")
    assert rows == want_start.tobytes()
    for g in gathered:
        assert g[2] == want_end.tobytes() and g[3] == want_prn.tobytes()
    linked, walked = sum(g[4] for g in gathered), sum(g[5] for g in gathered)
    assert linked > (4 if world < 8 else 1) * walked, (linked, walked)       # (world > 2 links twice: once to compose the range, once for the states)
