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
