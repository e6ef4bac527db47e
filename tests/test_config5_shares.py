"""BASELINE config 5 beyond the first GPU's share, against the reference's bytes: "8xMI355X time-sharded, 25 Msps int16, 16
channels, 3600 s" is eight shares of 450 s; tests/golden/program_config35_static.npz holds the SHA-256 of every block of ALL
EIGHT (oracle/_ref/gps-sim-ref-25M -d 3600: 35 999 blocks of 2.5 * 10^6 samples, 360 GB, 5.4 CPU-hours of the reference;
tests/golden/make_golden.py --config35-only cfg5).  Tests (a)-(c) use the first two shares (8 999 blocks), (d) all of them.  A later share starts from the carrier state the reference's own run reaches at its start: in
GPSIQ_NCO_REFERENCE that is what the carrier chain gives (gpsiq_reference_chain: host only, serial per channel, ~1 us per
block and channel), after which the share is evaluated and rendered with no reference to the blocks before it.

  (a) both shares in one call: gpsiq_generate_batch_multi, two contexts, each rendering its 450 s == the digests, every block;
  (b) the second share ALONE, as a rank of a time-sharded run would render it: chain over the timeline, gpsiq_reference_seeded +
      gpsiq_set_descriptors / gpsiq_set_patches / gpsiq_launch for blocks 4 500 - 8 998 only == their digests;
  (c) the default fixed-point model: the second share, seeded with the exact carrier prefix, == the oracle on the blocks
      checked (it differs from the reference in a few elements per 10^7, tier T2 -- at 5 * 10^6 elements per block that is
      every block of this share);
  (d) the whole hour: chain once, then each of the eight shares alone (gpsiq_generate_seeded) == the digests, all 35 999 blocks.
The descriptors come from the library's own host chain (RINEX reader, allocation, nav words, batched refresh)."""
import hashlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import gpsiq
from _oracle import apply_patches
from _program import LLH, RINEX16, ROOT
from gpsiq.abi import NCO_REFERENCE, SC16
from test_config4 import start_time

GOLD = os.path.join(ROOT, "tests", "golden", "program_config35_static.npz")
FS, NS, NCHAN, NB = 25000000, 2500000, 16, 8999
BLK = NS * 4
CUT = 4500                                        # gpsiq_shard_range(8999, 0, 2) = [0, 4500)


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    sha = [str(s) for s in z["cfg5_sha16"]]
    if len(sha) < NB:
        pytest.skip("the 900 s capture of config 5 is not in the fixture (make_golden.py --config35-only cfg5)")
    return {"sha": sha, "heads": {int(b): h for b, h in zip(z["cfg5_head_blocks"], z["cfg5_heads"])}}


def host_chain(nblocks):
    """Config 5's first `nblocks` blocks as the library's host chain gives them: static BASELINE position, 16 satellites in the file."""
    from gpsiq.pipeline import RunAheadAllocating
    eph, utc, n = gpsiq.rinex_read(RINEX16, 2)
    week, sec = start_time(eph)
    lat, lon, h = (float(v) for v in LLH.split(","))
    xyz = np.tile(gpsiq.llh_to_ecef(lat / 57.2957795131, lon / 57.2957795131, h), (nblocks + 1, 1))
    return RunAheadAllocating(eph[:n], utc, NCHAN, week, sec, xyz[0], ieph=gpsiq.rinex_select(eph, n, week, sec)).descriptors(xyz[1:])


@pytest.fixture(scope="module")
def desc():
    return host_chain(NB)


def test_the_chain_and_the_second_share_on_the_host(gold, desc, oracle):
    """CPU half of (b): the carrier chain over all 8 999 blocks (gpsiq_reference_chain), then ONLY blocks of the second share
    evaluated from their start states (gpsiq_reference_seeded) -- the oracle's closed form + patches hashes to the reference
    program's digests of blocks 4 500, 4 501 and 8 998; the seeded rows equal gpsiq_reference_batch over the whole timeline."""
    starts, end, last = gpsiq.reference_chain(gpsiq.chain_inputs(desc), float(FS), NS)
    assert (desc["prn"][0] > 0).sum() >= 12 and np.array_equal(last, np.maximum(desc["prn"][-1], 0))     # 14 of the file's 16 satellites are up at the start
    for b0, b1 in ((CUT, CUT + 2), (NB - 1, NB)):
        q, patches = gpsiq.reference_seeded(desc[b0:b1], float(FS), NS, starts[b0:b1])
        for k in range(b1 - b0):
            o = oracle.block_fixed(q[k], NS, SC16, seq=True)
            apply_patches(oracle, q[k], o, patches[patches["block"] == k], SC16)
            assert hashlib.sha256(o.tobytes()).hexdigest() == gold["sha"][b0 + k], b0 + k
            if b0 + k in gold["heads"]:
                assert np.array_equal(o[:4096], gold["heads"][b0 + k])
    # the same rows from one call over a stretch that starts in the first share
    qa, pa, ce = gpsiq.reference_blocks(desc[CUT - 40:CUT + 3], float(FS), NS)
    # (that call is seeded from desc[CUT - 40].carr_phase, not from the chain: only the chain links it to the run's start)
    qb, pb = gpsiq.reference_seeded(desc[CUT - 40:CUT + 3], float(FS), NS, np.vstack([starts[CUT - 40:CUT - 39], starts[CUT - 39:CUT + 3]]))
    assert qb[0].tobytes() != qa[0].tobytes() or np.array_equal(starts[CUT - 40], desc["carr_phase"][CUT - 40])
    s2, _, _ = gpsiq.reference_chain(gpsiq.chain_inputs(desc[CUT - 40:]), float(FS), NS, carr_in=starts[CUT - 40], prn_in=desc["prn"][CUT - 41])
    assert np.array_equal(s2, starts[CUT - 40:]), "a chain continued from a published state goes on exactly"


def _digests(dev_tensor, nblocks):
    """SHA-256 of every block of a device buffer: pieces of 64 blocks through page-locked memory, hashed on eight threads."""
    import torch
    out = [None] * nblocks
    step = 64
    pin = [torch.empty(step * BLK, dtype=torch.uint8).pin_memory() for _ in range(2)]
    with ThreadPoolExecutor(8) as pool:
        pending = None
        for k, b0 in enumerate(range(0, nblocks, step)):
            nb = min(step, nblocks - b0)
            buf = pin[k & 1]
            buf[:nb * BLK].copy_(dev_tensor[b0 * BLK:(b0 + nb) * BLK], non_blocking=False)
            if pending is not None:
                for b, h in pending:
                    out[b] = h.result()
            host = buf.numpy()
            pending = [(b0 + i, pool.submit(lambda a: hashlib.sha256(a).hexdigest(), host[i * BLK:(i + 1) * BLK])) for i in range(nb)]
            if k & 1 == 0 and nb == step:
                continue                                  # the other buffer is free: copy the next piece while these hash
            for b, h in pending:
                out[b] = h.result()
            pending = None
        if pending is not None:
            for b, h in pending:
                out[b] = h.result()
    return out


@pytest.mark.gpu
def test_both_shares_in_one_multi_device_call(gold, desc):
    """(a) gpsiq_generate_batch_multi over the 8 999 blocks, two contexts in GPSIQ_NCO_REFERENCE, each rendering its own 450 s into
    its own 45 GB of device memory: every block == the reference program's digest, the carried phase == the chain's."""
    import torch
    ctxs = [gpsiq.Context(0), gpsiq.Context(0)]
    try:
        for c in ctxs:
            c.set_nco_mode(NCO_REFERENCE)
        bufs = [torch.empty(CUT * BLK, dtype=torch.uint8, device="cuda"), torch.empty((NB - CUT) * BLK, dtype=torch.uint8, device="cuda")]
        carr = np.zeros(NCHAN)
        gpsiq.generate_batch_multi(ctxs, desc, NS, float(FS), SC16, device_ptrs=[b.data_ptr() for b in bufs], carr_out=carr)
        torch.cuda.synchronize()
        sha = _digests(bufs[0], CUT) + _digests(bufs[1], NB - CUT)
    finally:
        for c in ctxs:
            c.close()
    bad = [b for b in range(NB) if sha[b] != gold["sha"][b]]
    assert not bad, f"{len(bad)} blocks differ from the reference program's output, first {bad[:10]}"
    _, end, _ = gpsiq.reference_chain(gpsiq.chain_inputs(desc), float(FS), NS)
    assert np.array_equal(carr, end)


@pytest.mark.gpu
def test_the_second_share_alone(gold, desc):
    """(b) blocks 4 500 - 8 998 rendered with no block of the first share evaluated or rendered: start states from the chain,
    gpsiq_reference_seeded, resident descriptors + patches, one launch == the reference program's digests of those blocks."""
    import torch
    starts, _, _ = gpsiq.reference_chain(gpsiq.chain_inputs(desc), float(FS), NS)
    q, patches = gpsiq.reference_seeded(desc[CUT:], float(FS), NS, starts[CUT:])
    ctx = gpsiq.Context(0)
    try:
        buf = torch.empty((NB - CUT) * BLK, dtype=torch.uint8, device="cuda")
        ctx.set_descriptors(q)
        ctx.set_patches(patches)
        ctx.launch(0, NB - CUT, NS, SC16, buf.data_ptr(), BLK)
        ctx.synchronize()
        sha = _digests(buf, NB - CUT)
        # the same as ONE call (what a C rank makes, host/gpsiq_shard.c): evaluated and rendered in pieces from the start states
        buf.zero_()
        ctx.generate_seeded(desc[CUT:CUT + 700], NS, float(FS), SC16, starts[CUT:CUT + 700], device_ptr=buf.data_ptr())
        assert _digests(buf, 700) == gold["sha"][CUT:CUT + 700]
    finally:
        ctx.close()
    bad = [CUT + b for b in range(NB - CUT) if sha[b] != gold["sha"][CUT + b]]
    assert len(patches) > 100 and not bad, f"{len(bad)} blocks differ from the reference program's output, first {bad[:10]}"


@pytest.mark.gpu
def test_the_second_share_in_the_fixed_point_model(gold, desc, oracle):
    """(c) the default model: the second share seeded with the exact carrier prefix of the first (gpsiq_quantize_batch over the
    timeline, its rows 4 500 -) == the oracle's closed form on the blocks checked; and it is the reference's bytes in all but
    a minority of blocks (tier T2)."""
    import torch
    q, _ = gpsiq.quantize_blocks(desc, float(FS), NS)
    ctx = gpsiq.Context(0)
    try:
        buf = torch.empty((NB - CUT) * BLK, dtype=torch.uint8, device="cuda")
        ctx.generate_quantized(q[CUT:], NS, SC16, device_ptr=buf.data_ptr())
        buf_head = {}
        for b in (CUT, CUT + 1, 6000, NB - 1):
            got = buf[(b - CUT) * BLK:(b - CUT + 1) * BLK].cpu().numpy().view(np.int16)
            assert np.array_equal(got, oracle.block_fixed(q[b], NS, SC16)), b
            buf_head[b] = got[:4096].copy()
        sha = _digests(buf, NB - CUT)
    finally:
        ctx.close()
    differing = sum(sha[b] != gold["sha"][CUT + b] for b in range(NB - CUT))
    print("config 5, second share, fixed-point NCO: %d of %d blocks hold an element that differs from the reference" % (differing, NB - CUT))
    # 450 s into the run every block of 5 * 10^6 elements holds one of the few-in-10^7 elements where the exact carrier carry and the
    # reference's rounded double have parted ways (measured: all 4 499; config 4's shorter blocks: 632 of 2 999) -- which is why the
    # reference's bytes are GPSIQ_NCO_REFERENCE's business; the captured heads (4 096 elements) still agree but for a handful
    assert differing > 0
    for b in (CUT, CUT + 1, NB - 1):
        if b in gold["heads"]:
            got = buf_head[b]
            assert (got != gold["heads"][b]).sum() <= 8, b


NB_FULL = 35999                                   # -d 3600: BASELINE config 5 in full


@pytest.mark.gpu
def test_all_eight_shares_of_the_whole_run_each_alone(gold):
    """BASELINE config 5 IN FULL: 3 600 s at 25 Msps int16 = 35 999 blocks, 360 GB, as eight time shares -- the carrier chain once
    over the whole timeline on the host (gpsiq_reference_chain), then every share rendered ALONE from its blocks' start states
    (gpsiq_generate_seeded: what rank r of an 8-GPU run does; here one GPU after the other into the same 45 GB of device
    memory).  Every one of the 35 999 blocks == the reference program's digest."""
    import torch
    if len(gold["sha"]) < NB_FULL:
        pytest.skip("the 3 600 s capture of config 5 is not in the fixture (make_golden.py --config35-only cfg5: about three hours of the reference)")
    d = host_chain(NB_FULL)
    starts, _, last = gpsiq.reference_chain(gpsiq.chain_inputs(d), float(FS), NS)
    assert np.array_equal(last, np.maximum(d["prn"][-1], 0))
    ctx = gpsiq.Context(0)
    bad = []
    try:
        buf = torch.empty(4500 * BLK, dtype=torch.uint8, device="cuda")
        for r in range(8):
            b0, b1 = gpsiq.shard_range(NB_FULL, r, 8)
            ctx.generate_seeded(d[b0:b1], NS, float(FS), SC16, starts[b0:b1], device_ptr=buf.data_ptr())
            sha = _digests(buf, b1 - b0)
            bad += [b0 + b for b in range(b1 - b0) if sha[b] != gold["sha"][b0 + b]]
    finally:
        ctx.close()
    assert not bad, f"{len(bad)} of {NB_FULL} blocks differ from the reference program's output, first {bad[:10]}"
