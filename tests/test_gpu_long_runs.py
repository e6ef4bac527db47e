"""Full-length BASELINE configs on the GPU: config 3 (10 Msps int16, 16 ch, 300 s = 2 999 blocks) and one
GPU's share of config 5 (25 Msps int16, 16 ch, 3600 s / 8 = 4 500 blocks) through gpsiq_generate_batch into a
device ring that is reused, spot-checked against the oracle at random places incl. the far end of the run
(the carrier prefix after thousands of blocks, the far end of 2.5 M-sample blocks); and the one-call
multi-device entry point."""
import numpy as np
import pytest

import gpsiq
from gpsiq.abi import NCO_REFERENCE, SC08, SC16
from gpsiq.scenario import synth_blocks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU path in libgpsiq"
    c = gpsiq.Context(0)
    yield c
    c.close()


def exact_prefix(oracle, d, fs, ns):
    """carrier phase (59-bit) at the start of every block + after the last one, from the oracle's quantiser
    and Python integers: p_{k+1} = p_k + nsamp*step_k (mod 2^59), re-seeded where a slot changes PRN"""
    nb, nc = d.shape
    start = np.zeros((nb + 1, nc), dtype=object)
    cur = [None] * nc
    prev = [0] * nc
    for b in range(nb):
        q, _ = oracle.quantize(d[b], fs, ns)
        for c in range(nc):
            if q[c]["prn"] == 0:
                cur[c], prev[c] = None, 0
                continue
            if cur[c] is None or prev[c] != int(q[c]["prn"]):
                cur[c] = int(q[c]["carr_phase"])
            start[b, c] = cur[c]
            cur[c] = (cur[c] + int(q[c]["carr_step"]) * ns) % (1 << 59)
            prev[c] = int(q[c]["prn"])
    for c in range(nc):
        start[nb, c] = cur[c] if cur[c] is not None else 0
    return start


@pytest.mark.parametrize("fs,nb,per_call,seed", [(10000000, 2999, 500, 31),      # config 3, full length
                                                  (25000000, 4500, 200, 32)])     # config 5, one GPU's 450 s
def test_full_length_runs(ctx, oracle, fs, nb, per_call, seed):
    import torch
    ns, ss, nc = fs // 10, SC16, 16
    d = synth_blocks(nb, nc, seed=seed)
    start = exact_prefix(oracle, d, float(fs), ns)
    ring = torch.empty(per_call * 4 * ns, dtype=torch.uint8, device="cuda")
    rng = np.random.default_rng(seed)
    carr = None
    for b0 in range(0, nb, per_call):
        b1 = min(nb, b0 + per_call)
        part = d[b0:b1].copy()
        if carr is not None:
            part["carr_phase"][0] = carr                      # what the previous call handed out
        carr = np.zeros(nc)
        ctx.generate_batch(part, ns, float(fs), ss, device_ptr=ring.data_ptr(), carr_out=carr)
        torch.cuda.synchronize()
        view = ring[: (b1 - b0) * 4 * ns].view(torch.int16).view(b1 - b0, 2 * ns)
        checks = [(int(rng.integers(b0, b1)), int(rng.integers(0, ns - 4096))) for _ in range(16)]
        checks += [(b0, 0), (b1 - 1, ns - 4096)]
        whole = [b for b in (0, nb - 1) if b0 <= b < b1]
        for b, n0 in checks:
            q, _ = oracle.quantize(d[b], float(fs), ns, carry_in=np.array([int(x) for x in start[b]], dtype=np.uint64))
            want = oracle.block_fixed_range(q, n0, 4096, ss)
            got = view[b - b0, 2 * n0: 2 * (n0 + 4096)].cpu().numpy()
            assert np.array_equal(got, want), (b, n0)
        for b in whole:                                       # first and last block of the run, every element
            q, _ = oracle.quantize(d[b], float(fs), ns, carry_in=np.array([int(x) for x in start[b]], dtype=np.uint64))
            assert np.array_equal(view[b - b0].cpu().numpy(), oracle.block_fixed(q, ns, ss, seq=True)), b
        # the phase handed on is the exact prefix, to the last bit
        want_carr = np.array([float(int(start[b1, c])) / 2.0 ** 59 for c in range(nc)])
        assert np.array_equal(carr, want_carr), b1


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_multi_device_entry_point_on_one_gpu(ctx, oracle, ndev):
    """gpsiq_generate_batch_multi with ndev contexts (all on device 0 here: the driver's box has one GPU) ==
    gpsiq_generate_batch: host destination, per-context device destinations, both NCO models, chained calls."""
    import torch
    fs, ns, nb, nc = 2.6e6, 26000, 11, 9
    d = synth_blocks(nb, nc, seed=71)
    d["prn"][4:, 2] = 0
    d["prn"][7:, 2] = 27
    d["carr_phase"][7:, 2] = 0.4375
    ctxs = [gpsiq.Context(0) for _ in range(ndev)]
    try:
        for ss in (SC08, SC16):
            carr_one, carr_multi = np.zeros(nc), np.zeros(nc)
            want = ctx.generate_batch(d, ns, fs, ss, carr_out=carr_one)
            got = gpsiq.generate_batch_multi(ctxs, d, ns, fs, ss, carr_out=carr_multi)
            assert np.array_equal(got, want) and np.array_equal(carr_one, carr_multi)
            # a second call continues the first when the phase is handed back in
            d2 = d.copy()
            d2["carr_phase"][0] = carr_one
            want2 = ctx.generate_batch(d2, ns, fs, ss)
            d2["carr_phase"][0] = carr_multi
            assert np.array_equal(gpsiq.generate_batch_multi(ctxs, d2, ns, fs, ss), want2)
            # device destinations: range i packed into its own buffer
            bufs, ranges = [], [gpsiq.shard_range(nb, i, ndev) for i in range(ndev)]
            for b0, b1 in ranges:
                bufs.append(torch.zeros(max(1, (b1 - b0) * 2 * ns * ss), dtype=torch.uint8, device="cuda"))
            gpsiq.generate_batch_multi(ctxs, d, ns, fs, ss, device_ptrs=[b.data_ptr() for b in bufs])
            torch.cuda.synchronize()
            for (b0, b1), buf in zip(ranges, bufs):
                if b1 > b0:
                    part = buf[: (b1 - b0) * 2 * ns * ss].cpu().numpy().view(np.int8 if ss == SC08 else np.int16).reshape(b1 - b0, 2 * ns)
                    assert np.array_equal(part, want[b0:b1])
        if ndev > 1:                                                  # one context per range, no sharing
            with pytest.raises(gpsiq.GpsiqError):
                gpsiq.generate_batch_multi([ctxs[0], ctxs[0]], d, ns, fs, SC08)
        # GPSIQ_NCO_REFERENCE through the same entry point (the mode is taken from the first context)
        one = gpsiq.Context(0)
        one.set_nco_mode(NCO_REFERENCE)
        ctxs[0].set_nco_mode(NCO_REFERENCE)
        d25 = synth_blocks(2, 16, seed=3032)                      # 5 patches
        want = one.generate_batch(d25, 2500000, 25e6, SC16)
        assert np.array_equal(gpsiq.generate_batch_multi(ctxs, d25, 2500000, 25e6, SC16), want)
        one.close()
    finally:
        for c in ctxs:
            c.close()
