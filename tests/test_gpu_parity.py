"""Parity tests proper: the HIP path through the C-ABI (libgpsiq.so) against the oracle.
Bit-exact: this is integer/byte work.  Run on the MI355X box with `-m gpu`."""
import hashlib

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import SC08, SC16
from gpsiq.scenario import synth_blocks
from test_golden import CASES, check_fixed_block_against_golden, load_case, start_state

pytestmark = pytest.mark.gpu

VARIANTS = ["generic", "rows", "rowsx", "tile", "seg", "segh", "segm", "segb"]


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU path in libgpsiq"
    c = gpsiq.Context(0)
    yield c
    c.close()


def run_device(ctx, q, nsamp, ss, variant, block0=0, nblocks=None):
    """Launch on resident descriptors into a torch buffer; returns [nblocks][2*nsamp]."""
    import torch
    nb = q.shape[0] if nblocks is None else nblocks
    blk = 2 * nsamp * ss
    stride = (blk + 15) & ~15
    buf = torch.full((nb * stride + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    ctx.launch(block0, nb, nsamp, ss, buf.data_ptr(), stride,
               stream=torch.cuda.current_stream().cuda_stream, variant=gpsiq.variants()[variant])
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    assert (host[nb * stride:] == 0x5A).all(), "kernel wrote past the ring"
    rows = host[: nb * stride].reshape(nb, stride)
    if stride != blk:
        assert (rows[:, blk:] == 0x5A).all(), "kernel wrote into the stride padding"
    return np.ascontiguousarray(rows[:, :blk]).view(np.int8 if ss == SC08 else np.int16)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("fs,nchan,ss,nb", [
    (2600000, 12, SC08, 3),      # BASELINE config 2
    (2600000, 16, SC08, 2),      # BASELINE metric workload
    (2600000, 16, SC16, 2),      # config 4 format
    (3000000, 8, SC08, 2),       # reference constants
    (10000000, 16, SC16, 1),     # config 3
    (25000000, 16, SC16, 1),     # config 5
])
def test_blocks_bit_exact_vs_oracle(ctx, oracle, variant, fs, nchan, ss, nb):
    d = synth_blocks(nb, nchan, seed=fs // 1000 + nchan + ss)
    ns = fs // 10
    q, _ = gpsiq.quantize_blocks(d, fs, ns)
    ctx.set_descriptors(q)
    got = run_device(ctx, q, ns, ss, variant)
    for b in range(nb):
        want = oracle.block_fixed(q[b], ns, ss, seq=True)
        bad = np.nonzero(got[b] != want)[0]
        assert bad.size == 0, f"block {b}: {bad.size} differing elements, first at {bad[:5]}"


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("nsamp", [1, 63, 64, 65, 1000, 16383, 16384, 16385, 40000])
def test_ragged_block_lengths(ctx, oracle, variant, nsamp):
    d = synth_blocks(2, 7, seed=nsamp)
    q, _ = gpsiq.quantize_blocks(d, 2.6e6, nsamp)
    ctx.set_descriptors(q)
    for ss in (SC08, SC16):
        got = run_device(ctx, q, nsamp, ss, variant)
        for b in range(2):
            assert np.array_equal(got[b], oracle.block_fixed(q[b], nsamp, ss))


@pytest.mark.parametrize("variant", VARIANTS)
def test_unused_slots_and_single_channel(ctx, oracle, variant):
    d = synth_blocks(2, 16, seed=11)
    d["prn"][:, [0, 3, 4, 15]] = 0
    d["prn"][1, 7] = 0
    ns = 30000
    q, _ = gpsiq.quantize_blocks(d, 3e6, ns)
    ctx.set_descriptors(q)
    got = run_device(ctx, q, ns, SC16, variant)
    for b in range(2):
        assert np.array_equal(got[b], oracle.block_fixed(q[b], ns, SC16))
    d1 = synth_blocks(1, 1, seed=12)
    q1, _ = gpsiq.quantize_blocks(d1, 3e6, ns)
    ctx.set_descriptors(q1)
    assert np.array_equal(run_device(ctx, q1, ns, SC08, variant)[0], oracle.block_fixed(q1[0], ns, SC08))
    dz = synth_blocks(1, 4, seed=13)
    dz["prn"][:] = 0                                  # nothing visible: silence
    qz, _ = gpsiq.quantize_blocks(dz, 3e6, 5000)
    ctx.set_descriptors(qz)
    assert not run_device(ctx, qz, 5000, SC16, variant).any()


@pytest.mark.parametrize("variant", VARIANTS)
def test_boundary_phases(ctx, oracle, variant):
    """Phases that sit exactly on chip / LUT / period / nav-bit / word boundaries, zero and
    extreme Doppler, binary-fraction rates (every boundary is hit exactly)."""
    ns = 70000
    d = synth_blocks(1, 16, seed=21)[0]
    d["code_phase"] = [0.0, 1.0, 1022.0, 1022.999999, 511.5, 0.25, 1022.5, 33.0, 0.0, 1000.0, 7.75, 1.5, 2.0, 3.0, 4.0, 5.0]
    d["carr_phase"] = [0.0, 0.5, 1.0 - 2.0 ** -53, 1.0 / 512, 255.0 / 512, 0.75, 0.25, 2.0 ** -40, 0.0, 0.999, 0.1, 0.2, 0.3, 0.4, 0.6, 0.7]
    fs = 4092000.0                                     # f_code/fs = 0.25 exactly when f_carr = 0
    d["f_carr"] = [0.0, 0.0, fs / 512, -fs / 512, fs / 1024, -fs / 4096, 5000.0, -5000.0, 12345.678, -9876.5, 0.001, -0.001, 2500.0, -2500.0, 100.0, -100.0]
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    d["f_code"][:2] = 1.023e6
    d["icode"] = [19, 0, 19, 19, 10, 19, 19, 0, 19, 19, 5, 6, 7, 8, 9, 18]
    d["ibit"] = [29, 0, 29, 28, 15, 29, 29, 0, 29, 29, 1, 2, 3, 4, 5, 6]
    d["iword"] = np.arange(16) * 3
    for ss in (SC08, SC16):
        q, _ = gpsiq.quantize(d, fs, ns)
        ctx.set_descriptors(q[None, :])
        got = run_device(ctx, q[None, :], ns, ss, variant)[0]
        assert np.array_equal(got, oracle.block_fixed(q, ns, ss))


@pytest.mark.parametrize("variant", VARIANTS)
def test_int8_wraps_like_the_reference(ctx, oracle, variant):
    """|sum| > 2047 must wrap modulo 256 after >>4, not saturate (gps.c:2845)."""
    d = synth_blocks(1, 16, seed=31)
    d["gain"] = 1.9
    ns = 50000
    q, _ = gpsiq.quantize_blocks(d, 2.6e6, ns)
    ctx.set_descriptors(q)
    g8 = run_device(ctx, q, ns, SC08, variant)[0]
    g16 = run_device(ctx, q, ns, SC16, variant)[0]
    assert (np.abs(g16.astype(np.int32)) > 2047).any(), "test input does not overflow int8"
    assert np.array_equal(g8, (g16 >> 4).astype(np.int8))
    assert np.array_equal(g8, oracle.block_fixed(q[0], ns, SC08))
    d["gain"] = 300.0                                  # sums far beyond int16 too: (short) wraps, gps.c:2834
    q, _ = gpsiq.quantize_blocks(d, 2.6e6, ns)
    ctx.set_descriptors(q)
    assert np.array_equal(run_device(ctx, q, ns, SC16, variant)[0], oracle.block_fixed(q[0], ns, SC16))


@pytest.mark.parametrize("fs,expect", [(2.048e6, "segh"), (2.0e6, "segh"), (1.5e6, "segh"), (1.1e6, "segh"),
                                       (1.0e6, "generic"), (0.6e6, "generic")])
def test_low_sample_rates(ctx, oracle, fs, expect):
    """f_code/fs > 31/63 chip per sample (fs < 2.08 Msps): a 64-sample row no longer fits one
    32-chip window; auto takes the half-row-window kernel down to one chip per sample
    (1.023 Msps) and the generic kernel below that.  Asking for a kernel the rate does not
    allow is an error, not wrong output."""
    d = synth_blocks(2, 11, seed=41)
    ns = int(fs) // 10
    q, _ = gpsiq.quantize_blocks(d, fs, ns)
    ctx.set_descriptors(q)
    want = [oracle.block_fixed(q[b], ns, SC16) for b in range(2)]
    got = run_device(ctx, q, ns, SC16, "auto")
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    got = run_device(ctx, q, ns, SC08, expect)
    assert np.array_equal(got[1], oracle.block_fixed(q[1], ns, SC08))
    for v in ("rows", "seg") + (("segh",) if expect == "generic" else ()):
        with pytest.raises(gpsiq.GpsiqError) as e:
            run_device(ctx, q, ns, SC16, v)
        assert e.value.code == -2
    # auto == the expected kernel, bit for bit and by name
    assert np.array_equal(run_device(ctx, q, ns, SC16, expect), run_device(ctx, q, ns, SC16, "auto"))


def test_half_row_kernel_fuzz(ctx, oracle):
    """Random quantised descriptors with code steps up to the half-row kernel's limit
    (one chip per sample), long enough to cross chunk boundaries."""
    from gpsiq.abi import QCHAN_DTYPE
    rng = np.random.default_rng(4242)
    max_step = ((31 << 56) - 1) // 31
    for case in range(6):
        nb, nc = 2, int(rng.integers(1, 17))
        ns = int(rng.choice([1, 31, 33, 2047, 2049, 16384, 16385, 70001]))
        q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
        q["prn"] = rng.integers(0, 33, size=(nb, nc))
        q["carr_phase"] = rng.integers(0, 1 << 59, size=(nb, nc), dtype=np.uint64)
        q["carr_step"] = rng.integers(-(1 << 58) + 1, 1 << 58, size=(nb, nc))
        q["code_frac"] = rng.integers(0, 1 << 56, size=(nb, nc), dtype=np.uint64)
        q["code_step"] = rng.integers(max_step // 2, max_step + 1, size=(nb, nc), dtype=np.uint64)
        q["code_step"][0, :] = max_step
        q["chip0"] = rng.integers(0, 1023, size=(nb, nc))
        q["chip0"][:, ::3] = 1022
        q["icode"] = rng.integers(0, 20, size=(nb, nc))
        q["nav_bits"] = rng.integers(0, 1 << 32, size=(nb, nc), dtype=np.uint64).astype(np.uint32)
        q["gain"] = rng.choice([0.0, -0.7, 1.0, 0.3333, 2.0, 17.25], size=(nb, nc))
        ctx.set_descriptors(q)
        for ss in (SC08, SC16):
            got = run_device(ctx, q, ns, ss, "segh")
            for b in range(nb):
                assert np.array_equal(got[b], oracle.block_fixed(q[b], ns, ss)), (case, b, ss, ns, nc)


@pytest.mark.parametrize("name", CASES)
def test_golden_blocks_on_gpu(ctx, name):
    """The committed captures of the reference's own loop, reproduced by the HIP path in the default
    (fixed-point) model given each block's start state as captured: tier T1.  The t1diff_* captures are
    blocks on which the two models are known to differ: there the GPU must differ from the reference
    in exactly the recorded elements (test_gpu_reference_nco.py reproduces them in GPSIQ_NCO_REFERENCE)."""
    g = load_case(name)
    for b in range(len(g["sha"])):
        q, _ = gpsiq.quantize(start_state(g, b), g["fs"], g["nsamp"])
        ctx.set_descriptors(q[None, :])
        got = run_device(ctx, q[None, :], g["nsamp"], g["ss"], "auto")[0]
        check_fixed_block_against_golden(g, b, got)
        if not (g["t1_block"] == b).any() or (g["t1_elem"][g["t1_block"] == b] >= g["head"].shape[1]).all():
            assert np.array_equal(got[: g["head"].shape[1]], g["head"][b])


def test_drop_in_block_calls_carry_the_carrier(ctx, oracle):
    """gpsiq_generate_block called once per 0.1 s block, handing carr_phase back each time
    (the drop-in pattern of INTEGRATION.md) == gpsiq_generate_batch == oracle with exact carry."""
    fs, ns, nb, nc = 2.6e6, 26000, 4, 9
    d = synth_blocks(nb, nc, seed=51)
    d["prn"][2:, 4] = 0                                # slot 4 goes out of view
    d["prn"][3, 6] = 30                                # slot 6 re-allocated to a new SV
    d["carr_phase"][3, 6] = 0.8125
    qo = oracle.quantize_blocks(d, fs, ns)
    want = np.stack([oracle.block_fixed(qo[b], ns, SC16) for b in range(nb)])
    batch = ctx.generate_batch(d, ns, fs, SC16)
    assert np.array_equal(batch, want)
    carr = None
    for b in range(nb):
        db = d[b].copy()
        if carr is not None:
            keep = d[b]["prn"] == d[b - 1]["prn"]
            db["carr_phase"] = np.where(keep, carr, db["carr_phase"])
        out, carr = ctx.generate_block(db, ns, fs, SC16)
        assert np.array_equal(out, want[b]), b


def test_batches_chain_like_one_batch(ctx, oracle):
    """Two gpsiq_generate_batch calls, the second fed the carr_phase the first handed out
    (the 30 s epoch loop of INTEGRATION.md section 3), equal one call over all blocks; a batch also
    continues a gpsiq_generate_block call and vice versa."""
    fs, ns, nb, nc = 2.6e6, 26000, 6, 7
    d = synth_blocks(nb, nc, seed=52)
    qo = oracle.quantize_blocks(d, fs, ns)
    want = np.stack([oracle.block_fixed(qo[b], ns, SC08) for b in range(nb)])
    carr = np.zeros(nc)
    first = ctx.generate_batch(d[:2], ns, fs, SC08, carr_out=carr)
    d2 = d[2:5].copy()
    d2["carr_phase"][0] = carr
    second = ctx.generate_batch(d2, ns, fs, SC08, carr_out=carr)
    d3 = d[5].copy()
    d3["carr_phase"] = carr
    third, _ = ctx.generate_block(d3, ns, fs, SC08)
    assert np.array_equal(np.concatenate([first, second, third[None, :]]), want)
    # not handing the phase back re-seeds from the double: a (tiny) phase jump, different samples
    again = ctx.generate_batch(d[2:5], ns, fs, SC08)
    assert not np.array_equal(again, want[2:5])


@pytest.mark.parametrize("world", [2, 8])
def test_quantised_shards_equal_the_batch_call(ctx, oracle, world):
    """The C-ABI's multi-GPU recipe on one device: gpsiq_quantize_batch over the whole timeline,
    gpsiq_shard_range per rank, gpsiq_generate_quantized per shard == one gpsiq_generate_batch
    over everything == the oracle, including a slot that is re-allocated mid-run."""
    fs, ns, nb, nc = 2.6e6, 52000, 21, 9
    d = synth_blocks(nb, nc, seed=88)
    d["prn"][6:9, 4] = 0
    d["prn"][9:, 4] = 31
    q, _ = gpsiq.quantize_blocks(d, fs, ns)
    assert q.tobytes() == oracle.quantize_blocks(d, fs, ns).tobytes()
    shards = []
    for r in range(world):
        b0, b1 = gpsiq.shard_range(nb, r, world)
        shards.append(ctx.generate_quantized(q[b0:b1], ns, SC16))
    got = np.concatenate(shards)
    assert got.shape == (nb, 2 * ns)
    whole = gpsiq.Context().generate_batch(d, ns, fs, SC16)       # fresh context: nothing handed over
    assert np.array_equal(got, whole)
    for b in (0, 8, 9, 20):
        assert np.array_equal(got[b], oracle.block_fixed(q[b], ns, SC16))
    assert ctx.generate_quantized(q[:0], ns, SC16).shape == (0, 2 * ns)


def test_time_sharding_is_seamless(ctx, oracle):
    """Any sub-range of blocks launched on its own equals the same blocks of one big launch
    (the property the multi-GPU time sharding rests on), and int8 == int16>>4 throughout."""
    fs, ns, nb = 2.6e6, 260000, 12
    d = synth_blocks(nb, 16, seed=61)
    q, _ = gpsiq.quantize_blocks(d, fs, ns)
    ctx.set_descriptors(q)
    whole = run_device(ctx, q, ns, SC16, "auto")
    parts = np.concatenate([run_device(ctx, q, ns, SC16, "auto", block0=b0, nblocks=n)
                            for b0, n in ((0, 5), (5, 1), (6, 6))])
    assert np.array_equal(whole, parts)
    w8 = run_device(ctx, q, ns, SC08, "auto")
    assert np.array_equal(w8, (whole >> 4).astype(np.int8))
    for b in (0, 7, 11):
        n0 = 1000 * b + 17
        assert np.array_equal(whole[b][2 * n0: 2 * (n0 + 4096)], oracle.block_fixed_range(q[b], n0, 4096, SC16))


def test_baseline_size_properties(ctx, oracle):
    """BASELINE config 3 size (10 Msps, int16, 16 ch): 96 blocks = 384 MB per launch; checks
    that do not need the oracle over the whole output: both kernels agree byte for byte,
    int8 == int16>>4, and random windows match the oracle."""
    import torch
    fs, ns, nb = 10e6, 1000000, 96
    pattern = synth_blocks(8, 16, seed=71)
    d = np.concatenate([pattern] * (nb // 8))
    q, _ = gpsiq.quantize_blocks(d, fs, ns)
    ctx.set_descriptors(q)
    stride = 4 * ns
    a = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    b_ = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ctx.launch(0, nb, ns, SC16, a.data_ptr(), stride, stream=s, variant=gpsiq.variants()["rowsx"])
    ctx.launch(0, nb, ns, SC16, b_.data_ptr(), stride, stream=s, variant=gpsiq.variants()["generic"])
    torch.cuda.synchronize()
    assert torch.equal(a, b_)
    ctx.launch(0, nb, ns, SC16, b_.data_ptr(), stride, stream=s, variant=gpsiq.variants()["rows"])
    torch.cuda.synchronize()
    assert torch.equal(a, b_)
    c = torch.empty(nb * 2 * ns, dtype=torch.uint8, device="cuda")
    ctx.launch(0, nb, ns, SC08, c.data_ptr(), 2 * ns, stream=s, variant=gpsiq.variants()["rowsx"])
    torch.cuda.synchronize()
    assert torch.equal((a.view(torch.int16) >> 4).to(torch.int8), c.view(torch.int8))
    rng = np.random.default_rng(5)
    a16 = a.view(torch.int16)
    for _ in range(12):
        blk, n0 = int(rng.integers(nb)), int(rng.integers(ns - 2048))
        got = a16[blk * 2 * ns + 2 * n0: blk * 2 * ns + 2 * (n0 + 2048)].cpu().numpy()
        assert np.array_equal(got, oracle.block_fixed_range(q[blk], n0, 2048, SC16))


@pytest.mark.parametrize("variant", VARIANTS)
def test_quantised_descriptor_fuzz(ctx, oracle, variant):
    """Random QUANTISED descriptors straight into the kernels: full-range carrier steps
    (|step| up to 0.5 cycle/sample), code steps up to the row kernel's limit, arbitrary
    fractions, nav bits, chips next to the period end, negative / zero / large gains."""
    from gpsiq.abi import QCHAN_DTYPE
    rng = np.random.default_rng(2025)
    max_step = ((31 << 56) - 1) // 63
    for case in range(12):
        nb, nc = int(rng.integers(1, 4)), int(rng.integers(1, 17))
        ns = int(rng.integers(1, 5000))
        q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
        q["prn"] = rng.integers(0, 33, size=(nb, nc))              # 0 = unused slot
        q["carr_phase"] = rng.integers(0, 1 << 59, size=(nb, nc), dtype=np.uint64)
        q["carr_step"] = rng.integers(-(1 << 58) + 1, 1 << 58, size=(nb, nc))
        q["code_frac"] = rng.integers(0, 1 << 56, size=(nb, nc), dtype=np.uint64)
        q["code_step"] = rng.integers(1, max_step + 1, size=(nb, nc), dtype=np.uint64)
        q["code_step"][0, :] = max_step                             # the limit itself
        q["chip0"] = rng.integers(0, 1023, size=(nb, nc))
        q["chip0"][:, ::3] = 1022
        q["icode"] = rng.integers(0, 20, size=(nb, nc))
        q["nav_bits"] = rng.integers(0, 1 << 32, size=(nb, nc), dtype=np.uint64).astype(np.uint32)
        q["gain"] = rng.choice([0.0, -0.7, 1.0, 0.3333, 2.0, -300.0, 1e-3, 17.25], size=(nb, nc))
        ctx.set_descriptors(q)
        for ss in (SC08, SC16):
            got = run_device(ctx, q, ns, ss, variant)
            for b in range(nb):
                assert np.array_equal(got[b], oracle.block_fixed(q[b], ns, ss)), (case, b, ss)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_quantised_descriptor_fuzz_long_blocks(ctx, oracle, seed):
    """The same kind of random quantised descriptors over blocks long enough to cross tile
    and chunk boundaries of every row kernel (32768-sample tiles, up to 4 chunks per wave),
    with block lengths that leave partial rows and partial tiles, and a strided destination."""
    from gpsiq.abi import QCHAN_DTYPE
    rng = np.random.default_rng(7700 + seed)
    max_step = ((31 << 56) - 1) // 63
    nb, nc = 3, int(rng.choice([3, 4, 7, 8, 11, 12, 16]))
    ns = int(rng.choice([32768, 32769, 65535, 131072 + 64, 200001, 262144 + 17]))
    q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
    q["prn"] = rng.integers(0, 33, size=(nb, nc))
    q["carr_phase"] = rng.integers(0, 1 << 59, size=(nb, nc), dtype=np.uint64)
    q["carr_step"] = rng.integers(-(1 << 58) + 1, 1 << 58, size=(nb, nc))
    q["code_frac"] = rng.integers(0, 1 << 56, size=(nb, nc), dtype=np.uint64)
    # keep the block inside the 32 nav bits a descriptor carries (icode and chip0 add up to one more bit)
    room = (30 * 20 * 1023) << 56
    q["code_step"] = rng.integers(1, min(max_step, room // ns) + 1, size=(nb, nc), dtype=np.uint64)
    q["chip0"] = rng.integers(0, 1023, size=(nb, nc))
    q["icode"] = rng.integers(0, 20, size=(nb, nc))
    q["nav_bits"] = rng.integers(0, 1 << 32, size=(nb, nc), dtype=np.uint64).astype(np.uint32)
    q["gain"] = rng.choice([0.0, -0.7, 1.0, 0.3333, 2.0, 1e-3], size=(nb, nc))
    ctx.set_descriptors(q)
    want = {ss: [oracle.block_fixed(q[b], ns, ss) for b in range(nb)] for ss in (SC08, SC16)}
    for variant in VARIANTS:
        for ss in (SC08, SC16):
            got = run_device(ctx, q, ns, ss, variant)
            for b in range(nb):
                assert np.array_equal(got[b], want[ss][b]), (variant, ss, b, ns, nc)


@pytest.mark.parametrize("variant", ["tile", "seg", "segh"])
@pytest.mark.parametrize("gain,expect_wrap", [(8.19, False), (8.3, True), (-8.19, False), (131.0, True)])
def test_amplitude_bound_and_the_plain_add_kernels(ctx, oracle, variant, gain, expect_wrap):
    """The tile kernels sum the channels with plain 32-bit adds (entry = I + 65536*Q, chip sign as
    half a carrier cycle) whenever no block's sum of (int)(250*|gain|) exceeds 32767, and with packed
    16-bit multiply-adds otherwise.  16 identical channels make every sample the worst case:
    16 * 2047 = 32752 stays inside int16 (plain adds), 16 * 2075 = 33200 wraps like the reference's
    (short) cast (packed path); both equal the oracle in both formats."""
    fs, ns = 2.6e6, 40000
    d = synth_blocks(2, 16, seed=123)
    for f in ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "carr_phase", "code_phase", "dwrd"):
        d[f][:, 1:] = d[f][:, :1]
    d["gain"] = gain
    q, _ = gpsiq.quantize_blocks(d, fs, ns)
    ctx.set_descriptors(q)
    want16 = [oracle.block_fixed(q[b], ns, SC16) for b in range(2)]
    peak = max(int(np.abs(w.astype(np.int32)).max()) for w in want16)
    assert (peak > 32752) == expect_wrap or expect_wrap            # the wrapped cases reach the int16 limits
    got16 = run_device(ctx, q, ns, SC16, variant)
    got8 = run_device(ctx, q, ns, SC08, variant)
    for b in range(2):
        assert np.array_equal(got16[b], want16[b])
        assert np.array_equal(got8[b], oracle.block_fixed(q[b], ns, SC08))
    if not expect_wrap:
        assert peak == 16 * 2047                                   # the largest sum the plain-add kernels may see


def test_mixed_gains_next_to_the_amplitude_bound(ctx, oracle):
    """One block just under the bound and one just over in the same launch: the whole launch takes
    the packed path; relaunching only the small block takes the plain-add path; same bytes."""
    fs, ns = 2.6e6, 30000
    d = synth_blocks(2, 16, seed=321)
    d["gain"][0] = 32767.0 / 250.0 / 16.0 - 1e-3           # 16 * 2047 = 32752 <= 32767
    d["gain"][1] = 32767.0 / 250.0 / 16.0 + 0.2            # 16 * 2097 > 32767
    q, _ = gpsiq.quantize_blocks(d, fs, ns)
    ctx.set_descriptors(q)
    both = run_device(ctx, q, ns, SC16, "auto")
    ctx.set_descriptors(q[:1])
    small = run_device(ctx, q[:1], ns, SC16, "auto")
    assert np.array_equal(both[0], small[0])
    for b in range(2):
        assert np.array_equal(both[b], oracle.block_fixed(q[b], ns, SC16))


def test_empty_and_maximum_block_sizes(ctx, oracle):
    """Empty launches are no-ops; the longest block the descriptor format allows (32 nav bits
    = 0.62 s of signal) is exact from the first to the last sample; one sample more is an error."""
    import torch
    d = synth_blocks(1, 16, seed=91)
    d["icode"], d["ibit"], d["iword"] = 0, 0, 0
    d["code_phase"] = d["code_phase"] % 500.0
    fs = 13.0e6
    ns_max = int(0.6 * fs)                               # 7.8 M samples, 30 nav bits
    q, _ = gpsiq.quantize_blocks(d, fs, ns_max)
    ctx.set_descriptors(q)
    ctx.launch(0, 0, ns_max, SC16, 0, 4 * ns_max)        # zero blocks: nothing to do, no error
    ctx.launch(0, 1, 0, SC16, 0, 0)                      # zero samples
    a = torch.empty(4 * ns_max, dtype=torch.uint8, device="cuda")
    b = torch.empty(4 * ns_max, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ctx.launch(0, 1, ns_max, SC16, a.data_ptr(), 4 * ns_max, stream=s, variant=gpsiq.variants()["seg"])
    ctx.launch(0, 1, ns_max, SC16, b.data_ptr(), 4 * ns_max, stream=s, variant=gpsiq.variants()["generic"])
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    a16 = a.view(torch.int16)
    for n0 in (0, 1234567, ns_max - 4096):
        got = a16[2 * n0: 2 * (n0 + 4096)].cpu().numpy()
        assert np.array_equal(got, oracle.block_fixed_range(q[0], n0, 4096, SC16))
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.quantize_blocks(d, fs, int(0.7 * fs))      # would need more than 32 nav bits
    out = ctx.generate_batch(d[:0], 100, fs, SC16)       # empty batch
    assert out.shape == (0, 200)


def test_bad_arguments_are_errors(ctx):
    import torch
    d = synth_blocks(1, 2, seed=81)
    q, _ = gpsiq.quantize_blocks(d, 2.6e6, 100)
    ctx.set_descriptors(q)
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(gpsiq.GpsiqError):
        ctx.launch(0, 5, 100, SC16, buf.data_ptr(), 400)      # only 1 block resident
    with pytest.raises(gpsiq.GpsiqError):
        ctx.launch(0, 1, 100, SC16, buf.data_ptr(), 396)      # stride smaller than a block
    with pytest.raises(gpsiq.GpsiqError):
        ctx.launch(0, 1, 100, SC16, 0, 400)                   # null destination
    with pytest.raises(gpsiq.GpsiqError):
        ctx.launch(0, 1, 100, SC16, buf.data_ptr(), 400, variant=99)
    bad = d.copy()
    bad["prn"][0, 0] = 40
    with pytest.raises(gpsiq.GpsiqError):
        ctx.generate_batch(bad, 100, 2.6e6, SC16)
    bad = d.copy()
    bad["code_phase"][0, 0] = 1023.0
    with pytest.raises(gpsiq.GpsiqError):
        ctx.generate_batch(bad, 100, 2.6e6, SC16)
    with pytest.raises(gpsiq.GpsiqError):
        ctx.generate_batch(d, 100, 2.6e6, 3)


def test_asynchronous_block_calls(ctx, oracle):
    """gpsiq_generate_block_async: blocks queued back to back into separate page-locked buffers (more than the ring of
    four holds), the carrier handed back in at once, interleaved with a synchronous call: after gpsiq_wait every
    buffer holds what the synchronous calls produce, i.e. the oracle's blocks with the exact carry."""
    import torch
    fs, ns, nb, nc = 2.6e6, 52000, 11, 9
    d = synth_blocks(nb, nc, seed=83)
    d["prn"][5:, 4] = 0
    d["prn"][8:, 6] = 30
    d["carr_phase"][8:, 6] = 0.8125
    qo = oracle.quantize_blocks(d, fs, ns)
    for ss in (SC08, SC16):
        want = np.stack([oracle.block_fixed(qo[b], ns, ss) for b in range(nb)])
        bufs = [torch.zeros(2 * ns * ss, dtype=torch.uint8).pin_memory() for _ in range(nb)]
        carr = None
        for b in range(nb):
            db = d[b].copy()
            if carr is not None:
                keep = d[b]["prn"] == d[b - 1]["prn"]
                db["carr_phase"] = np.where(keep, carr, db["carr_phase"])
            if b == 6:                                   # a synchronous call in between queues behind the asynchronous ones
                out, carr = ctx.generate_block(db, ns, fs, ss)
                bufs[b].numpy().view(out.dtype)[:] = out
            else:
                carr = ctx.generate_block_async(db, ns, fs, ss, bufs[b].data_ptr())
        ctx.wait()
        for b in range(nb):
            got = bufs[b].numpy().view(np.int8 if ss == SC08 else np.int16)
            assert np.array_equal(got, want[b]), (ss, b)


def test_descriptor_sets_swap_under_running_launches(ctx, oracle):
    """A stream of rounds without a synchronisation in between: every round uploads a new descriptor set and
    launches on the caller's stream while the launches of the rounds before are still running (the sets
    alternate between two device buffers; the third set waits for the first one's launch only).  Every round's
    output equals what the same set gives on its own, and the oracle."""
    import torch
    fs, ns, nb, nc, rounds = 2.6e6, 260000, 96, 16, 7
    blk = 2 * ns
    sets = [gpsiq.quantize_blocks(synth_blocks(nb, nc, seed=300 + m), fs, ns)[0] for m in range(rounds)]
    out = torch.zeros(rounds * nb * blk, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for m in range(rounds):
            ctx.set_descriptors(sets[m])
            # two launches per set, so that a set is still in use when the next but one arrives
            ctx.launch(0, nb // 2, ns, SC08, out.data_ptr() + m * nb * blk, blk, stream=side.cuda_stream)
            ctx.launch(nb // 2, nb - nb // 2, ns, SC08, out.data_ptr() + (m * nb + nb // 2) * blk, blk, stream=side.cuda_stream)
    side.synchronize()
    got = out.cpu().numpy().view(np.int8).reshape(rounds, nb, blk)
    for m in range(rounds):
        ctx.set_descriptors(sets[m])
        alone = run_device(ctx, sets[m], ns, SC08, "auto")
        assert np.array_equal(got[m], alone), m
        for b in (0, nb // 2 - 1, nb // 2, nb - 1):
            assert np.array_equal(got[m, b], oracle.block_fixed(sets[m][b], ns, SC08)), (m, b)


def test_launches_of_one_set_on_two_streams_are_both_waited_for(ctx, oracle):
    """a long launch on stream A, then a short one on stream B, then two new descriptor sets.  The second
    gpsiq_set_descriptors reuses the buffer the first set lives in: it has to wait for A's launch too, not only for
    the one recorded last (B's)."""
    import torch
    fs, ns, nb, nc = 2.6e6, 260000, 1536, 16
    blk = 2 * ns
    first = gpsiq.quantize_blocks(synth_blocks(nb, nc, seed=41), fs, ns)[0]
    other = [gpsiq.quantize_blocks(synth_blocks(4, nc, seed=42 + k), fs, ns)[0] for k in range(2)]
    out = torch.zeros(nb * blk, dtype=torch.uint8, device="cuda")
    small = torch.zeros(2 * blk, dtype=torch.uint8, device="cuda")
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    ctx.set_descriptors(first)
    ctx.launch(0, nb, ns, SC08, out.data_ptr(), blk, stream=a.cuda_stream)            # ~1 ms of kernel
    ctx.launch(0, 2, ns, SC08, small.data_ptr(), blk, stream=b.cuda_stream)           # a few microseconds
    ctx.set_descriptors(other[0])          # the other buffer: no wait
    ctx.set_descriptors(other[1])          # first's buffer: must not be overwritten under stream A's kernel
    a.synchronize(); b.synchronize()
    got = out.cpu().numpy().view(np.int8).reshape(nb, blk)
    for blkno in (0, 1, nb // 2, nb - 2, nb - 1):
        assert np.array_equal(got[blkno], oracle.block_fixed(first[blkno], ns, SC08)), blkno
    ctx.set_descriptors(first)
    alone = run_device(ctx, first, ns, SC08, "auto")
    assert np.array_equal(got, alone)


@pytest.mark.parametrize("piece", ["2", "5", "0"])
def test_device_batch_quantised_and_rendered_in_pieces(ctx, oracle, monkeypatch, piece):
    """A long gpsiq_generate_batch into device memory quantises, validates and uploads piece k+1 under the kernel of piece k:
    whatever the piece length (here forced to 2 / 5 blocks / one piece) the blocks are the oracle's with the exact carrier
    prefix -- slots going out of view and coming back re-seed inside and across pieces -- and the phase handed out chains
    the next call."""
    import torch
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", piece)
    fs, ns, nb, nc = 2.6e6, 26000, 13, 9
    d = synth_blocks(nb, nc, seed=58)
    d["prn"][3:6, 2] = 0                       # out of view over a piece edge, back with its own phase
    d["carr_phase"][6:, 2] = 0.5625
    d["prn"][4:, 5] = 31                       # re-allocated at a piece edge (pieces of 2) / inside a piece (pieces of 5)
    d["carr_phase"][4:, 5] = 0.1875
    qo = oracle.quantize_blocks(d, fs, ns)
    for ss in (SC08, SC16):
        want = np.stack([oracle.block_fixed(qo[b], ns, ss) for b in range(nb)])
        buf = torch.zeros(nb * 2 * ns * ss, dtype=torch.uint8, device="cuda")
        carr = np.zeros(nc)
        ctx.generate_batch(d[:10], ns, fs, ss, device_ptr=buf.data_ptr(), carr_out=carr)
        d2 = d[10:].copy()
        d2["carr_phase"][0] = carr
        ctx.generate_batch(d2, ns, fs, ss, device_ptr=buf.data_ptr() + 10 * 2 * ns * ss)
        torch.cuda.synchronize()
        got = buf.cpu().numpy().view(np.int8 if ss == SC08 else np.int16).reshape(nb, 2 * ns)
        assert np.array_equal(got, want), ss


@pytest.mark.parametrize("mode", [0, 1])
def test_an_error_in_a_late_piece_of_a_batch(ctx, oracle, monkeypatch, mode):
    """A batch that is worked through in pieces (both NCO models) and has a descriptor outside the NCO format in a LATE
    piece: the call reports it (the earlier pieces are already on the device, everything queued is drained before the
    call returns), and the context goes on working."""
    import torch
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", "3")
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", "3")
    fs, ns, nb, nc = 2.6e6, 26000, 11, 6
    d = synth_blocks(nb, nc, seed=91)
    bad = d.copy()
    bad["f_code"][8, 2] = 0.0
    buf = torch.zeros(nb * 2 * ns, dtype=torch.uint8, device="cuda")
    ctx.set_nco_mode(mode)
    try:
        with pytest.raises(gpsiq.GpsiqError) as ei:
            ctx.generate_batch(bad, ns, fs, SC08, device_ptr=buf.data_ptr())
        assert "block" in str(ei.value)
        torch.cuda.synchronize()
        if mode == 0:
            got = ctx.generate_batch(d, ns, fs, SC08)
            qo = oracle.quantize_blocks(d, fs, ns)
            for b in (0, 5, 10):
                assert np.array_equal(got[b], oracle.block_fixed(qo[b], ns, SC08)), b
        else:
            got = ctx.generate_batch(d[:4], ns, fs, SC08)
            o, _ = oracle.block_float(d[0], ns, fs, SC08)
            assert np.array_equal(got[0], o)
    finally:
        ctx.set_nco_mode(0)
