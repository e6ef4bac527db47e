"""The carrier chain of GPSIQ_NCO_REFERENCE, parallel in time (csrc/gpsiq_lane.h, gpsiq_chain.cpp, gpsiq_chain_kernels.hip):
every block walked on its own from a representative start state (level 1: certified maps), the chain itself one exact
subtraction / range check / addition per block (level 2).  The result must be the serial chain's (gpsiq_reference_chain =
NcoWalk, pinned to the plain loop of gps.c:2821-2826 by test_reference_nco_host.py and the soaks), bit for bit.

CPU tests run the host twin of the lane code; the GPU tests run the kernels (same source) through the C-ABI."""
import os
import subprocess

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import CHAIN_IN_DTYPE, CHAIN_EST_DTYPE, CHAIN_EXACT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timeline(seed, nblocks, nchan, fmax=6000.0, modes=True):
    """Chain inputs of a synthetic timeline: Doppler ramps, some through zero, slots re-allocated and unused."""
    rng = np.random.default_rng(seed)
    cin = np.zeros((nblocks, nchan), dtype=CHAIN_IN_DTYPE)
    b = np.arange(nblocks)
    for i in range(nchan):
        mode = int(rng.integers(0, 6)) if modes else 0
        f0, df = rng.uniform(-fmax, fmax), rng.uniform(-0.9, 0.9)
        if mode == 1:
            f0, df = rng.uniform(-30, 30), rng.uniform(-3, 3)            # through zero Doppler: slow blocks, a sign change
        cin["f_carr"][:, i] = f0 + df * b + rng.uniform(-0.02, 0.02, nblocks)
        cin["carr_phase"][:, i] = rng.uniform(0, 1, nblocks)
        prn = np.full(nblocks, 1 + int(rng.integers(0, 32)), dtype=np.int32)
        if mode == 2:
            cut = int(rng.integers(1, nblocks))
            prn[cut:] = 1 + (prn[0] % 32)                                 # the slot gets another satellite
        if mode == 3:
            prn[nblocks // 3: nblocks // 2] = 0                           # unused for a while
        cin["prn"][:, i] = prn
    return cin


def assert_same_chain(got, want):
    for g, w, what in zip(got, want, ("carr_start", "carr_end", "last_prn")):
        assert g.tobytes() == w.tobytes(), what


@pytest.mark.parametrize("fs,nsamp", [(2.6e6, 260000), (10e6, 1000000), (25e6, 2500000), (2.6e6, 33333)])
@pytest.mark.parametrize("stretches", [1, 5, 32])
def test_maps_and_link_equal_the_serial_chain(fs, nsamp, stretches):
    cin = timeline(int(fs) % 997 + stretches, 40, 6)
    want = gpsiq.reference_chain(cin, fs, nsamp)
    maps, _ = gpsiq.chain_maps(cin, fs, nsamp, max_stretches=stretches)
    assert_same_chain(gpsiq.chain_link(cin, maps, fs, nsamp), want)
    ok = maps["ok"][cin["prn"] > 0]
    assert ok.mean() > 0.6                      # the maps do the work (slow and irregular blocks are walked)


def test_fast_channels_are_linked_not_walked():
    """Doppler of +-100 Hz .. 6 kHz, no special cases: all but the odd block (an exact tie on a wrap, an addend the fast walk
    does not take and the block after it) go through their map, and the estimate of a start state is good to ~1e-11 cycle."""
    fs, nsamp = 2.6e6, 260000
    cin = timeline(5, 120, 8, modes=False)
    cin["f_carr"] += np.where(np.abs(cin["f_carr"]) < 100.0, 200.0, 0.0)
    want = gpsiq.reference_chain(cin, fs, nsamp)
    before = gpsiq.chain_stats()
    maps, _ = gpsiq.chain_maps(cin, fs, nsamp, max_stretches=8)
    assert_same_chain(gpsiq.chain_link(cin, maps, fs, nsamp), want)
    linked, walked = (a - b for a, b in zip(gpsiq.chain_stats(), before))
    assert linked + walked == cin.size and walked <= 0.03 * cin.size, (linked, walked)
    good = maps["ok"] != 0
    d = (want[0][good] - maps["xs"][good]) * 2.0 ** 53
    assert np.all(d == np.rint(d)) and np.abs(d).max() * 2.0 ** -53 < 1e-9


def test_a_continued_timeline_and_ranges_of_ranks():
    """Two calls, the second handed the accumulator the first left; and three ranges each summarised on its own, folded, and
    linked one after the other -- what the ranks of a time-sharded run do (gpsiq/shard.py::reference_chain_by_time)."""
    fs, nsamp = 3.0e6, 300000
    cin = timeline(11, 50, 7)
    want = gpsiq.reference_chain(cin, fs, nsamp)
    cut = 19
    m0, _ = gpsiq.chain_maps(cin[:cut], fs, nsamp)
    s0, e0, p0 = gpsiq.chain_link(cin[:cut], m0, fs, nsamp)
    st = np.zeros(cin.shape[1], dtype=CHAIN_EST_DTYPE)
    st["carr"], st["prn"], st["flags"], st["f_carr"] = e0, p0, CHAIN_EXACT, cin["f_carr"][cut - 1]
    m1, _ = gpsiq.chain_maps(cin[cut:], fs, nsamp, start=st)
    s1, e1, p1 = gpsiq.chain_link(cin[cut:], m1, fs, nsamp, e0, p0)
    assert_same_chain((np.concatenate([s0, s1]), e1, p1), want)

    cuts = [0, 13, 13, 31, 50]                                   # one rank without blocks
    ranges = [cin[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    sums = np.stack([gpsiq.chain_summary(r, fs, nsamp) for r in ranges])
    sums2 = np.stack([gpsiq.chain_summary(r, fs, nsamp, start=gpsiq.chain_fold(sums[:k])) for k, r in enumerate(ranges)])
    starts, carr, prn = [], None, None
    for k, r in enumerate(ranges):
        maps, _ = gpsiq.chain_maps(r, fs, nsamp, start=gpsiq.chain_fold(sums2[:k]), max_stretches=4)
        s, carr, prn = gpsiq.chain_link(r, maps, fs, nsamp, carr, prn)
        starts.append(s)
    assert_same_chain((np.concatenate(starts), carr, prn), want)


@pytest.mark.parametrize("modes", [False, True])
def test_ranges_composed_and_folded_equal_the_serial_chain(modes):
    """The relay of a time-sharded chain (gpsiq_chain_range / gpsiq_chain_range_fold, what gpsiq/shard.py::reference_chain_by_time
    does over an all-gather, here in one process): eight ranges of a timeline, each with its maps composed into one map per slot;
    folding them gives the state every range is entered with; a rank whose row is known links its blocks and publishes the state
    it ended on; repeat.  Every start state == the serial chain's, bit for bit; ordinary timelines need one or two rounds,
    not eight."""
    fs, ns, nb, nc = 2.6e6, 260000, 90, 9
    rounds_total = 0
    for seed in range(12):
        cin = timeline(100 + seed, nb, nc, modes=modes)
        if not modes:
            cin["f_carr"] += np.where(np.abs(cin["f_carr"]) < 100.0, 200.0, 0.0)
        want_start, want_end, want_prn = gpsiq.reference_chain(cin, fs, ns)
        cuts = [0] + sorted(np.random.default_rng(seed).integers(0, nb + 1, 7).tolist()) + [nb]        # uneven, some empty
        rng_of = lambda r: cin[cuts[r]:cuts[r + 1]]
        phase = np.stack([gpsiq.chain_summary(rng_of(r), fs, ns) for r in range(8)])
        drift = np.stack([gpsiq.chain_summary(rng_of(r), fs, ns, start=gpsiq.chain_fold(phase[:r])) for r in range(8)])
        maps = [gpsiq.chain_maps(rng_of(r), fs, ns, start=gpsiq.chain_fold(drift[:r]), max_stretches=8)[0] for r in range(8)]
        ranges = np.stack([gpsiq.chain_range(rng_of(r), maps[r], fs, ns) for r in range(8)])
        true_end, true_prn, true_known = np.zeros((8, nc)), np.zeros((8, nc), dtype=np.int32), np.zeros(8, dtype=np.uint8)
        got = [None] * 8
        for rounds in range(1, 10):
            _, carr_at, prn_at, known = gpsiq.chain_range_fold(ranges, true_end, true_prn, true_known)
            for r in range(8):
                if got[r] is None and known[r].all():
                    got[r] = gpsiq.chain_link(rng_of(r), maps[r], fs, ns, carr_at[r] if r else None, prn_at[r] if r else None)
            for r in range(8):                                    # (the exchange: what has been linked by the end of this round)
                if got[r] is not None:
                    true_known[r], true_end[r], true_prn[r] = 1, got[r][1], got[r][2]
            if true_known.all():
                break
        assert true_known.all() and rounds <= 8, rounds
        rounds_total += rounds
        for r in range(8):
            assert got[r][0].tobytes() == want_start[cuts[r]:cuts[r + 1]].tobytes(), (seed, r)
        every, carr_at, prn_at, _ = gpsiq.chain_range_fold(ranges, true_end, true_prn, true_known)
        assert every and carr_at[8].tobytes() == want_end.tobytes() and prn_at[8].tobytes() == want_prn.tobytes(), seed
    print("rounds per timeline:", rounds_total / 12)
    assert rounds_total <= (12 * 3 if not modes else 12 * 8), rounds_total


def test_soak_program_random_and_adversarial_timelines(tmp_path):
    """tests/chain_parallel.cpp: exact-tie addends, Doppler through zero, re-seeded and unused slots, continued and
    range-sharded timelines, 1..16 stretches, all sample rates."""
    csrc = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc")
    exe = str(tmp_path / "chain_parallel")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", exe,
                    os.path.join(ROOT, "tests", "chain_parallel.cpp"), os.path.join(csrc, "gpsiq_host.cpp"), "-lpthread", "-lm"], check=True)
    for seed in (1, 2):
        r = subprocess.run([exe, str(seed), "45"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and " bad=0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]



def test_lane_code_under_asan_and_ubsan(tmp_path):
    """The same program with -fsanitize=address,undefined (no recovery): the lanes' shifts, 128-bit arithmetic, bit casts and table
    indices on random and adversarial timelines -- the code the kernels run, where a sanitizer can see it."""
    csrc = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc")
    exe = str(tmp_path / "chain_parallel_san")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", exe,
                    os.path.join(ROOT, "tests", "chain_parallel.cpp"), os.path.join(csrc, "gpsiq_host.cpp"), "-lpthread", "-lm"], check=True)
    r = subprocess.run([exe, "5", "24"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " bad=0" in r.stdout and "runtime error" not in r.stderr, r.stdout[-2000:] + r.stderr[-2000:]

# ---- on the GPU --------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    c = gpsiq.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fs,nsamp,nblocks,nchan", [(2.6e6, 260000, 300, 16), (10e6, 1000000, 120, 12), (25e6, 2500000, 100, 16), (2.6e6, 33333, 700, 5)])
@pytest.mark.parametrize("stretches", [3, 8, 16, 32])
def test_device_maps_and_link_equal_the_serial_chain(ctx, fs, nsamp, nblocks, nchan, stretches):
    cin = timeline(nblocks + stretches, nblocks, nchan)
    want = gpsiq.reference_chain(cin, fs, nsamp)
    before = gpsiq.chain_stats()
    maps, end, ms = gpsiq.chain_maps(cin, fs, nsamp, max_stretches=stretches, ctx=ctx)
    assert_same_chain(gpsiq.chain_link(cin, maps, fs, nsamp), want)
    linked, walked = (a - b for a, b in zip(gpsiq.chain_stats(), before))
    assert linked > 0.6 * (cin["prn"] > 0).sum(), (linked, walked)
    # the estimator state the device hands on == the host twin's (phase exactly, drift to rounding)
    _, end_host = gpsiq.chain_maps(cin, fs, nsamp, max_stretches=1)
    for f in ("r_hi", "r_lo", "prn"):
        assert np.array_equal(end[f], end_host[f]), f
    assert np.allclose(end["drift"], end_host["drift"], rtol=1e-6, atol=1e-18)


@pytest.mark.gpu
def test_device_maps_continue_a_timeline(ctx):
    fs, nsamp = 2.6e6, 260000
    cin = timeline(77, 260, 16)
    want = gpsiq.reference_chain(cin, fs, nsamp)
    cut = 101
    m0, est, _ = gpsiq.chain_maps(cin[:cut], fs, nsamp, ctx=ctx)
    s0, e0, p0 = gpsiq.chain_link(cin[:cut], m0, fs, nsamp)
    m1, _, _ = gpsiq.chain_maps(cin[cut:], fs, nsamp, start=est, ctx=ctx)          # from the ESTIMATE the device handed on
    s1, e1, p1 = gpsiq.chain_link(cin[cut:], m1, fs, nsamp, e0, p0)
    assert_same_chain((np.concatenate([s0, s1]), e1, p1), want)
    exact = np.zeros(16, dtype=CHAIN_EST_DTYPE)
    exact["carr"], exact["prn"], exact["flags"], exact["f_carr"] = e0, p0, CHAIN_EXACT, cin["f_carr"][cut - 1]
    m2, _, _ = gpsiq.chain_maps(cin[cut:], fs, nsamp, start=exact, ctx=ctx)        # from the accumulator itself
    assert_same_chain((np.concatenate([s0, gpsiq.chain_link(cin[cut:], m2, fs, nsamp, e0, p0)[0]]), e1, p1), want)


@pytest.mark.gpu
def test_device_chain_is_mostly_maps_at_the_headline_workload(ctx):
    """2 000 blocks x 16 channels at 2.6 Msps, Doppler uniform in +-5 kHz (bench.py's reference-NCO leg): fewer than 2 % of the
    blocks fall back to the true walk, and the chain is the serial chain's."""
    from gpsiq.scenario import synth_blocks
    fs, nsamp = 2.6e6, 260000
    cin = gpsiq.chain_inputs(synth_blocks(2000, 16, seed=3))
    want = gpsiq.reference_chain(cin, fs, nsamp)
    before = gpsiq.chain_stats()
    maps, _, ms = gpsiq.chain_maps(cin, fs, nsamp, ctx=ctx)
    assert_same_chain(gpsiq.chain_link(cin, maps, fs, nsamp), want)
    linked, walked = (a - b for a, b in zip(gpsiq.chain_stats(), before))
    assert walked < 0.02 * cin.size, (linked, walked)
    print(f"device chain, 2000 x 16 blocks: kernels {ms:.3f} ms, linked {linked}, walked {walked}")


BAD = [("f_carr", np.nan), ("f_carr", np.inf), ("f_carr", 1.4e6), ("f_carr", -1.31e6), ("f_carr", 5e-324), ("f_carr", 0.0),
       ("carr_phase", 1.5), ("carr_phase", np.nan), ("carr_phase", 1.0), ("carr_phase", -0.25)]


def spoiled(field, value, nblocks=120, at=70):
    """A timeline with ONE input the NCO format may not take (or just takes) in a late block."""
    cin = timeline(5, nblocks, 6, modes=False)
    cin[field][at, 2] = value
    if field == "carr_phase":
        cin["prn"][at:, 2] = 1 + cin["prn"][0, 2] % 32                    # another satellite: the descriptor's phase is taken
    return cin


def chain_or_error(fn):
    try:
        return fn(), None
    except gpsiq.GpsiqError as e:
        return None, str(e)


def assert_same_outcome(cin, maps, fs, nsamp):
    want, werr = chain_or_error(lambda: gpsiq.reference_chain(cin, fs, nsamp))
    got, gerr = chain_or_error(lambda: gpsiq.chain_link(cin, maps, fs, nsamp))
    assert werr == gerr
    if want is not None:
        assert_same_chain(got, want)
    return werr


@pytest.mark.parametrize("field,value", BAD)
def test_inputs_outside_the_nco_format(field, value):
    """NaN / infinite / too fast Doppler, a phase outside [0, 1], and the edge cases that ARE in the format (zero and denormal
    Doppler, phase 1.0): level 1 terminates whatever it is given, and level 2 reports exactly what the serial chain reports."""
    fs, nsamp = 2.6e6, 260000
    cin = spoiled(field, value)
    errs = set()
    for stretches in (1, 8, 32):
        maps, _ = gpsiq.chain_maps(cin, fs, nsamp, max_stretches=stretches)
        errs.add(assert_same_outcome(cin, maps, fs, nsamp))
    assert len(errs) == 1
    err = errs.pop()
    in_format = (field == "f_carr" and abs(value) < 1e6) or (field == "carr_phase" and value == 1.0)
    assert (err is None) == in_format, err
    if err:
        assert "block 70" in err


@pytest.mark.gpu
@pytest.mark.parametrize("field,value", BAD)
def test_device_inputs_outside_the_nco_format(ctx, field, value):
    fs, nsamp = 2.6e6, 260000
    cin = spoiled(field, value)
    for stretches in (4, 32):
        maps, _, _ = gpsiq.chain_maps(cin, fs, nsamp, max_stretches=stretches, ctx=ctx)
        assert_same_outcome(cin, maps, fs, nsamp)


@pytest.mark.gpu
@pytest.mark.parametrize("field,value,at", [("f_carr", np.nan, 70), ("f_carr", 1.4e6, 5), ("carr_phase", 1.5, 119), ("f_code", 0.0, 100)])
def test_a_batch_with_the_chain_on_the_device_and_a_descriptor_outside_the_format(ctx, oracle, monkeypatch, field, value, at):
    """gpsiq_generate_batch in GPSIQ_NCO_REFERENCE with level 1 on the device (two launches, maps landing through callbacks) and
    a descriptor the walk or the evaluation refuses: the call reports the block, drains both chain streams and the rendering
    stream before it returns, and the context goes on working -- with the device chain -- bit-exactly."""
    import torch
    from gpsiq.abi import NCO_REFERENCE, SC08
    from gpsiq.scenario import synth_blocks
    monkeypatch.setenv("GPSIQ_CHAIN", "device")
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", "8")
    fs, ns, nb, nc = 2.6e6, 26000, 120, 6
    d = synth_blocks(nb, nc, seed=17)
    bad = d.copy()
    bad[field][at, 2] = value
    if field == "carr_phase":
        bad["prn"][at:, 2] = 1 + bad["prn"][0, 2] % 32
    buf = torch.zeros(nb * 2 * ns, dtype=torch.uint8, device="cuda")
    ctx.set_nco_mode(NCO_REFERENCE)
    try:
        for _ in range(2):
            with pytest.raises(gpsiq.GpsiqError) as ei:
                ctx.generate_batch(bad, ns, fs, SC08, device_ptr=buf.data_ptr())
            assert "block" in str(ei.value)
        before = gpsiq.chain_stats()
        ctx.generate_batch(d, ns, fs, SC08, device_ptr=buf.data_ptr())
        torch.cuda.synchronize()
        linked, walked = (a - b for a, b in zip(gpsiq.chain_stats(), before))
        assert linked > 0.5 * nb * nc, (linked, walked)                     # the chain did run through the device's maps
        got = buf.cpu().numpy().view(np.int8).reshape(nb, 2 * ns)
        carr, prev = None, None
        for b in range(nb):
            db = d[b].copy()
            if b:
                db["carr_phase"] = np.where((prev == db["prn"]) & (db["prn"] > 0), carr, db["carr_phase"])
            o, carr = oracle.block_float(db, ns, fs, SC08)
            prev = db["prn"].copy()
            if b in (0, 1, 7, 8, 60, 119):
                assert np.array_equal(got[b], o), b
    finally:
        ctx.set_nco_mode(0)
