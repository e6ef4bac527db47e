"""GPSIQ_NCO_REFERENCE, host half, on the CPU: gpsiq_reference_batch (carrier walk, candidate search, patches)
+ the fixed-point oracle + the patches applied == the float loop (oracle_block_float, pinned to the reference in
test_oracle_vs_ref.py), on random scenarios and on the corners of the double arithmetic."""
import numpy as np
import pytest

import gpsiq
from _oracle import apply_patches
from gpsiq.abi import SC08, SC16
from gpsiq.scenario import synth_blocks


def float_chain(oracle, d, fs, ns, ss):
    out, carr, prev = [], None, None
    for b in range(len(d)):
        db = d[b].copy()
        if b:
            db["carr_phase"] = np.where((prev == db["prn"]) & (db["prn"] > 0), carr, db["carr_phase"])
        o, carr = oracle.block_float(db, ns, fs, ss)
        out.append(o)
        prev = db["prn"].copy()
    return np.stack(out), carr


def product_chain(oracle, d, fs, ns, ss):
    q, patches, carr = gpsiq.reference_blocks(d, fs, ns)
    out = []
    for b in range(len(d)):
        o = oracle.block_fixed(q[b], ns, ss, seq=True)
        apply_patches(oracle, q[b], o, patches[patches["block"] == b], ss)
        out.append(o)
    return np.stack(out), carr, patches


@pytest.mark.parametrize("fs,ns,nchan,nb,ss,seed", [(2.6e6, 260000, 16, 6, SC16, 1), (25e6, 2500000, 16, 2, SC08, 3032),
                                                    (10e6, 300000, 9, 5, SC16, 3), (1.2e6, 120000, 5, 4, SC08, 4)])
def test_random_scenarios(oracle, fs, ns, nchan, nb, ss, seed):
    d = synth_blocks(nb, nchan, seed=seed)
    d["prn"][nb // 2:, 1] = 0
    want, carr_want = float_chain(oracle, d, fs, ns, ss)
    got, carr, _ = product_chain(oracle, d, fs, ns, ss)
    assert np.array_equal(got, want)
    act = d[-1]["prn"] > 0
    assert np.array_equal(carr[act], carr_want[act])


def test_corners_of_the_double_arithmetic(oracle):
    """Phases exactly on a power of two with a negative step (the sum falls into the finer binade underneath), steps
    below half / a quarter of an ulp (the phase moves only from a binade edge, or not at all), zero Doppler, phases on
    LUT boundaries where every sample is a candidate, a step that is an exact tie, one chip per sample."""
    d = synth_blocks(3, 12, seed=5)
    d["f_carr"][:] = [0.0, -4999.7, 4999.7, 1e-9, -1e-9, 0.25, -1234.5, 3e-14, -3e-10, -1.2e-9, 2.6e6 / 2 ** 20, -2.6e6 / 2 ** 21]
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    d["carr_phase"][:] = [0.0, 0.0, 0.999999999999, 0.5, 0.5, 0.0, 1e-300, 0.75, 0.25, 0.125, 0.5 + 2.0 ** -53, 0.25]
    d["code_phase"][0, :4] = [0.0, 1022.9999999999, 511.99999999999994, 512.0]
    for fs, ns in ((2.6e6, 70001), (25e6, 20000), (2.6e6, 1), (2.6e6, 65), (1.023e6, 30000)):
        want, carr_want = float_chain(oracle, d, fs, ns, SC16)
        got, carr, patches = product_chain(oracle, d, fs, ns, SC16)
        assert np.array_equal(got, want), (fs, ns)
        assert np.array_equal(carr, carr_want), (fs, ns)
        # and the oracle's own closed form of the double path agrees as well
        cin = d[0]["carr_phase"].copy()
        for b in range(3):
            db = d[b].copy()
            db["carr_phase"] = cin
            o, cin = oracle.block_float_closed(db, ns, fs, SC16)
            assert np.array_equal(o, want[b]), (fs, ns, b)
        assert np.array_equal(cin, carr_want)


def test_carrier_walk_fuzz_against_the_plain_loop(oracle):
    """The carrier phase gpsiq_reference_batch hands on after a block (per-binade run lengths from a table, real
    additions at the crossings) against the plain loop of additions, over addends from far below an ulp to half a
    cycle per sample, both signs, and start phases on and next to powers of two."""
    rng = np.random.default_rng(77)
    fs = 2.6e6
    for it in range(400):
        ns = int(rng.integers(1, 30000))
        d = synth_blocks(1, 16, seed=1000 + it)
        mag = 10.0 ** rng.uniform(-13.0, np.log10(0.49 * fs), 16)
        if it % 3 == 1:                                   # Doppler-sized addends: the chain's own walk (NcoWalk)
            mag = rng.uniform(1.0, 9000.0, 16)
        d["f_carr"][0] = mag * rng.choice([-1.0, 1.0], 16)
        if it % 5 == 0:                                   # addends that are exact binary fractions of a cycle: ties
            d["f_carr"][0, :8] = fs * 2.0 ** -rng.integers(8, 60, 8) * rng.choice([-1.0, 1.0, 1.5, -3.0], 8)
        d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
        ph = rng.uniform(0.0, 1.0, 16)
        k = rng.integers(1, 40, 16)
        edge = 2.0 ** -k.astype(np.float64)
        ph = np.where(rng.random(16) < 0.3, edge * (1.0 + rng.integers(-2, 3, 16) * 2.0 ** -52), ph)
        d["carr_phase"][0] = np.clip(ph, 0.0, np.nextafter(1.0, 0.0))
        _, want = oracle.block_float(d[0], ns, fs, SC08)
        _, _, got = gpsiq.reference_blocks(d, fs, ns)
        assert got.tobytes() == want.tobytes(), (it, ns, d["f_carr"][0][got != want], d["carr_phase"][0][got != want])


def test_carrier_wrap_that_rounds_to_one(oracle):
    """A negative addend taking the phase a hair below zero: the wrap y + 1.0 rounds to exactly 1.0 (the reference
    then indexes its table at 512, see evaluate_block) and the walk goes on from 1.0, outside every binade of [0, 1)."""
    fs, ns = 2.6e6, 5000
    d = synth_blocks(1, 16, seed=9)
    i = np.arange(16)
    c = -(2.0 ** -(8.0 + i % 5))
    d["f_carr"][0] = c * fs
    assert np.array_equal(d["f_carr"][0] / fs, c)
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    x0 = np.nextafter(-c, 0.0)
    x0[i % 3 == 1] = np.nextafter(x0[i % 3 == 1], 0.0)            # one or two ulps (2^-61..2^-65) short of |c|: the first sum
    d["carr_phase"][0] = x0                                       # is that much below zero, and + 1.0 rounds to 1.0
    assert ((d["carr_phase"][0] + c) + 1.0 == 1.0).all() and (d["carr_phase"][0] + c < 0.0).all()
    _, want = oracle.block_float(d[0], ns, fs, SC08)
    _, _, got = gpsiq.reference_blocks(d, fs, ns)
    assert got.tobytes() == want.tobytes()


def test_a_block_that_starts_on_one(oracle):
    """The wrap that rounds to exactly 1.0 as the LAST addition of a block: the reference simply goes on from 1.0 in
    the next block (its sample 0 indexes the table at 512, where the library takes entry 511).  The chain accepts the
    value -- handed back by the caller or met inside a batch -- quantises the block as phase 0 and patches sample 0."""
    fs = 2.6e6
    d = synth_blocks(1, 16, seed=9)
    i = np.arange(16)
    c = -(2.0 ** -(8.0 + i % 5))
    d["f_carr"][0] = c * fs
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    d["carr_phase"][0] = np.nextafter(-c, 0.0)
    assert ((d["carr_phase"][0] + c) + 1.0 == 1.0).all()
    _, _, one = gpsiq.reference_blocks(d, fs, 1)                      # a block of one sample ends on the wrap
    assert (one == 1.0).all()
    # (a) the caller hands 1.0 back, as the drop-in binding does with chan[i].carr_phase
    ns = 5000
    d2 = synth_blocks(2, 16, seed=9)
    d2["f_carr"][:] = d["f_carr"][0]
    d2["f_code"] = 1.023e6 + d2["f_carr"] / 1540.0
    d2["carr_phase"][0] = one
    want, carr_want = float_chain(oracle, d2, fs, ns, SC16)
    got, carr, patches = product_chain(oracle, d2, fs, ns, SC16)
    first = patches[(patches["block"] == 0) & (patches["sample"] == 0)]
    assert len(first) == 16 and (first["lut"] == 511).all()
    # sample 0 of block 0 is the reference's out-of-table read: compare everything after it
    assert np.array_equal(got[0, 2:], want[0, 2:]) and np.array_equal(got[1], want[1])
    assert np.array_equal(carr, carr_want)
    # (b) the same value met inside a batch: blocks of one sample, the second one starts on 1.0
    d3 = synth_blocks(3, 16, seed=9)
    d3["f_carr"][:] = d["f_carr"][0]
    d3["f_code"] = 1.023e6 + d3["f_carr"] / 1540.0
    d3["carr_phase"][0] = d["carr_phase"][0]
    q, patches, carr = gpsiq.reference_blocks(d3, fs, 1)
    _, carr_want = float_chain(oracle, d3, fs, 1, SC16)
    assert np.array_equal(carr, carr_want)
    assert (q[1]["carr_phase"] == 0).all() and len(patches[(patches["block"] == 1) & (patches["lut"] == 511)]) == 16


def chain_case(oracle, fs, ns, f_carr, x0):
    """gpsiq_reference_batch's carried phase for 16 channels at once against the plain loop of gps.c:2821-2826."""
    d = synth_blocks(1, 16, seed=1)
    d["f_carr"][0] = f_carr
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    d["carr_phase"][0] = x0
    _, _, got = gpsiq.reference_blocks(d, fs, ns)
    want = np.array([oracle.carrier_chain(x0[i], f_carr[i] * (1.0 / fs), ns) for i in range(16)])
    assert got.tobytes() == want.tobytes(), (fs, ns, f_carr[got != want], x0[got != want])


def test_wrap_to_wrap_table_against_the_plain_loop(oracle):
    """The carrier chain's table of whole cycles (csrc/gpsiq_exact.cpp, NcoWalk): blocks of hundreds of carrier
    cycles, where all but the first dozen or two cycles are look-ups -- Doppler-sized addends of both signs, addends
    that are short binary fractions (every rounding a tie or exact), start phases on and next to binade edges and on
    the post-wrap grid, whole-run lengths at all four sample rates."""
    rng = np.random.default_rng(2024)
    cases = [(2.6e6, 260000), (3e6, 300000), (10e6, 1000000), (25e6, 2500000), (2.6e6, 259999), (2.6e6, 70001)]
    for it in range(36):
        fs, ns = cases[it % len(cases)]
        mag = rng.uniform(300.0, 6500.0, 16)
        if it % 6 == 4:
            mag = rng.uniform(8000.0, 39000.0, 16)                    # up to 2^-6 cycle per sample at 2.6 Msps: the table's limit
        f = mag * rng.choice([-1.0, 1.0], 16)
        if it % 4 == 1:                                               # dyadic addends: c = k * 2^-j exactly
            f[:8] = fs * rng.integers(1, 16, 8) * 2.0 ** -rng.integers(12, 26, 8) * rng.choice([-1.0, 1.0], 8)
        if it % 4 == 3:                                               # addends one ulp either side of a dyadic one
            c = rng.integers(1, 8, 8) * 2.0 ** -rng.integers(11, 20, 8) * rng.choice([-1.0, 1.0], 8)
            f[8:] = np.nextafter(c, rng.choice([-1.0, 1.0], 8)) * fs
        x0 = rng.uniform(0.0, 1.0, 16)
        k = rng.integers(1, 30, 16)
        x0 = np.where(rng.random(16) < 0.25, 2.0 ** -k.astype(np.float64) * (1.0 + rng.integers(-2, 3, 16) * 2.0 ** -52), x0)
        x0 = np.where(rng.random(16) < 0.2, rng.integers(0, 2 ** 40, 16) * 2.0 ** -52, x0)          # the post-wrap grid itself
        x0 = np.where(rng.random(16) < 0.1, 1.0 - rng.integers(1, 2 ** 40, 16) * 2.0 ** -53, x0)
        chain_case(oracle, fs, ns, f, np.clip(x0, 0.0, np.nextafter(1.0, 0.0)))


def test_wrap_to_wrap_table_chained_over_blocks(oracle):
    """A run of blocks with the Doppler drifting from block to block, as a moving receiver has it: the phase handed from
    block to block is the plain loop's, for every block."""
    fs, ns, nb = 2.6e6, 260000, 12
    rng = np.random.default_rng(5)
    d = synth_blocks(nb, 16, seed=3)
    f0 = rng.uniform(-5000.0, 5000.0, 16)
    d["f_carr"] = f0[None, :] + np.cumsum(rng.uniform(-0.8, 0.8, (nb, 16)), axis=0)
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    _, _, got = gpsiq.reference_blocks(d, fs, ns)
    x = d["carr_phase"][0].copy()
    for b in range(nb):
        x = np.array([oracle.carrier_chain(x[i], d["f_carr"][b, i] * (1.0 / fs), ns) for i in range(16)])
    assert got.tobytes() == x.tobytes()


@pytest.mark.parametrize("fs,ns,seed", [(10e6, 300000, 1), (25e6, 100000, 2), (2.6e6, 130000, 3)])
def test_candidates_evaluated_from_the_wrap_tables(oracle, fs, ns, seed):
    """Blocks made to have many candidate samples in both accumulators (carrier phases a hair off LUT boundaries, code phases a
    hair off chip boundaries, addends that bring them back there again and again): the states the wrap-to-wrap walks report at
    those samples -- carrier cycles and code periods -- give the reference's LUT index and sign at every one of them."""
    rng = np.random.default_rng(seed)
    d = synth_blocks(2, 16, seed=40 + seed)
    k = rng.integers(0, 512, 16)
    d["carr_phase"][0] = (k + rng.choice([1e-13, -1e-13, 3e-12, 0.0], 16)) / 512.0 % 1.0
    d["code_phase"][:] = (rng.integers(0, 1023, (2, 16)) + rng.choice([1e-10, 2e-9, 0.0, 1.0 - 1e-10], (2, 16))) % 1023.0
    f = rng.uniform(500.0, 6000.0, 16) * rng.choice([-1.0, 1.0], 16)
    f[:4] = fs / 512.0 / rng.integers(3, 40, 4) * rng.choice([-1.0, 1.0], 4)        # a whole number of samples per LUT step: back on the boundary every time
    d["f_carr"][:] = f
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    d["f_code"][:, 4:8] = fs / rng.integers(3, 30, 4)                               # a whole number of samples per chip
    want, carr_want = float_chain(oracle, d, fs, ns, SC16)
    got, carr, patches = product_chain(oracle, d, fs, ns, SC16)
    assert np.array_equal(got, want)
    assert np.array_equal(carr, carr_want)
    assert len(patches) > 0


def test_wrap_to_wrap_table_with_exact_tie_binades(oracle):
    """Addends whose mantissa ends in 1 followed by s - 1 zeros: in the binade s above the addend's own, c / ulp ends in exactly
    one half and every addition there is a tie (to even).  One addend in 4 has such a binade above the plain additions; the
    fast walk takes one real addition to make the mantissa even and the even step from its table after that, and tables
    whole cycles as for any other addend (except a tie in the top binade of a descending carrier: the probing walk)."""
    rng = np.random.default_rng(99)
    fs, ns = 2.6e6, 260000
    delt = 1.0 / fs
    ties = 0
    for it in range(24):
        f = rng.uniform(400.0, 6000.0, 16) * rng.choice([-1.0, 1.0], 16)
        s = rng.integers(2, 13, 16).astype(np.uint64)                  # tie binade: the addend's low s bits = 100..0
        want = (((f * delt).view(np.uint64) >> s << s) | (np.uint64(1) << (s - np.uint64(1)))).view(np.float64)
        f = want / delt                                                # the library forms the addend as f_carr * (1 / fs): aim at it
        for _ in range(4):                                             # ... and step f until the product is the wanted double
            got = f * delt
            f = np.where(got < want, np.nextafter(f, np.inf), np.where(got > want, np.nextafter(f, -np.inf), f))
        hit = (f * delt) == want
        ties += int(hit.sum())
        x0 = rng.uniform(0.0, 1.0, 16)
        x0 = np.where(rng.random(16) < 0.3, rng.integers(0, 2 ** 40, 16) * 2.0 ** -52, x0)
        chain_case(oracle, fs, ns if it % 3 else 70001, f, np.clip(x0, 0.0, np.nextafter(1.0, 0.0)))
    assert ties > 200                                                  # most of the 384 addends really are of that kind


def test_candidates_decided_from_the_start_state_equal_the_walked_ones():
    """The evaluation half decides nearly every candidate from the block's START state (the drift enclosure, which is what
    lets blocks be evaluated on any thread, device or rank once the chain has run); GPSIQ_NO_DRIFT=1 walks both accumulators to
    every candidate instead (the round-3 path).  Same descriptors, patches and carried phases on random timelines at 10-25 Msps
    (about one candidate per block and channel, start states pushed next to LUT / chip boundaries); tests/soak_drift.py is the
    time-bounded form (1.4 M decisions per minute)."""
    import soak_drift
    rng = np.random.default_rng(11)
    s0 = gpsiq.reference_stats()
    patches = sum(soak_drift.one(rng) for _ in range(1500))
    s1 = gpsiq.reference_stats()
    assert patches > 200 and s1[1] - s0[1] > 20000                         # decided without a walk, and checked against the walk
    assert (s1[2] - s0[2]) + (s1[3] - s0[3]) >= (s1[0] - s0[0]) // 2       # the NO_DRIFT halves walked everything


@pytest.mark.parametrize("fs", [2.6e6, 25e6])
def test_table_built_from_the_previous_blocks_layout(oracle, fs):
    """On hosts with AVX-512 the wrap-to-wrap table of a block is built eight cycles at a time, and from the second block of a
    channel on the start states walked are where the previous block's entries lay (the Doppler moves by a fraction of a hertz
    per block).  Forty blocks per channel with a slow Doppler drift, a jump (no usable hint), a sign change and a slot that
    is re-allocated: the carried phase after every block == the plain loop of gps.c:2821-2826."""
    ns = int(fs) // 10
    nb, nc = 40, 16
    d = synth_blocks(nb, nc, seed=404, doppler_hz=6000.0, drift_hz=0.3)
    d["f_carr"][20:, 3] += 700.0                        # a jump: the hint does not apply to block 20
    d["f_carr"][10:, 5] *= -1.0                         # the other direction from block 10 on
    d["f_carr"][:, 7] = 30.0                            # three cycles per block: never tabled
    d["prn"][25:, 9] = 17                               # re-allocated: the chain restarts from its carr_phase
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    x = d["carr_phase"][0].copy()
    for b in range(nb):
        x = np.where(b == 25, np.where(np.arange(nc) == 9, d["carr_phase"][b], x), x)
        _, _, got = gpsiq.reference_blocks(d[:b + 1], fs, ns) if b in (0, 1, 19, 20, 21, 39) else (None, None, None)
        x = np.array([oracle.carrier_chain(x[i], d["f_carr"][b, i] * (1.0 / fs), ns) for i in range(nc)])
        if got is not None:
            assert got.tobytes() == x.tobytes(), b


def test_chain_and_seeded_edge_cases():
    """gpsiq_reference_chain / gpsiq_reference_seeded: empty timelines, error reporting (a start phase outside [0, 1], a NaN
    Doppler names its block), and a chain continued from the state another call published (carr_in / prn_in) goes on exactly."""
    from gpsiq.abi import CHAIN_IN_DTYPE, CHAN_DTYPE
    st, e, last = gpsiq.reference_chain(np.zeros((0, 5), dtype=CHAIN_IN_DTYPE), 2.6e6, 260000)
    assert st.shape == (0, 5) and not e.any() and not last.any()
    q, p = gpsiq.reference_seeded(np.zeros((0, 5), dtype=CHAN_DTYPE), 2.6e6, 260000, np.zeros((0, 5)))
    assert q.shape == (0, 5) and len(p) == 0
    d = synth_blocks(3, 4, seed=1)
    with pytest.raises(gpsiq.GpsiqError, match="start phase"):
        gpsiq.reference_seeded(d, 2.6e6, 260000, np.full((3, 4), 1.5))
    bad = d.copy()
    bad["f_carr"][1, 2] = np.nan
    for call in (lambda: gpsiq.reference_chain(gpsiq.chain_inputs(bad), 2.6e6, 260000), lambda: gpsiq.reference_blocks(bad, 2.6e6, 260000)):
        with pytest.raises(gpsiq.GpsiqError, match="block 1"):
            call()
    d = synth_blocks(9, 4, seed=2)
    d["prn"][6:, 1] = 23                                      # re-allocated in the second half
    cin = gpsiq.chain_inputs(d)
    s_all, e_all, l_all = gpsiq.reference_chain(cin, 10e6, 1000000)
    s1, e1, l1 = gpsiq.reference_chain(cin[:4], 10e6, 1000000)
    s2, e2, l2 = gpsiq.reference_chain(cin[4:], 10e6, 1000000, carr_in=e1, prn_in=l1)
    assert np.array_equal(np.vstack([s1, s2]), s_all) and np.array_equal(e2, e_all) and np.array_equal(l2, l_all)
    # columns are independent: a subset of the channels gives those channels' columns
    s_sub, e_sub, _ = gpsiq.reference_chain(np.ascontiguousarray(cin[:, 1:3]), 10e6, 1000000)
    assert np.array_equal(s_sub, s_all[:, 1:3]) and np.array_equal(e_sub, e_all[1:3])
