"""Pin the oracle to the reference's own code (oracle/_ref, built from /root/reference).

Runs where /root/reference exists (this container).  On the GPU box the prebuilt
oracle/_ref/libgpsref.so travels with the snapshot, so these run there too; if it is
absent they skip and tests/test_golden.py (committed captures) is the pin.
"""
from fractions import Fraction

import numpy as np
import pytest

from gpsiq.abi import SC08, SC16, SINK_HACKRF, SINK_IQFILE, SINK_PLUTOSDR
from gpsiq.scenario import synth_blocks


def test_tables_match_reference(oracle, ref):
    s, c = oracle.tables()
    rs, rc = ref.tables()
    assert (s == rs).all() and (c == rc).all()


def test_ca_codes_match_reference(oracle, ref):
    for prn in range(1, 33):
        assert (oracle.codegen(prn) == ref.codegen(prn)).all(), prn


@pytest.mark.parametrize("fs,nchan,ss", [(3000000, 12, SC08), (2600000, 16, SC16), (10000000, 5, SC08)])
def test_float_restatement_is_bit_exact(oracle, ref, fs, nchan, ss):
    """oracle_block_float == the reference loop, every element, incl. the carried carr_phase."""
    d = synth_blocks(3, nchan, seed=fs + nchan)
    ns = fs // 10
    out, chunks, carr = ref.run_blocks(d, fs, ss, SINK_IQFILE)
    assert (chunks == 2 * ns).all() and len(chunks) == 3
    for b in range(3):
        db = d[b].copy()
        if b:
            db["carr_phase"] = carr[b - 1]
        fo, fc = oracle.block_float(db, ns, fs, ss)
        assert (fo == out[b * 2 * ns:(b + 1) * 2 * ns]).all()
        assert (fc == carr[b]).all()


def _closed_run(oracle, d, fs, ss):
    """oracle_block_float_closed over consecutive blocks, carrying carr_phase the way the loop does
    (re-seeded from the descriptor when a slot's PRN changes, gps.c:2208-2214)."""
    ns = fs // 10
    out, carr, prev = [], None, None
    for b in range(len(d)):
        db = d[b].copy()
        if b:
            keep = (prev == db["prn"]) & (db["prn"] > 0)
            db["carr_phase"] = np.where(keep, carr, db["carr_phase"])
        o, carr = oracle.block_float_closed(db, ns, fs, ss)
        out.append(o)
        prev = db["prn"].copy()
    return np.concatenate(out), carr


@pytest.mark.parametrize("fs,nchan,ss,nb,seed", [(2600000, 16, SC16, 60, 11), (3000000, 12, SC08, 20, 12),
                                                 (10000000, 16, SC16, 8, 13), (25000000, 16, SC16, 4, 14),
                                                 (1500000, 7, SC08, 10, 15)])
def test_float_loop_in_closed_form_is_the_reference(oracle, ref, fs, nchan, ss, nb, seed):
    """The reference's double accumulators are piecewise linear in exact integers (one piece per
    binade); evaluated piece by piece, with no per-sample recurrence, they give the reference's
    output on every element and its carried carr_phase after every block (whole-run T2 = 0).
    tests/t2_report.py runs the same check over 299 / 100 blocks per rate."""
    d = synth_blocks(nb, nchan, seed=seed)
    want, _, carr = ref.run_blocks(d, fs, ss, SINK_IQFILE)
    got, carr_end = _closed_run(oracle, d, fs, ss)
    assert np.array_equal(got, want)
    assert np.array_equal(carr_end, carr[-1])


def test_float_closed_form_edge_cases(oracle):
    """Zero and negative Doppler, a phase that starts at 0, a step below half an ulp, one chip per
    sample, very short blocks: closed form == recurrence (both in the oracle; the recurrence is
    pinned to the reference above)."""
    d = synth_blocks(1, 8, seed=5)[0]
    d["f_carr"] = [0.0, -4999.7, 4999.7, 1e-9, -1e-9, 0.25, -1234.5, 3e-14]
    d["carr_phase"] = [0.0, 0.0, 0.999999999999, 0.5, 0.5, 0.0, 1e-300, 0.75]
    d["code_phase"][:3] = [0.0, 1022.9999999999, 511.99999999999994]
    for fs, ns in ((2.6e6, 70001), (1.023e6, 30000), (25e6, 120000), (2.6e6, 1), (2.6e6, 2), (2.6e6, 3)):
        a, ca = oracle.block_float(d, ns, fs, SC16)
        b, cb = oracle.block_float_closed(d, ns, fs, SC16)
        assert np.array_equal(a, b), (fs, ns)
        assert np.array_equal(ca, cb), (fs, ns)


def _explain_mismatch(d, fs, n, drift_steps):
    """True if at sample n some channel's exact (rational) phase lies within the float
    path's own accumulated-rounding bound of a chip or LUT boundary."""
    delt = 1.0 / fs
    for c in range(len(d)):
        if d[c]["prn"] <= 0:
            continue
        T = Fraction(float(d[c]["code_phase"])) + n * Fraction(float(d[c]["f_code"]) * delt)
        fr = T - int(T)
        # each += may lose up to half an ulp of a value < 1024: 2^-44 chip
        if min(fr, 1 - fr) <= drift_steps * Fraction(1, 2 ** 44):
            return True
        P = (Fraction(float(d[c]["carr_phase"])) + n * Fraction(float(d[c]["f_carr"]) * delt)) * 512
        pf = P - (P.numerator // P.denominator)
        # value < 1: half ulp = 2^-54 cycle = 2^-45 LUT step
        if min(pf, 1 - pf) <= drift_steps * Fraction(1, 2 ** 45):
            return True
    return False


@pytest.mark.parametrize("fs,nchan,ss,seed", [(3000000, 12, SC08, 1), (2600000, 16, SC16, 2),
                                              (10000000, 16, SC16, 3), (25000000, 16, SC08, 25000016)])
def test_fixed_point_model_vs_reference_T1(oracle, ref, fs, nchan, ss, seed):
    """Tier T1: the closed-form fixed-point model, given the block-start state, equals the
    reference's double-accumulator loop except where a real-valued phase sits within the
    double path's own rounding drift of a boundary; every differing sample is checked
    against that bound with exact rational arithmetic, and they must be rare."""
    d = synth_blocks(2, nchan, seed=seed)
    ns = fs // 10
    out, _, carr = ref.run_blocks(d, fs, ss, SINK_IQFILE)
    total = 0
    for b in range(2):
        db = d[b].copy()
        if b:
            db["carr_phase"] = carr[b - 1]
        q, _ = oracle.quantize(db, fs, ns)
        fx = oracle.block_fixed(q, ns, ss, seq=True)
        bad = np.nonzero(fx != out[b * 2 * ns:(b + 1) * 2 * ns])[0]
        for n in sorted(set(int(i) // 2 for i in bad)):
            assert _explain_mismatch(db, fs, n, n + 1), (b, n)
        total += len(set(bad // 2))
    assert total <= 8, total


def test_fixed_forms_agree(oracle):
    d = synth_blocks(1, 16, seed=77)[0]
    q, _ = oracle.quantize(d, 2.6e6, 50000)
    a = oracle.block_fixed(q, 50000, SC16)
    b = oracle.block_fixed(q, 50000, SC16, seq=True)
    assert (a == b).all()
    assert (oracle.block_fixed_range(q, 12345, 777, SC16) == a[2 * 12345: 2 * (12345 + 777)]).all()


@pytest.mark.parametrize("sink", [SINK_IQFILE, SINK_HACKRF, SINK_PLUTOSDR])
def test_chunk_plan_matches_reference(oracle, ref, sink):
    d = synth_blocks(3, 2, seed=5)
    out, chunks, _ = ref.run_blocks(d, 3000000, SC08, sink)
    plan = oracle.chunk_plan(sink, 600000, 3)
    assert (plan == chunks).all()
    assert len(out) == chunks.sum()


def test_compute_code_phase_consistency(ref):
    """computeCodePhase (gps.c:2033-2064) leaves dataBit/codeCA equal to the closed-form
    functions of (dwrd, iword, ibit) and (ca, code_phase) that gpsiq_chan_t relies on."""
    rng = np.random.default_rng(3)
    for _ in range(20):
        dwrd = rng.integers(0, 1 << 30, size=60, dtype=np.uint32)
        t = float(rng.integers(0, 25)) + 0.1 * float(rng.integers(0, 10))
        rho0 = 2.0e7 + 5e6 * rng.random()
        ch = ref.compute_code_phase(rho0, (2100, 1000.0 + t), (2100, 1000.0), rho0 + 60.0 * (rng.random() - 0.5), 0.1,
                                    dwrd, int(rng.integers(1, 33)))
        assert 0 <= ch["iword"] < 60 and 0 <= ch["ibit"] < 30 and 0 <= ch["icode"] < 20
        assert 0.0 <= ch["code_phase"] < 1023.0
        assert abs(ch["f_code"] - (1.023e6 + ch["f_carr"] / 1540.0)) < 1e-6
