"""Time-bounded random soak of the DEVICE carrier chain (gpsiq_chain_maps_device + gpsiq_chain_link) against the serial chain on
host threads (gpsiq_reference_chain = NcoWalk, itself soaked against the plain loop of gps.c:2821-2826 by soak_carrier_walk.py):
random timelines -- Doppler ramps, some through zero, slots re-allocated and unused, exact-tie addends (a power-of-two sample
rate), all BASELINE rates and short blocks, 1..32 stretches, continued timelines -- every start state, end state and last_prn
equal, bit for bit.  Also the whole GPSIQ_NCO_REFERENCE batch call with the chain on the device against the same call with the
chain on host threads (GPSIQ_CHAIN), every output byte.   usage: python tests/soak_chain_device.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpsiq  # noqa: E402
from gpsiq.abi import CHAIN_EST_DTYPE, CHAIN_EXACT, NCO_REFERENCE, SC08, SC16  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402
from test_chain_parallel import timeline  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = gpsiq.Context(0)
    t_end = time.time() + budget
    runs = blocks = bad = batches = 0
    s0 = gpsiq.chain_stats()
    while time.time() < t_end:
        fs, nsamp = [(2.6e6, 260000), (3.0e6, 300000), (10e6, 1000000), (25e6, 2500000), (2097152.0, 209715), (2.6e6, int(rng.integers(20000, 90000)))][int(rng.integers(0, 6))]
        nb, nc = int(rng.integers(20, 700)), int(rng.integers(1, 17))
        cin = timeline(int(rng.integers(0, 1 << 30)), nb, nc)
        if rng.integers(0, 4) == 0:                                  # addends with trailing zeros: exact-tie binades
            z = int(rng.integers(20, 44))
            f = cin["f_carr"].view(np.uint64)
            f &= ~np.uint64((1 << z) - 1)
        seg = int(rng.integers(1, 33))
        want = gpsiq.reference_chain(cin, fs, nsamp)
        cut = int(rng.integers(0, nb))
        if cut and rng.integers(0, 2):
            m0, est, _ = gpsiq.chain_maps(cin[:cut], fs, nsamp, max_stretches=seg, ctx=ctx)
            s_0, e0, p0 = gpsiq.chain_link(cin[:cut], m0, fs, nsamp)
            if rng.integers(0, 2):                                   # from the estimate the device handed on, or from the accumulator itself
                est = np.zeros(nc, dtype=CHAIN_EST_DTYPE)
                est["carr"], est["prn"], est["flags"], est["f_carr"] = e0, p0, CHAIN_EXACT, cin["f_carr"][cut - 1]
            m1, _, _ = gpsiq.chain_maps(cin[cut:], fs, nsamp, start=est, max_stretches=seg, ctx=ctx)
            s_1, e1, p1 = gpsiq.chain_link(cin[cut:], m1, fs, nsamp, e0, p0)
            got = (np.concatenate([s_0, s_1]), e1, p1)
        else:
            maps, _, _ = gpsiq.chain_maps(cin, fs, nsamp, max_stretches=seg, ctx=ctx)
            got = gpsiq.chain_link(cin, maps, fs, nsamp)
        ok = all(g.tobytes() == w.tobytes() for g, w in zip(got, want))
        bad += not ok
        if not ok:
            print(f"MISMATCH: fs {fs} nsamp {nsamp} blocks {nb} slots {nc} stretches {seg} cut {cut}", flush=True)
        runs += 1
        blocks += nb * nc
        if runs % 8 == 0:                                            # the batch call either way
            ss = SC08 if rng.integers(0, 2) else SC16
            fsb, nsb = (2.6e6, 26000) if rng.integers(0, 2) else (10e6, 100000)
            d = synth_blocks(int(rng.integers(50, 400)), 16, seed=int(rng.integers(0, 1 << 30)), doppler_hz=float(rng.choice([300.0, 5000.0, 60000.0])))
            ctx.set_nco_mode(NCO_REFERENCE)
            os.environ["GPSIQ_CHAIN"] = "device"
            a = ctx.generate_batch(d, nsb, fsb, ss)
            os.environ["GPSIQ_CHAIN"] = "host"
            b = ctx.generate_batch(d, nsb, fsb, ss)
            del os.environ["GPSIQ_CHAIN"]
            ctx.set_nco_mode(0)
            if not np.array_equal(a, b):
                bad += 1
                print("MISMATCH in the batch call", flush=True)
            batches += 1
    s1 = gpsiq.chain_stats()
    ctx.close()
    print(f"seed {seed}: {runs} timelines, {blocks} blocks x slots, linked {s1[0] - s0[0]}, walked {s1[1] - s0[1]}, {batches} batch calls either way, {bad} mismatches", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
