#!/usr/bin/env python3
"""Tier T2 report, CPU only, needs oracle/_ref.

    python tests/t2_report.py [fs] [nchan] [nblocks]            default model
    python tests/t2_report.py --closed [fs] [nchan] [nblocks]   the double NCOs in closed form

Default: a whole BASELINE config-1 length run (299 blocks = 29.9 s) through the reference's own loop
(carrier phase carried by its double accumulator) against the fixed-point oracle with the library's exact
carrier carry: counts the differing elements.
--closed: the same run against (a) oracle_block_float_closed, the piecewise closed form of the reference's
double accumulators, and (b) the product's host half of GPSIQ_NCO_REFERENCE (gpsiq_reference_batch: fixed-point
samples + patches, applied here on the CPU): both must show 0 differing elements and the reference's carried
carr_phase after every block."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import _oracle  # noqa: E402
from gpsiq.abi import SC16  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
closed = "--closed" in sys.argv
fs = int(float(argv[0])) if len(argv) > 0 else 2600000
nchan = int(argv[1]) if len(argv) > 1 else 16
nb = int(argv[2]) if len(argv) > 2 else 299
ss = SC16
ns = fs // 10
o, r = _oracle.load_oracle(), _oracle.load_ref()
d = synth_blocks(nb, nchan, seed=20250215)
t0 = time.time()
ref_out, _, carr = r.run_blocks(d, fs, ss)
t_ref = time.time() - t0
print(f"fs={fs} nchan={nchan} blocks={nb} ({nb * ns} samples, {2 * nb * ns} int16 elements); reference loop {t_ref:.1f} s")

if closed:
    import gpsiq
    bad_closed = bad_carr = 0
    cin = d[0]["carr_phase"].copy()
    t0 = time.time()
    for b in range(nb):
        db = d[b].copy()
        db["carr_phase"] = cin
        out, cin = o.block_float_closed(db, ns, float(fs), ss)
        bad_closed += int(np.count_nonzero(out != ref_out[b * 2 * ns:(b + 1) * 2 * ns]))
        bad_carr += int(np.count_nonzero(cin != carr[b]))
    t_closed = time.time() - t0
    print(f"oracle_block_float_closed: differing elements {bad_closed}, carried carr_phase values that differ {bad_carr} ({t_closed:.1f} s)")
    t0 = time.time()
    q, patches, carr_end = gpsiq.reference_blocks(d, float(fs), ns)
    t_host = time.time() - t0
    bad_ref = 0
    for b in range(nb):
        out = o.block_fixed(q[b], ns, ss, seq=True)
        _oracle.apply_patches(o, q[b], out, patches[patches["block"] == b], ss)
        bad_ref += int(np.count_nonzero(out != ref_out[b * 2 * ns:(b + 1) * 2 * ns]))
    print(f"gpsiq_reference_batch: {len(patches)} patches, {t_host * 1e3:.0f} ms on {os.cpu_count()} CPUs ({t_host / nb * 1e3:.2f} ms per block); "
          f"fixed-point samples + patches: differing elements {bad_ref}; carr_phase after the run equal: {bool(np.array_equal(carr_end, carr[-1]))}")
    sys.exit(0 if bad_closed == 0 and bad_carr == 0 and bad_ref == 0 and np.array_equal(carr_end, carr[-1]) else 1)

q = o.quantize_blocks(d, fs, ns)
bad_blocks, bad_elems, maxabs = 0, 0, 0
for b in range(nb):
    fx = o.block_fixed(q[b], ns, ss, seq=True)
    diff = fx.astype(np.int32) - ref_out[b * 2 * ns:(b + 1) * 2 * ns].astype(np.int32)
    n = int(np.count_nonzero(diff))
    if n:
        bad_blocks += 1
        bad_elems += n
        maxabs = max(maxabs, int(np.abs(diff).max()))
# carrier: exact carry vs the reference's rounded accumulator at the end of the run
end_fixed = (q[-1]["carr_phase"].astype(object) + q[-1]["carr_step"].astype(object) * ns) % (1 << 59)
drift = [abs(float(int(end_fixed[c])) / 2 ** 59 - carr[-1][c]) for c in range(nchan)]
drift = [min(x, 1 - x) for x in drift]
print(f"T2: differing elements {bad_elems} in {bad_blocks} blocks ({bad_elems / (2 * nb * ns):.2e} of all elements), max |diff| {maxabs} LSB")
print(f"carrier phase after {nb * 0.1:.1f} s: exact carry vs reference accumulator, max |diff| = {max(drift):.3e} cycles")
