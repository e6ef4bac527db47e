#!/usr/bin/env python3
"""Tier T2 report: a whole BASELINE config-1 length run (299 blocks = 29.9 s)
through the reference's own loop (carrier phase carried by its double accumulator) against
the fixed-point oracle with the library's exact carrier carry.  CPU only; needs oracle/_ref.
Usage: python tests/t2_report.py [fs] [nchan] [nblocks]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import _oracle  # noqa: E402
from gpsiq.abi import SC08, SC16  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402

fs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2600000
nchan = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 299
ss = SC16
ns = fs // 10
o, r = _oracle.load_oracle(), _oracle.load_ref()
d = synth_blocks(nb, nchan, seed=20250215)
t0 = time.time()
ref_out, _, carr = r.run_blocks(d, fs, ss)
t_ref = time.time() - t0
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
print(f"fs={fs} nchan={nchan} blocks={nb} ({nb * ns} samples, {2 * nb * ns} int16 elements); reference loop {t_ref:.1f} s")
print(f"T2: differing elements {bad_elems} in {bad_blocks} blocks ({bad_elems / (2 * nb * ns):.2e} of all elements), max |diff| {maxabs} LSB")
print(f"carrier phase after {nb * 0.1:.1f} s: exact carry vs reference accumulator, max |diff| = {max(drift):.3e} cycles")
