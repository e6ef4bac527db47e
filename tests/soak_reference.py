#!/usr/bin/env python3
"""Soak of GPSIQ_NCO_REFERENCE against the reference's own loop (oracle/_ref/libgpsref.so) on the GPU box: random
whole runs at several rates, every element and the carried phase compared.  Not part of the test suite (minutes of
host CPU for the reference loop); prints one summary line.   usage: python tests/soak_reference.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))   # _oracle: checkers live under tests/
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import _oracle  # noqa: E402
import gpsiq  # noqa: E402
from gpsiq.abi import NCO_REFERENCE  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
ref = _oracle.load_ref()
assert ref is not None, "needs oracle/_ref/libgpsref.so"
ctx = gpsiq.Context(0)
ctx.set_nco_mode(NCO_REFERENCE)
seed = int(os.environ.get("GPSIQ_SOAK_SEED", str(int(time.time()))))
rng = np.random.default_rng(seed)
t_end = time.time() + budget
runs = blocks = samples = patches = 0
while time.time() < t_end:
    fs = int(rng.choice([2600000, 3000000, 10000000, 25000000]))
    nb = int(rng.integers(2, 9)) if fs >= 10000000 else int(rng.integers(5, 40))
    nc, ss = int(rng.integers(4, 17)), int(rng.integers(1, 3))
    d = synth_blocks(nb, nc, seed=int(rng.integers(0, 1 << 30)), doppler_hz=float(rng.choice([5000.0, 8000.0, 500.0])))
    want, _, carr_ref = ref.run_blocks(d, fs, ss, 1)
    carr = np.zeros(nc)
    got = ctx.generate_batch(d, fs // 10, float(fs), ss, carr_out=carr)
    if not np.array_equal(got.reshape(-1), want) or not np.array_equal(carr, carr_ref[-1]):
        print("MISMATCH seed", seed, fs, nb, nc, ss, int((got.reshape(-1) != want).sum()))
        np.save(os.path.join(ROOT, "gpurun_out", "soak_fail_desc.npy"), d)
        sys.exit(1)
    _, p, _ = gpsiq.reference_blocks(d, float(fs), fs // 10)
    runs += 1
    blocks += nb
    samples += nb * (fs // 10)
    patches += len(p)
print(f"soak ok (seed {seed}): {runs} runs, {blocks} blocks, {samples / 1e9:.2f} G samples x up to 16 channels, {patches} patched samples, "
      f"all equal to the reference's own loop incl. the carried carr_phase ({budget:.0f} s)")
