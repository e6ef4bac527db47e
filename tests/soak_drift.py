#!/usr/bin/env python3
"""Time-bounded random soak of the drift enclosure END TO END: gpsiq_reference_batch with the candidates decided from the
block's start state (the default) against the same call with every candidate's accumulators walked (GPSIQ_NO_DRIFT=1, the
round-3 path, itself soaked against the float loop by tests/soak_reference_host.py) -- descriptors, patches and carried phase
must be identical.  Sample rates and lengths with about one candidate per block and channel (10 - 25 Msps), phases pushed next
to LUT / chip boundaries so that many candidates are real patches.  CPU only.  usage: python tests/soak_drift.py seed seconds"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "multi-sdr-gps-sim_amd"))
import numpy as np  # noqa: E402
import gpsiq  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402


def one(rng):
    fs = float(rng.choice([10e6, 16e6, 25e6, 25e6, 2.6e6]))
    ns = int(fs) // 10 if rng.random() < 0.7 else int(rng.integers(int(fs) // 40, int(fs) // 10))
    nb, nc = int(rng.integers(2, 9)), int(rng.integers(1, 17))
    d = synth_blocks(nb, nc, seed=int(rng.integers(1 << 30)))
    d["f_carr"] = d["f_carr"][0][None, :] + np.cumsum(rng.uniform(-0.3, 0.3, (nb, nc)), axis=0)
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    if rng.random() < 0.5:      # start states a hair off a LUT step / a chip: candidates at the very start, many of them patches
        d["carr_phase"][0] = ((rng.integers(0, 512, nc) + rng.choice([1e-13, -1e-13, 3e-12, 0.0], nc)) / 512.0) % 1.0
        d["code_phase"][:] = (rng.integers(0, 1023, (nb, nc)) + rng.choice([1e-10, 2e-9, 0.0, 1.0 - 1e-10], (nb, nc))) % 1023.0
    os.environ.pop("GPSIQ_NO_DRIFT", None)
    a = gpsiq.reference_blocks(d, fs, ns)
    os.environ["GPSIQ_NO_DRIFT"] = "1"
    b = gpsiq.reference_blocks(d, fs, ns)
    os.environ.pop("GPSIQ_NO_DRIFT", None)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes(), (fs, ns, nb, nc)
    return len(a[1])


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    t0, runs, npatch = time.time(), 0, 0
    s0 = gpsiq.reference_stats()
    while time.time() - t0 < budget:
        npatch += one(rng)
        runs += 1
    s1 = gpsiq.reference_stats()
    print("%d runs, %d patches: identical with and without the drift enclosure; candidate states %d, of which %d were decided from the start "
          "state in the default runs" % (runs, npatch, (s1[0] - s0[0]) // 2, s1[1] - s0[1]))


if __name__ == "__main__":
    main()
