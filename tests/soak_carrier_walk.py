"""Time-bounded random soak of the carrier chain's wrap-to-wrap table (csrc/gpsiq_exact.cpp, NcoWalk) against the
plain loop of gps.c:2821-2826 (oracle_carrier_chain).  CPU only.   python tests/soak_carrier_walk.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "multi-sdr-gps-sim_amd"))
import _oracle  # noqa: E402
import gpsiq  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    o = _oracle.load_oracle()
    t0, runs, cycles = time.time(), 0, 0.0
    while time.time() - t0 < budget:
        fs = float(rng.choice([2.6e6, 3e6, 10e6, 25e6, 2.048e6]))
        ns = int(rng.choice([int(fs) // 10, int(rng.integers(1, int(fs) // 5))]))
        nb = int(rng.integers(1, 4))
        d = synth_blocks(nb, 16, seed=int(rng.integers(1 << 30)))
        kind = rng.integers(5)
        mag = 10.0 ** rng.uniform(1.0, np.log10(0.03 * fs), 16) if kind == 0 else rng.uniform(100.0, 9000.0, 16)
        f = mag * rng.choice([-1.0, 1.0], 16)
        if kind == 1:
            f[:8] = fs * rng.integers(1, 64, 8) * 2.0 ** -rng.integers(10, 40, 8) * rng.choice([-1.0, 1.0], 8)
        if kind == 2:
            c = rng.integers(1, 64, 16) * 2.0 ** -rng.integers(10, 30, 16) * rng.choice([-1.0, 1.0], 16)
            for _ in range(int(rng.integers(1, 4))):
                c = np.nextafter(c, rng.choice([-1.0, 1.0], 16))
            f = c * fs
        d["f_carr"] = f[None, :] + np.cumsum(rng.uniform(-1.0, 1.0, (nb, 16)), axis=0) * (kind != 1)
        d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
        x0 = rng.uniform(0.0, 1.0, 16)
        x0 = np.where(rng.random(16) < 0.2, rng.integers(0, 2 ** 44, 16) * 2.0 ** -52, x0)
        x0 = np.where(rng.random(16) < 0.2, 1.0 - rng.integers(1, 2 ** 44, 16) * 2.0 ** -53, x0)
        x0 = np.where(rng.random(16) < 0.2, 2.0 ** -rng.integers(1, 40, 16).astype(np.float64) * (1.0 + rng.integers(-2, 3, 16) * 2.0 ** -52), x0)
        d["carr_phase"][0] = np.clip(x0, 0.0, np.nextafter(1.0, 0.0))
        _, _, got = gpsiq.reference_blocks(d, fs, ns)
        x = d["carr_phase"][0].copy()
        for b in range(nb):
            x = np.array([o.carrier_chain(x[i], d["f_carr"][b, i] * (1.0 / fs), ns) for i in range(16)])
            cycles += float(np.abs(d["f_carr"][b]).sum()) / fs * ns
        if got.tobytes() != x.tobytes():
            bad = got != x
            print("MISMATCH", fs, ns, nb, d["f_carr"][:, bad], d["carr_phase"][0][bad], got[bad], x[bad])
            sys.exit(1)
        runs += 1
    print("%d runs, %.3g carrier cycles walked, %.0f s: the carried phase equals the plain loop's in every one" % (runs, cycles, time.time() - t0))


if __name__ == "__main__":
    main()
