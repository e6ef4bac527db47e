"""Synthetic channel descriptors for tests and bench.py.

Per channel: PRN = c+1; code_phase ~ U[0,1023); carr_phase ~ U[0,1);
f_carr ~ U[-5000,5000] Hz; f_code = 1.023e6 + f_carr/1540 (reference gps.c:2043);
iword ~ U{0..58}; ibit ~ U{0..29}; icode ~ U{0..19}; gain ~ U[0.3,1.0];
dwrd = 60 random 30-bit words.  RNG = SplitMix64, default seed 20250215.
"""
import numpy as np

from .abi import CHAN_DTYPE, N_DWRD

_M = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed=20250215):
        self.s = seed & _M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
        return z ^ (z >> 31)

    def uniform(self):            # [0,1) with 53 bits
        return (self.next() >> 11) * (1.0 / (1 << 53))

    def below(self, n):
        return self.next() % n


def synth_blocks(nblocks, nchan, seed=20250215, doppler_hz=5000.0, prn0=1, drift_hz=0.4):
    """[nblocks][nchan] descriptors.  Code-side state is re-drawn every block (the
    reference re-seeds it from pseudorange every 0.1 s, gps.c:2047-2055); Doppler
    random-walks by +-drift_hz per block; nav words persist per channel; carr_phase is
    drawn once (it is carried by the loop, gps.c:2821)."""
    rng = SplitMix64(seed)
    d = np.zeros((nblocks, nchan), dtype=CHAN_DTYPE)
    for c in range(nchan):
        dwrd = np.array([rng.next() & 0x3FFFFFFF for _ in range(N_DWRD)], dtype=np.uint32)
        f_carr = (rng.uniform() * 2.0 - 1.0) * doppler_hz
        carr = rng.uniform()
        gain = 0.3 + 0.7 * rng.uniform()
        for b in range(nblocks):
            e = d[b, c]
            e["prn"] = (prn0 - 1 + c) % 32 + 1
            e["iword"] = rng.below(59)
            e["ibit"] = rng.below(30)
            e["icode"] = rng.below(20)
            e["f_carr"] = f_carr
            e["f_code"] = 1.023e6 + f_carr / 1540.0
            e["carr_phase"] = carr
            e["code_phase"] = rng.uniform() * 1023.0
            e["gain"] = gain
            e["dwrd"] = dwrd
            f_carr += (rng.uniform() * 2.0 - 1.0) * drift_hz
    return d
