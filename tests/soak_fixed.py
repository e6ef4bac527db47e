#!/usr/bin/env python3
"""Soak of the fixed-point kernels against the oracle on the GPU box: random QUANTISED descriptors (full-range carrier
steps, code steps from 25 Msps-like up to the row kernels' limit, arbitrary fractions, nav bits, gains incl. the int16
wrap region), random block lengths incl. ragged rows / tiles / chunks, 1-16 channels, int8 and int16, every kernel
variant.  Not part of the test suite; prints one summary line.   usage: python tests/soak_fixed.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))   # _oracle: checkers live under tests/
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import torch  # noqa: E402
import _oracle  # noqa: E402
import gpsiq  # noqa: E402
from gpsiq.abi import QCHAN_DTYPE, SC08, SC16  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
oracle = _oracle.load_oracle()
ctx = gpsiq.Context(0)
rng = np.random.default_rng(int(time.time()))
variants = gpsiq.variants()
max_step = ((31 << 56) - 1) // 63
t_end = time.time() + budget
cases = launches = 0
elems = 0
while time.time() < t_end:
    nb, nc = int(rng.integers(1, 4)), int(rng.integers(1, 17))
    ns = int(rng.choice([rng.integers(1, 300), rng.integers(300, 40000), rng.integers(40000, 300000)]))
    q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
    q["prn"] = rng.integers(0, 33, size=(nb, nc))
    q["carr_phase"] = rng.integers(0, 1 << 59, size=(nb, nc), dtype=np.uint64)
    span = int(rng.choice([1 << 58, 1 << 51, 1 << 40]))
    q["carr_step"] = rng.integers(-span + 1, span, size=(nb, nc))
    q["code_frac"] = rng.integers(0, 1 << 56, size=(nb, nc), dtype=np.uint64)
    room = ((30 * 20 * 1023) << 56) // ns                          # stay inside the 32 nav bits a descriptor carries
    hi = int(min(max_step, room, int(rng.choice([max_step, 1 << 55, 1 << 52]))))
    q["code_step"] = rng.integers(1, hi + 1, size=(nb, nc), dtype=np.uint64)
    q["chip0"] = rng.integers(0, 1023, size=(nb, nc))
    q["icode"] = rng.integers(0, 20, size=(nb, nc))
    q["nav_bits"] = rng.integers(0, 1 << 32, size=(nb, nc), dtype=np.uint64).astype(np.uint32)
    q["gain"] = rng.choice([0.0, -0.7, 1.0, 0.3333, 2.0, 1e-3, 8.19, 8.3, -131.0, 17.25], size=(nb, nc))
    ctx.set_descriptors(q)
    for ss in (SC08, SC16):
        want = np.stack([oracle.block_fixed(q[b], ns, ss) for b in range(nb)])
        blk = 2 * ns * ss
        stride = (blk + 15) & ~15
        for name, vid in variants.items():
            buf = torch.full((nb * stride + 64,), 0x5A, dtype=torch.uint8, device="cuda")
            ctx.launch(0, nb, ns, ss, buf.data_ptr(), stride, stream=torch.cuda.current_stream().cuda_stream, variant=vid)
            torch.cuda.synchronize()
            host = buf.cpu().numpy()
            rows = host[: nb * stride].reshape(nb, stride)
            got = np.ascontiguousarray(rows[:, :blk]).view(np.int8 if ss == SC08 else np.int16)
            ok = np.array_equal(got, want) and (host[nb * stride:] == 0x5A).all() and (stride == blk or (rows[:, blk:] == 0x5A).all())
            if not ok:
                print("MISMATCH", name, ss, nb, nc, ns)
                np.save(os.path.join(ROOT, "gpurun_out", "soak_fixed_fail_q.npy"), q)
                sys.exit(1)
            launches += 1
            elems += nb * 2 * ns
    cases += 1
print(f"soak ok: {cases} random descriptor sets, {launches} launches over {len(variants)} variants x int8/int16, "
      f"{elems / 1e9:.2f} G output elements, all equal to the oracle's closed form, nothing written outside the blocks ({budget:.0f} s)")
