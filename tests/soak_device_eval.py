#!/usr/bin/env python3
"""Time-bounded random soak of the DEVICE evaluation of the batch calls (csrc/gpsiq_evaldev.cpp) on the GPU box: random whole
runs -- both NCO models, all rates, 1..400 blocks (the device path forced also where the library would not choose it), random
piece sizes, descriptors in pageable / page-locked / device memory, rough timelines (Doppler through zero, re-seeded and unused
slots, exact-tie addends: slots the host walker repairs), the run-time self-check on or off -- rendered through
GPSIQ_EVAL=device and GPSIQ_EVAL=host (rounds 4-5's path, itself soaked against the reference's own loop): every byte and the
carried phases must be equal; small runs are also held against the reference's own loop (oracle/_ref) directly.
Not part of the test suite.   usage: python tests/soak_device_eval.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import _oracle  # noqa: E402
import gpsiq  # noqa: E402
from gpsiq.abi import NCO_FIXED, NCO_REFERENCE  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402
import torch  # noqa: E402


def rough(d, rng):
    nb, nc = d.shape
    b = np.arange(nb)
    i = int(rng.integers(0, nc))
    d["f_carr"][:, i] = (b - nb * rng.uniform(0.2, 0.8)) * rng.uniform(0.3, 3.0) + rng.uniform(-0.02, 0.02, nb)     # through zero
    d["f_code"][:, i] = 1.023e6 + d["f_carr"][:, i] / 1540.0
    if nc > 2 and nb > 3:
        j = (i + 1) % nc
        d["prn"][int(rng.integers(1, nb)):, j] = 1 + (int(d["prn"][0, j]) % 32)                                         # another satellite
        k = (i + 2) % nc
        lo = int(rng.integers(0, nb))
        d["prn"][lo: lo + int(rng.integers(1, 12)), k] = 0                                                              # unused for a while
    if nc > 4 and rng.integers(0, 3) == 0:
        m = (i + 3) % nc
        d["f_carr"][:, m] = 2600000.0 / 1024.0                                                                          # an exact-tie addend at 2.6 Msps
        d["f_code"][:, m] = 1.023e6 + d["f_carr"][:, m] / 1540.0
    return d


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    ref = _oracle.load_ref()
    ctx = gpsiq.Context(0)
    t_end = time.time() + budget
    runs = blocks = vs_ref = bad = 0
    s0 = gpsiq.device_eval_stats()
    while time.time() < t_end:
        fs, ns = [(2.6e6, 26000), (2.6e6, 260000), (3.0e6, 30000), (10e6, 100000), (25e6, 250000), (2.6e6, 3333)][int(rng.integers(0, 6))]
        nb = int(rng.integers(1, 400)) if ns <= 30000 else int(rng.integers(1, 120))
        nc, ss = int(rng.integers(1, 17)), int(rng.integers(1, 3))
        mode = NCO_REFERENCE if rng.integers(0, 3) else NCO_FIXED
        d = synth_blocks(nb, nc, seed=int(rng.integers(0, 1 << 30)), doppler_hz=float(rng.choice([5000.0, 8000.0, 500.0, 60.0])))
        if rng.integers(0, 2):
            d = rough(d, rng)
        kind = int(rng.integers(0, 3))
        keep = None
        if kind == 0:
            src = d
        else:
            keep = torch.from_numpy(d.view(np.uint8).reshape(-1).copy())
            keep = keep.pin_memory() if kind == 1 else keep.cuda()
            src = (keep.data_ptr(), nb, nc)
        for k in ("GPSIQ_PIECE_BLOCKS", "GPSIQ_CHAIN_VERIFY", "GPSIQ_CHAIN"):
            os.environ.pop(k, None)
        if rng.integers(0, 2):
            os.environ["GPSIQ_PIECE_BLOCKS"] = str(int(rng.integers(1, nb + 1)))
        if rng.integers(0, 4) == 0:
            os.environ["GPSIQ_CHAIN_VERIFY"] = str(int(rng.integers(1, 9)))
        ctx.set_nco_mode(mode)
        os.environ["GPSIQ_EVAL"] = "device"
        ca = np.zeros(nc)
        a = ctx.generate_batch(src, ns, fs, ss, carr_out=ca)
        os.environ["GPSIQ_EVAL"] = "host"
        os.environ.pop("GPSIQ_CHAIN_VERIFY", None)
        cb = np.zeros(nc)
        b = ctx.generate_batch(d, ns, fs, ss, carr_out=cb)
        ok = np.array_equal(a, b) and ca.tobytes() == cb.tobytes()
        if ok and mode == NCO_REFERENCE and ref is not None and nb * ns <= 600000 and int(fs) == fs and ns * 10 == fs:
            want, _, carr_ref = ref.run_blocks(d, int(fs), ss, 1)
            ok = np.array_equal(a.reshape(-1), want) and np.array_equal(ca, carr_ref[-1])
            vs_ref += 1
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed} run {runs}: fs {fs} ns {ns} nb {nb} nc {nc} ss {ss} mode {mode} kind {kind} env "
                  f"{os.environ.get('GPSIQ_PIECE_BLOCKS')} {os.environ.get('GPSIQ_CHAIN_VERIFY')}", flush=True)
            np.save(os.path.join(ROOT, "gpurun_out", f"soak_device_eval_fail_{seed}_{runs}.npy"), d)
        del keep
        runs += 1
        blocks += nb * nc
    s1 = gpsiq.device_eval_stats()
    ctx.close()
    print(f"seed {seed}: {runs} runs either way ({vs_ref} of them also against the reference's own loop), {blocks} blocks x channels; device evaluation: "
          f"{s1[0] - s0[0]} calls, {s1[2] - s0[2]} pairs to the host walker, {s1[3] - s0[3]} slots repaired, {s1[4] - s0[4]} patches, "
          f"{s1[5] - s0[5]} fall-backs; {bad} mismatches", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
