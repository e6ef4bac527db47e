"""The device evaluation of GPSIQ_NCO_REFERENCE on the CPU: csrc/gpsiq_eval.h is the code the kernels of gpsiq_eval_kernels.hip
run (one lane per block and channel: quantiser, Euclid descent for the candidate samples, drift enclosure, patch emission; the
carrier chain's level 2 as a scan), compiled here for the host and held against what it replaces -- quantize_one, candidates(),
eval_block, gpsiq_chain_link -- on random and adversarial descriptors (tests/eval_twin.cpp).  The GPU tests
(test_gpu_device_eval.py) then run the kernels through the C-ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc")


def build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++17", "-ffp-contract=off", *flags, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", exe,
                    os.path.join(ROOT, "tests", "eval_twin.cpp"), os.path.join(CSRC, "gpsiq_host.cpp"), "-lpthread", "-lm"], check=True)
    return exe


def test_lane_code_equals_the_host_evaluation(tmp_path):
    exe = build(tmp_path, "eval_twin", ["-O2"])
    for seed in (1, 2, 3):
        r = subprocess.run([exe, str(seed), "60"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and " bad=0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        f = dict(kv.split("=") for kv in r.stdout.strip().splitlines()[-1].split())
        assert int(f["patches"]) > 50 and int(f["known"]) > 10000        # the comparison had something to compare
        # what the lanes hand to the host walker when the descriptor is seeded from the start state or a usable estimate of it: the
        # adversarial sixth of the cases (a start state or a code phase ON a boundary) and a start of exactly 1.0 -- not the rule
        # (estimates 1e-6 .. 0.3 cycle off, a quarter of the cases, go there by design: every sample is a candidate)
        assert int(f["host_near"]) < 0.15 * int(f["evals_near"])


def test_lane_code_under_asan_and_ubsan(tmp_path):
    """The same program with -fsanitize=address,undefined (no recovery): the 128-bit arithmetic, shifts, the descent's explicit
    stack, the nav-bit window -- the code the kernels run, where a sanitizer can see it."""
    exe = build(tmp_path, "eval_twin_san", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = subprocess.run([exe, "7", "12"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " bad=0" in r.stdout and "runtime error" not in r.stderr, r.stdout[-2000:] + r.stderr[-2000:]
