"""BASELINE configs 3 and 5 against the reference's bytes.  Config 3: "--iq16 @10 Msps, 16 channels, 300 s" = 2 999 blocks of
10^6 samples (12 GB); config 5: "8xMI355X time-sharded, 25 Msps int16, 16 channels, 3600 s" -- the first GPU's share, 450 s =
4 499 blocks of 2.5 * 10^6 samples (45 GB).  tests/golden/program_config35_static.npz holds the SHA-256 of every block the
reference program writes when rebuilt at those sample rates (oracle/_ref/gps-sim-ref-10M / -25M: TX_SAMPLERATE and MAX_CHAN
are compile-time constants of the reference, sdr.h:21, gps.h:36) at the static BASELINE position with 16 satellites in view
(tests/golden/synth_static16.21n), made by tests/golden/make_golden.py --config35-only.  On the GPU the same program with its
gps thread on libgpsiq (GPSIQ_NCO_REFERENCE) writes those bytes, every block; so does the library's own host chain
(host/gpsiq_runahead.c).  The streams go through a named pipe and are hashed as they arrive."""
import os
import subprocess

import numpy as np
import pytest

import gpsiq
from _program import LLH, RINEX16, ROOT, program, program_block_digests
from test_config4 import start_time

GOLD = os.path.join(ROOT, "tests", "golden", "program_config35_static.npz")
HOST = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "host")
CASES = {"cfg3": ("10M", 10000000, 300, 2999), "cfg5": ("25M", 25000000, 450, 4499)}


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return {name: {"sha": [str(s) for s in z[name + "_sha16"]],
                   "heads": {int(b): h for b, h in zip(z[name + "_head_blocks"], z[name + "_heads"])}} for name in CASES}


@pytest.mark.parametrize("name", sorted(CASES))
def test_unpatched_program_reproduces_the_capture(gold, tmp_path, name):
    """Pins the fixture: the first 5 / 3 s here (a shorter -d does not change the blocks it does render), the whole capture
    by make_golden.py."""
    suffix, fs, _, _ = CASES[name]
    ref = program("gps-sim-ref-" + suffix)
    if ref is None:
        pytest.skip(f"oracle/_ref/gps-sim-ref-{suffix} not built (no /root/reference here)")
    secs = 5 if fs <= 10000000 else 3
    sha, heads = program_block_digests(ref, str(tmp_path), None, secs, secs * 10 - 1, fs=fs, keep=(0, 1))
    assert sha == gold[name]["sha"][:secs * 10 - 1]
    assert np.array_equal(heads[0], gold[name]["heads"][0]) and np.array_equal(heads[1], gold[name]["heads"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_thread_on_the_gpu_at_full_length(gold, tmp_path, name):
    """The reference program with its sample loop replaced by gpsiq_generate_block on the GPU, `-r iqfile --iq16 -d 300`
    at 10 Msps / `-d 450` at 25 Msps, 16 channels: the bytes of the unpatched program, every block."""
    suffix, fs, seconds, nblocks = CASES[name]
    patched = program("gps-sim-gpsiq-" + suffix)
    assert patched is not None, f"oracle/_ref/gps-sim-gpsiq-{suffix} missing: run __graft_entry__.build() where /root/reference exists"
    sha, heads = program_block_digests(patched, str(tmp_path), None, seconds, nblocks, fs=fs, keep=tuple(gold[name]["heads"]), timeout=1500)
    bad = [b for b in range(nblocks) if sha[b] != gold[name]["sha"][b]]
    assert len(sha) == nblocks and not bad, f"{len(bad)} blocks differ from the reference program's output, first {bad[:10]}"
    for b, h in heads.items():
        assert np.array_equal(h, gold[name]["heads"][b]), b


@pytest.mark.gpu
def test_config3_run_ahead_chain_reference_nco(gold, tmp_path):
    """Config 3 with nothing of the reference in the process: host/gpsiq_runahead.c (RINEX reader, allocation, navigation
    words, batched refresh, gpsiq_generate_batch 100 blocks per call) in GPSIQ_NCO_REFERENCE == the reference program's
    file, all 2 999 blocks."""
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    eph, _, _ = gpsiq.rinex_read(RINEX16, 2)
    week, sec = start_time(eph)
    args = [os.path.join(HOST, "gpsiq_runahead"), RINEX16, "2", str(week), repr(sec), LLH, "2999", "16", repr(1.0e7), "2", "iq.bin"]
    sha = []

    def on_block(i, b):
        import hashlib
        sha.append(hashlib.sha256(b).hexdigest())
    from _program import stream_blocks
    stream_blocks(args, str(tmp_path), "iq.bin", 1000000 * 4, 2999, {"GPSIQ_NCO": "reference"}, timeout=1200, idles=False, on_block=on_block)
    bad = [b for b in range(2999) if sha[b] != gold["cfg3"]["sha"][b]]
    assert not bad, f"{len(bad)} blocks differ from the reference program's output, first {bad[:10]}"
