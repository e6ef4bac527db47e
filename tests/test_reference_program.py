"""The drop-in inside the reference's own thread.  oracle/_ref/gps-sim-gpsiq is the reference program
(gps-sim.c, sdr.c, sdr_iqfile.c, almanac.c, gui.c and gps.c itself) with gps_thread_ep()'s sample loop
+ pack + fifo hand-off (gps.c:2767-2865) cut out and replaced by gpsiq_generate_block() and the chunker,
exactly as INTEGRATION.md section 2 describes (oracle/patch_gps_thread.py applies it at build time);
oracle/_ref/gps-sim-ref is the same program unpatched.  Both run headless on the same RINEX file and
must write the same iqdata.bin, byte for byte."""
import hashlib
import os

import numpy as np
import pytest

from _program import FS, RINEX16, program, run_program, write_circle_motion

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "program_static_30s.npz")


def block_digests(data, iq16, fs=FS):
    blk = (fs // 10) * 2 * (2 if iq16 else 1)
    return [hashlib.sha256(data[i:i + blk]).hexdigest() for i in range(0, len(data), blk)]


@pytest.mark.parametrize("iq16", [False, True])
def test_unpatched_program_reproduces_its_committed_capture(tmp_path, iq16):
    """Pins the fixture (tests/golden/make_golden.py --program-only): the reference program as shipped,
    30 s, static position, int8 and --iq16."""
    ref = program("gps-sim-ref")
    if ref is None:
        pytest.skip("oracle/_ref/gps-sim-ref not built (no /root/reference here)")
    z = np.load(GOLD)
    data = run_program(ref, str(tmp_path), 30, iq16)
    assert block_digests(data, iq16) == [str(s) for s in z["sha16" if iq16 else "sha8"]]


@pytest.mark.gpu
@pytest.mark.parametrize("iq16", [False, True])
def test_patched_reference_thread_writes_the_same_file(tmp_path, iq16):
    """gps_thread_ep() producing its bytes through gpsiq_generate_block (GPSIQ_NCO_REFERENCE) on the GPU ==
    the committed capture of the unpatched program, all 299 blocks; and == the unpatched program run here."""
    patched = program("gps-sim-gpsiq")
    assert patched is not None, "oracle/_ref/gps-sim-gpsiq missing: run __graft_entry__.build() where /root/reference exists"
    z = np.load(GOLD)
    data = run_program(patched, str(tmp_path), 30, iq16)
    got = block_digests(data, iq16)
    want = [str(s) for s in z["sha16" if iq16 else "sha8"]]
    assert len(got) == 299
    bad = [b for b in range(299) if got[b] != want[b]]
    assert not bad, f"blocks {bad[:10]} differ from the reference program's output"
    ref = program("gps-sim-ref")
    if ref is not None:
        os.makedirs(tmp_path / "ref")
        assert run_program(ref, str(tmp_path / "ref"), 30, iq16) == data


@pytest.mark.gpu
def test_patched_reference_thread_fixed_point_model(tmp_path):
    """The same program in the library's default NCO model (GPSIQ_NCO=fixed): everything but a handful of
    elements per 10^8 equals the reference program's file (tier T2)."""
    patched = program("gps-sim-gpsiq")
    assert patched is not None
    z = np.load(GOLD)
    data = run_program(patched, str(tmp_path), 30, False, {"GPSIQ_NCO": "fixed"})
    got = block_digests(data, False)
    want = [str(s) for s in z["sha8"]]
    assert sum(g != w for g, w in zip(got, want)) <= 40           # blocks that hold a differing element


def test_unpatched_program_moving_receiver_capture(tmp_path):
    """The fixture's third capture: the reference program on a user-motion file (a circle, this repository's own
    track in the reference's CSV format), --iq16."""
    ref = program("gps-sim-ref")
    if ref is None:
        pytest.skip("oracle/_ref/gps-sim-ref not built (no /root/reference here)")
    z = np.load(GOLD)
    data = run_program(ref, str(tmp_path), 30, True, motion=write_circle_motion(str(tmp_path / "circle.csv"), 30))
    assert block_digests(data, True) == [str(s) for s in z["sha16_circle"]]


@pytest.mark.gpu
def test_patched_reference_thread_moving_receiver(tmp_path):
    """BASELINE config 4's kind of scenario inside the reference's own thread: a moving receiver (user-motion file), the
    reference's host model refreshing range and Doppler for every 0.1 s block, every block synthesised by
    gpsiq_generate_block on the GPU: the same iqdata.bin as the unpatched program, all 299 blocks, int16."""
    patched = program("gps-sim-gpsiq")
    assert patched is not None
    z = np.load(GOLD)
    data = run_program(patched, str(tmp_path), 30, True, motion=write_circle_motion(str(tmp_path / "circle.csv"), 30))
    got, want = block_digests(data, True), [str(s) for s in z["sha16_circle"]]
    bad = [b for b in range(299) if got[b] != want[b]]
    assert len(got) == 299 and not bad, f"blocks {bad[:10]} differ from the reference program's output"


def test_unpatched_program_at_the_baseline_constants(tmp_path):
    """BASELINE config 1 as the reference itself renders it: the reference program rebuilt with TX_SAMPLERATE 2600000
    and MAX_CHAN 16 (its two compile-time constants, oracle/Makefile), 16 satellites in view, int8, 30 s."""
    ref = program("gps-sim-ref-2M6")
    if ref is None:
        pytest.skip("oracle/_ref/gps-sim-ref-2M6 not built (no /root/reference here)")
    z = np.load(GOLD)
    data = run_program(ref, str(tmp_path), 30, False, fs=2600000, rinex=RINEX16)
    assert block_digests(data, False, 2600000) == [str(s) for s in z["sha8_2M6_16ch"]]


@pytest.mark.gpu
def test_patched_reference_thread_at_the_baseline_constants(tmp_path):
    """BASELINE config 2: the same static scenario on the GPU -- 2.6 Msps, int8, 16 channels -- bit-exact against the
    CPU iqfile, with BOTH sides being the reference program: its gps thread on libgpsiq against its own loop."""
    patched = program("gps-sim-gpsiq-2M6")
    assert patched is not None
    z = np.load(GOLD)
    data = run_program(patched, str(tmp_path), 30, False, fs=2600000, rinex=RINEX16)
    got, want = block_digests(data, False, 2600000), [str(s) for s in z["sha8_2M6_16ch"]]
    bad = [b for b in range(299) if got[b] != want[b]]
    assert len(got) == 299 and not bad, f"blocks {bad[:10]} differ from the reference program's output"


def test_unpatched_program_across_nav_refreshes(tmp_path):
    """65 s of the reference program: its 30 s refreshes (navigation-message roll, allocateChannel) happen twice."""
    ref = program("gps-sim-ref")
    if ref is None:
        pytest.skip("oracle/_ref/gps-sim-ref not built (no /root/reference here)")
    z = np.load(GOLD)
    assert block_digests(run_program(ref, str(tmp_path), 65, False), False) == [str(s) for s in z["sha8_65s"]]


@pytest.mark.gpu
def test_patched_reference_thread_across_nav_refreshes(tmp_path):
    """The patched thread over 649 blocks incl. two of the reference's own 30 s refreshes: the same file."""
    patched = program("gps-sim-gpsiq")
    assert patched is not None
    z = np.load(GOLD)
    got = block_digests(run_program(patched, str(tmp_path), 65, False), False)
    want = [str(s) for s in z["sha8_65s"]]
    bad = [b for b in range(len(want)) if got[b] != want[b]]
    assert len(got) == 649 and not bad, f"blocks {bad[:10]} differ from the reference program's output"
