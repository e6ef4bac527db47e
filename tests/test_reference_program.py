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


def test_the_wholly_unmodified_reference_writes_the_captured_blocks(tmp_path):
    """The captures come from oracle/_ref/gps-sim-ref*, which are the reference's sources linked with THIS repository's
    host/fifo.c behind the reference's fifo.h -- the one substitution under the oracle pin, made because the reference's own
    FIFO drops blocks (fifo.c:166-168 never advances the tail: what is enqueued while two buffers wait overwrites the
    newest).  Here the reference with NOTHING substituted (oracle/_ref/gps-sim-ref-ownfifo: its own fifo.c:33-205 too, real
    ncurses / curl / zlib) runs the same 30 s: every block it writes must be one of the committed capture's blocks, in
    order, with only the blocks its FIFO loses missing (1-6 of 299 on every machine tried: the producer outruns the file
    sink while the queue fills).  So the substitution changes no sample, and the fixtures are the reference's bytes."""
    from _program import run_program_lossy
    ref = program("gps-sim-ref-ownfifo")
    if ref is None:
        pytest.skip("oracle/_ref/gps-sim-ref-ownfifo not built (no /root/reference here)")
    want = [str(s) for s in np.load(GOLD)["sha8"]]
    where = {h: b for b, h in enumerate(want)}
    assert len(where) == 299, "two captured blocks with one digest: the mapping below would be ambiguous"
    got = block_digests(run_program_lossy(ref, str(tmp_path), 30, False), False)
    idx = [where.get(h, -1) for h in got]
    assert -1 not in idx, f"block {idx.index(-1)} of the unmodified program's file is not in the capture"
    assert idx == sorted(set(idx)), "blocks out of order or written twice"
    missing = sorted(set(range(299)) - set(idx))
    print("unmodified reference: %d of 299 blocks written, its FIFO lost blocks %s" % (len(idx), missing))
    assert idx[0] == 0 and idx[-1] == 298 and 1 <= len(missing) <= 24 and missing[-1] < 40, missing
    assert missing == list(range(1, 7)) or os.environ.get("GPSIQ_ANY_FIFO_LOSS"), missing     # always blocks 1-6


def test_the_fixed_point_models_differing_blocks_are_the_committed_list(oracle):
    """Tier T2 exact (CPU half): the list `fixed_differing_blocks_sha8` the GPU test below compares against is what the
    checker computes -- host chain -> exact carrier carry -> the oracle's closed form, block by block, SHA-256 against the
    capture of the reference program (tests/golden/make_golden.py --t2-only)."""
    import gpsiq
    from _program import LLH, RINEX
    from gpsiq.abi import SC08
    from gpsiq.pipeline import RunAheadAllocating
    z = np.load(GOLD)
    want = [str(s) for s in z["sha8"]]
    eph, utc, n = gpsiq.rinex_read(RINEX, 2)
    sv = int(np.nonzero(eph[0]["vflg"])[0][0])
    week, sec = int(eph[0, sv]["toc_week"]), float(eph[0, sv]["nav"]["toc_sec"])
    lat, lon, h = (float(v) for v in LLH.split(","))
    xyz = np.tile(gpsiq.llh_to_ecef(lat / 57.2957795131, lon / 57.2957795131, h), (300, 1))
    desc = RunAheadAllocating(eph[:n], utc, 12, week, sec, xyz[0], ieph=gpsiq.rinex_select(eph, n, week, sec)).descriptors(xyz[1:])
    q, _ = gpsiq.quantize_blocks(desc, float(FS), FS // 10)
    differing = [b for b in range(299) if hashlib.sha256(oracle.block_fixed(q[b], FS // 10, SC08, seq=True).tobytes()).hexdigest() != want[b]]
    assert differing == [int(b) for b in z["fixed_differing_blocks_sha8"]] == [125, 138, 172, 175]


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
    # exactly the blocks in which the oracle's closed form with the exact carry differs from the capture
    # (tests/golden/make_golden.py --t2-only; deterministic: a changed list is a changed model)
    differing = [b for b in range(299) if got[b] != want[b]]
    assert differing == [int(b) for b in z["fixed_differing_blocks_sha8"]] == [125, 138, 172, 175]


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
