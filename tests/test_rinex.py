"""RINEX 2/3 navigation readers (gpsiq_rinex_read) against the
reference's readRinex2/readRinex3 lines on synthetic files (the reference ships none), and
the whole chain RINEX -> nav words + host refresh -> descriptors against the reference."""
import os

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import NAV_STATE_DTYPE, SC08, TRACK_DTYPE
from gpsiq.scenario import llh_to_ecef, synth_rinex_records, write_rinex_nav

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOKYO = llh_to_ecef(35.681298, 139.766247, 10.0)
WEEK, SEC = 2190, 270000.0
UTC = dict(alpha=[0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06], beta=[0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07],
           A0=-0.931322574615e-09, A1=-0.355271367880e-14, tot=233472, wnt=2190, dtls=18)


def same(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize("version", [2, 3])
@pytest.mark.parametrize("gz", [False, True])
def test_reader_matches_reference(ref, tmp_path, version, gz):
    recs = synth_rinex_records(12, TOKYO, WEEK, SEC, seed=version, sets=3)
    path = write_rinex_nav(str(tmp_path / ("nav.%dn%s" % (version, ".gz" if gz else ""))), recs, UTC, version, gzip_it=gz)
    eph, utc, n = gpsiq.rinex_read(path, version)
    reph, rutc, rn = ref.read_rinex(path, version)
    assert n == rn == 3
    assert same(eph, reph) and same(np.asarray(utc), np.asarray(rutc))
    assert utc["vflg"] == 1 and eph["vflg"][:3, :12].all() and not eph["vflg"][:, 12:].any()
    assert eph["svh"][0, 3] == 37                      # health 5 -> MSB set (gps.c:1463-1464)
    # the set the simulator would pick for a start inside the first / second issue
    assert gpsiq.rinex_select(eph, n, WEEK, float(recs[0]["toc_sec"]) + 100.0) == 0
    assert gpsiq.rinex_select(eph, n, WEEK, float(recs[12]["toc_sec"]) + 100.0) == 1
    assert gpsiq.rinex_select(eph, n, WEEK, SEC) == 1       # toc 266400: dt = 3600 is not < 1 h (gps.c:2594)
    assert gpsiq.rinex_select(eph, n, WEEK + 1, SEC) == -1


def test_reader_edge_cases_match_reference(ref, tmp_path):
    recs = synth_rinex_records(5, TOKYO, WEEK, SEC, seed=9, sets=1)
    # no iono/UTC records -> vflg 0; wrong-version and wrong-type files -> the reference's codes
    p = write_rinex_nav(str(tmp_path / "noutc.n"), recs, None, 2)
    eph, utc, n = gpsiq.rinex_read(p, 2)
    reph, rutc, rn = ref.read_rinex(p, 2)
    assert n == rn == 1 and utc["vflg"] == 0 and same(eph, reph)
    p3 = write_rinex_nav(str(tmp_path / "v3.n"), recs, UTC, 3)
    assert gpsiq.rinex_read(p3, 2)[2] == ref.read_rinex(p3, 2)[2] == -2
    assert gpsiq.rinex_read(p, 3)[2] == ref.read_rinex(p, 3)[2] == -2
    assert gpsiq.rinex_read(str(tmp_path / "missing.n"), 2)[2] == ref.read_rinex(str(tmp_path / "missing.n"), 2)[2] == -1
    txt = open(p).read().replace("N: GPS NAV DATA", "O: OBSERVATION ")
    open(str(tmp_path / "obs.n"), "w").write(txt)
    assert gpsiq.rinex_read(str(tmp_path / "obs.n"), 2)[2] == ref.read_rinex(str(tmp_path / "obs.n"), 2)[2] == -3
    # more than 13 hourly sets: the reader stops like the reference
    many = synth_rinex_records(2, TOKYO, WEEK, 0.0, seed=4, sets=15)
    pm = write_rinex_nav(str(tmp_path / "many.n"), many, UTC, 2)
    a, _, na = gpsiq.rinex_read(pm, 2)
    b, _, nb = ref.read_rinex(pm, 2)
    assert nb == 14 and na == 13 and same(a, b)       # the reference reports one set more than its array holds; the library clamps (documented)
    # a truncated last record is dropped (vflg stays 0)
    lines = open(p).read().splitlines(True)
    open(str(tmp_path / "cut.n"), "w").write("".join(lines[:-3]))
    a, _, na = gpsiq.rinex_read(str(tmp_path / "cut.n"), 2)
    b, _, nb = ref.read_rinex(str(tmp_path / "cut.n"), 2)
    assert na == nb and same(a, b) and a["vflg"][0].sum() == 4


def test_golden_rinex_capture():
    """Committed synthetic RINEX file + the reference reader's output for it."""
    from gpsiq.abi import NAV_UTC_DTYPE, RINEX_EPH_DTYPE
    z = np.load(os.path.join(GOLD, "rinex_parsed.npz"))
    eph, utc, n = gpsiq.rinex_read(os.path.join(GOLD, "synth_static.21n"), 2)
    assert n == int(z["nsets"]) == 2
    assert eph.tobytes() == np.ascontiguousarray(z["eph"]).view(RINEX_EPH_DTYPE).tobytes()
    assert np.asarray(utc).tobytes() == np.ascontiguousarray(z["utc"]).view(NAV_UTC_DTYPE).tobytes()


def chain_from_rinex(path, version, nblocks=100):
    """RINEX -> (per channel) subframes -> nav words, track init, host refresh: descriptors."""
    eph, utc, n = gpsiq.rinex_read(path, version)
    ieph = gpsiq.rinex_select(eph, n, WEEK, SEC)
    vis = [sv for sv in range(32) if eph[ieph, sv]["vflg"]]
    trk = np.zeros(len(vis), dtype=TRACK_DTYPE)
    for i, sv in enumerate(vis):
        sbf = gpsiq.nav_subframes(eph[ieph, sv]["nav"], utc)
        st = np.zeros(1, dtype=NAV_STATE_DTYPE)
        gpsiq.nav_message(sbf, WEEK, SEC, True, st)
        trk[i]["prn"] = sv + 1
        trk[i]["g0_week"], trk[i]["g0_sec"] = st[0]["g0_week"], st[0]["g0_sec"]
        trk[i]["dwrd"] = st[0]["dwrd"]
    orbit = np.ascontiguousarray(eph[ieph, vis]["orbit"])
    iono = np.zeros((), dtype=gpsiq.IONO_DTYPE)
    iono["enable"], iono["vflg"], iono["alpha"], iono["beta"] = 1, utc["vflg"], utc["alpha"], utc["beta"]
    xyz = np.repeat(TOKYO[None, :], nblocks + 1, axis=0)
    gpsiq.track_init(orbit, iono, WEEK, SEC, xyz[0], trk)
    carr = trk["carr_phase"].copy()
    desc = gpsiq.refresh_batch(orbit, iono, WEEK, SEC, xyz[1:], trk)
    desc["carr_phase"] = carr[None, :]
    return desc, (orbit, iono, xyz, vis, ieph, utc)


def test_rinex_to_descriptors_matches_reference(ref, tmp_path):
    """BASELINE config 1/2 shape: static lat/lon/h + RINEX v2 nav file -> the per-block channel
    state at gps.c:2766, through the reference's own readRinex2 / eph2sbf / generateNavMsg /
    computeRange / computeCodePhase lines vs the library's."""
    recs = synth_rinex_records(12, TOKYO, WEEK, SEC, seed=31, sets=2)
    path = write_rinex_nav(str(tmp_path / "static.21n"), recs, UTC, 2)
    desc, (orbit, iono, xyz, vis, ieph, utc) = chain_from_rinex(path, 2, nblocks=120)
    reph, rutc, rn = ref.read_rinex(path, 2)
    rtrk = np.zeros(len(vis), dtype=TRACK_DTYPE)
    for i, sv in enumerate(vis):
        st = np.zeros(1, dtype=NAV_STATE_DTYPE)
        ref.nav_message(ref.nav_subframes(reph[ieph, sv]["nav"], rutc), WEEK, SEC, True, st)
        rtrk[i]["prn"], rtrk[i]["g0_week"], rtrk[i]["g0_sec"], rtrk[i]["dwrd"] = sv + 1, st[0]["g0_week"], st[0]["g0_sec"], st[0]["dwrd"]
    want, carr = ref.refresh_blocks(np.ascontiguousarray(reph[ieph, vis]["orbit"]), iono, WEEK, SEC, xyz, rtrk)
    for f in ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "dwrd"):
        assert same(desc[f], want[f]), f
    assert same(desc["carr_phase"][0], carr)
    assert len(vis) == 12 and (desc["iword"] < 60).all() and ieph == 1


@pytest.mark.gpu
def test_rinex_to_samples_on_gpu(oracle, tmp_path):
    """The same chain down to IQ bytes: RINEX v3 (gzip) -> descriptors -> HIP synthesis ==
    oracle, int8 2.6 Msps (BASELINE config 2: 12 visible channels)."""
    recs = synth_rinex_records(12, TOKYO, WEEK, SEC, seed=33, sets=2)
    path = write_rinex_nav(str(tmp_path / "static.rnx.gz"), recs, UTC, 3, gzip_it=True)
    desc, _ = chain_from_rinex(path, 3, nblocks=10)
    fs, ns = 2.6e6, 260000
    ctx = gpsiq.Context(0)
    out = ctx.generate_batch(desc, ns, fs, SC08)
    q = oracle.quantize_blocks(desc, fs, ns)
    for b in (0, 4, 9):
        assert np.array_equal(out[b], oracle.block_fixed(q[b], ns, SC08, seq=True))
    ctx.close()


def test_start_time_overwrite_matches_reference(ref, tmp_path):
    """gpsiq_rinex_overwrite_time == the reference's -T lines (gps.c:2507-2513, 2534-2561): every time of clock, time of
    ephemeris, calendar time and the UTC reference, for start times years later, earlier, across week boundaries and
    with fractional seconds; afterwards the first set serves the new start time."""
    import ctypes as C
    L = ref.lib
    L.ref_time_overwrite.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double]
    recs = synth_rinex_records(11, TOKYO, WEEK, SEC, seed=12, sets=3)
    path = write_rinex_nav(str(tmp_path / "old.21n"), recs, UTC, 2)
    for week, sec in ((2400, 123456.789), (2190, 270000.0), (2191, 5.5), (2189, 604799.999), (1000, 7199.0), (2500, 0.0)):
        eph, utc, n = gpsiq.rinex_read(path, 2)
        want, wutc = eph.copy(), np.asarray(utc).copy().reshape(1)
        L.ref_time_overwrite(want.ctypes.data, n, wutc.ctypes.data, week, sec)
        utc1 = np.asarray(utc).copy().reshape(1)
        gpsiq.rinex_overwrite_time(eph, n, utc1, week, sec)
        assert same(eph, want) and same(utc1, wutc), (week, sec)
        assert utc1["wnt"][0] == week and utc1["tot"][0] == (int(sec) // 7200) * 7200
        assert gpsiq.rinex_select(eph, n, week, float(int(sec) // 7200 * 7200) + 10.0) == 0
        assert not same(eph, gpsiq.rinex_read(path, 2)[0]) or (week, sec) == (2190, 270000.0)
