"""The batched host refresh (gpsiq_refresh_batch) against the
reference's own computeRange / computeCodePhase / gain lines (oracle/_ref) and against
committed captures of them.  Bit-exact: same operand order, same libm."""
import os

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import CHAN_DTYPE, EPHEM_DTYPE, IONO_DTYPE, SC16, TRACK_DTYPE
from gpsiq.scenario import (circle_track, llh_to_ecef, synth_constellation, synth_iono, synth_tracks)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOKYO = llh_to_ecef(35.681298, 139.766247, 10.0)        # BASELINE configs 1/2: static lat/lon/h
WEEK, SEC = 2190, 259200.0 + 3.0 * 3600                 # a Wednesday 03:00 GPS time

FIELDS = ["prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "dwrd"]


def product_refresh(eph, iono, xyz, trk0, sec=SEC, gain_x2=False, nthreads=0):
    trk = trk0.copy()
    gpsiq.track_init(eph, iono, WEEK, sec, xyz[0], trk)
    carr = trk["carr_phase"].copy()
    out = gpsiq.refresh_batch(eph, iono, WEEK, sec, xyz[1:], trk, gain_x2=gain_x2, nthreads=nthreads)
    return out, carr, trk


def assert_same(got, want):
    for f in FIELDS:
        a, b = got[f], want[f]
        assert a.tobytes() == b.tobytes(), (f, np.argwhere(a != b)[:5], a[a != b][:3], b[a != b][:3])


@pytest.mark.parametrize("iono_kind", ["klobuchar", "flat", "off"])
def test_range_model_matches_reference(ref, iono_kind):
    iono = synth_iono(iono_kind)
    eph = synth_constellation(16, TOKYO, SEC, seed=3)
    trk = synth_tracks(16, WEEK, SEC)
    for dt in (0.0, 12.3, 1799.9, 7300.0):
        t2 = trk.copy()
        gpsiq.track_init(eph, iono, WEEK, SEC + dt, TOKYO, t2)
        for c in range(16):
            r = ref.compute_range(eph[c], iono, WEEK, SEC + dt, TOKYO)
            assert r["range"] == t2[c]["rho0_range"], (c, dt)
            if dt == 0.0:
                assert r["el"] > 0.0


@pytest.mark.parametrize("case", ["static", "circle", "fast_circle", "pluto", "iono_off"])
def test_refresh_batch_is_bit_exact_vs_reference(ref, case):
    nb, nc = 450, 16                                   # 45 s: crosses the 30 s frame boundary in iword
    iono = synth_iono("off" if case == "iono_off" else "klobuchar")
    eph = synth_constellation(nc, TOKYO, SEC, seed=5)
    trk = synth_tracks(nc, WEEK, SEC)
    if case == "static":
        xyz = np.repeat(TOKYO[None, :], nb + 1, axis=0)
    elif case == "fast_circle":
        xyz = circle_track(TOKYO, nb, radius_m=2000.0, period_s=20.0)
    else:
        xyz = circle_track(TOKYO, nb)
    want, carr_want = ref.refresh_blocks(eph, iono, WEEK, SEC, xyz, trk, sdr_type=3 if case == "pluto" else 1)
    got, carr, trk_end = product_refresh(eph, iono, xyz, trk, gain_x2=(case == "pluto"))
    assert_same(got, want)
    assert carr.tobytes() == carr_want.tobytes()
    assert (np.abs(got["f_carr"]) < 12000.0).all() and (got["gain"] > 0.05).all()
    # continuing a second batch from the carried track state == one long batch
    more = circle_track(TOKYO, nb + 50)[nb:] if case != "static" else np.repeat(TOKYO[None, :], 51, axis=0)
    if case in ("static", "circle"):
        full_xyz = np.concatenate([xyz, more[1:]])
        want2, _ = ref.refresh_blocks(eph, iono, WEEK, SEC, full_xyz, trk)
        w, s = WEEK, SEC
        for _ in range(nb):
            w, s = ref.inc_gps_time(w, s, 0.1)
        got2 = gpsiq.refresh_batch(eph, iono, w, s, more[1:], trk_end)
        assert_same(got2, want2[nb:])


def test_reference_circle_csv_when_present(ref):
    """BASELINE config 4's motion file (reference circle.csv, 3000 rows of t,x,y,z ECEF)."""
    path = "/root/reference/circle.csv"
    if not os.path.exists(path):
        pytest.skip("reference motion file not on this box")
    xyz = np.loadtxt(path, delimiter=",")[:601, 1:4]
    eph = synth_constellation(12, xyz[0], SEC, seed=9)
    trk = synth_tracks(12, WEEK, SEC)
    iono = synth_iono()
    want, carr_want = ref.refresh_blocks(eph, iono, WEEK, SEC, xyz, trk)
    got, carr, _ = product_refresh(eph, iono, xyz, trk)
    assert_same(got, want)
    assert carr.tobytes() == carr_want.tobytes()


def test_threads_do_not_change_results():
    eph = synth_constellation(9, TOKYO, SEC, seed=6)
    trk = synth_tracks(9, WEEK, SEC)
    xyz = circle_track(TOKYO, 1000)
    a, _, _ = product_refresh(eph, synth_iono(), xyz, trk, nthreads=1)
    b, _, _ = product_refresh(eph, synth_iono(), xyz, trk, nthreads=7)
    c, _, _ = product_refresh(eph, synth_iono(), xyz, trk, nthreads=0)
    assert a.tobytes() == b.tobytes() == c.tobytes()


def test_week_rollover_and_unused_slots(ref):
    sec = 604800.0 - 1.25                              # the batch crosses the end of the GPS week
    eph = synth_constellation(5, TOKYO, sec, seed=8)
    trk = synth_tracks(5, WEEK, sec)
    trk["prn"][2] = 0
    xyz = np.repeat(TOKYO[None, :], 41, axis=0)
    want, _ = ref.refresh_blocks(eph, synth_iono(), WEEK, sec, xyz, trk)
    got, _, trk_end = product_refresh(eph, synth_iono(), xyz, trk, sec=sec)
    assert_same(got, want)
    assert trk_end["rho0_week"][0] == WEEK + 1 and got["prn"][:, 2].max() == 0


def test_golden_refresh_capture():
    """Committed capture of the reference's refresh (runs without /root/reference)."""
    z = np.load(os.path.join(GOLD, "refresh_circle.npz"))
    eph = np.ascontiguousarray(z["eph"]).view(EPHEM_DTYPE).reshape(-1)
    iono = np.ascontiguousarray(z["iono"]).view(IONO_DTYPE).reshape(())
    trk = np.ascontiguousarray(z["trk"]).view(TRACK_DTYPE).reshape(-1)
    want = np.ascontiguousarray(z["desc"]).view(CHAN_DTYPE).reshape(z["desc"].shape[0], z["desc"].shape[1])
    got, carr, _ = product_refresh(eph, iono, z["xyz"], trk, sec=float(z["sec"]))
    assert_same(got, want)
    assert carr.tobytes() == z["carr_init"].tobytes()


@pytest.mark.gpu
def test_scenario_to_samples_end_to_end(oracle):
    """ephemeris -> refresh (host C) -> quantise -> HIP synthesis, dynamic receiver, int16 2.6 Msps
    (BASELINE config 4 shape): equals the oracle on the same descriptors, and the Doppler the
    refresh produced is what the circle implies."""
    fs, ns, nb, nc = 2.6e6, 260000, 20, 16
    eph = synth_constellation(nc, TOKYO, SEC, seed=12)
    trk = synth_tracks(nc, WEEK, SEC)
    xyz = circle_track(TOKYO, nb, radius_m=500.0, period_s=30.0)
    desc, carr, _ = product_refresh(eph, synth_iono(), xyz, trk)
    desc["carr_phase"] = carr[None, :]
    ctx = gpsiq.Context(0)
    out = ctx.generate_batch(desc, ns, fs, SC16)
    q = oracle.quantize_blocks(desc, fs, ns)
    for b in (0, 1, nb // 2, nb - 1):
        assert np.array_equal(out[b], oracle.block_fixed(q[b], ns, SC16, seq=True))
    assert np.ptp(desc["f_carr"], axis=0).max() > 50.0      # 105 m/s circle: hundreds of Hz of Doppler swing
    ctx.close()


def test_refresh_epochs_equals_one_call_per_epoch():
    """gpsiq_refresh_epochs (several navigation-message epochs in one threaded pass) == gpsiq_refresh_batch called once
    per epoch with the word buffer of that epoch, for random epoch cuts; and its argument checks."""
    import gpsiq
    from gpsiq.abi import CHAN_DTYPE, TRACK_DTYPE
    from gpsiq.scenario import circle_track, llh_to_ecef, synth_constellation, synth_iono, synth_tracks
    pos = llh_to_ecef(35.681298, 139.766247, 10.0)
    week, sec, nb, nc = 2190, 270000.0, 700, 9
    eph = synth_constellation(nc, pos, sec, seed=31)
    iono = synth_iono()
    xyz = circle_track(pos, nb, radius_m=120.0, period_s=40.0)
    rng = np.random.default_rng(4)
    for trial in range(4):
        cuts = [0] + sorted(set(int(c) for c in rng.integers(1, nb, size=int(rng.integers(0, 5)))))
        trk0 = synth_tracks(nc, week, sec, seed=trial)
        gpsiq.track_init(eph, iono, week, sec, xyz[0], trk0)
        trk_ep = np.stack([trk0.copy() for _ in cuts])
        for e in range(1, len(cuts)):                               # another word buffer and g0 per epoch
            trk_ep[e]["dwrd"] = rng.integers(0, 1 << 30, size=(nc, 60), dtype=np.uint32)
            trk_ep[e]["g0_sec"] = trk0["g0_sec"] + 30.0 * e
        got = gpsiq.refresh_epochs(eph, iono, week, sec, xyz[1:], np.ascontiguousarray(trk_ep), cuts)
        trk = trk0.copy()
        want = np.zeros((nb, nc), dtype=CHAN_DTYPE)
        for e, b0 in enumerate(cuts):
            b1 = cuts[e + 1] if e + 1 < len(cuts) else nb
            trk["dwrd"], trk["g0_week"], trk["g0_sec"] = trk_ep[e]["dwrd"], trk_ep[e]["g0_week"], trk_ep[e]["g0_sec"]
            t = round(round(sec * 1000.0) + 100.0 * b0) / 1000.0
            want[b0:b1] = gpsiq.refresh_batch(eph, iono, week, t, xyz[1 + b0:1 + b1], trk)
        assert got.tobytes() == want.tobytes(), cuts
        for f in ("rho0_week", "rho0_sec", "rho0_range"):
            assert np.array_equal(trk_ep[0][f], trk[f])
    bad = np.ascontiguousarray(np.stack([trk0, trk0]))
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.refresh_epochs(eph, iono, week, sec, xyz[1:], bad, [5, 10])          # must start at block 0
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.refresh_epochs(eph, iono, week, sec, xyz[1:], bad, [0, nb + 1])      # past the end
    bad[1]["prn"][0] = 31
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.refresh_epochs(eph, iono, week, sec, xyz[1:], bad, [0, 100])         # another satellite in an epoch


def test_receiver_position_inputs_match_reference(ref, tmp_path):
    """gpsiq_llh_to_ecef / gpsiq_ecef_to_llh == llh2xyz / xyz2llh (gps.c:361-447) bit for bit, and
    gpsiq_motion_read_csv == readUserMotion (gps.c:2253-2277) on well-formed and damaged files."""
    import ctypes as C
    import gpsiq
    L = ref.lib
    L.ref_llh2xyz.argtypes = [C.c_void_p, C.c_void_p]
    L.ref_xyz2llh.argtypes = [C.c_void_p, C.c_void_p]
    L.ref_read_user_motion.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(8)
    for _ in range(300):
        llh = np.array([rng.uniform(-np.pi / 2, np.pi / 2), rng.uniform(-np.pi, np.pi), rng.uniform(-500.0, 20000.0)])
        want = np.zeros(3)
        L.ref_llh2xyz(llh.ctypes.data, want.ctypes.data)
        got = gpsiq.llh_to_ecef(*llh)
        assert got.tobytes() == want.tobytes()
        back, back_ref = gpsiq.ecef_to_llh(got), np.zeros(3)
        L.ref_xyz2llh(got.ctypes.data, back_ref.ctypes.data)
        assert back.tobytes() == back_ref.tobytes()
    assert gpsiq.ecef_to_llh([0.0, 0.0, 0.0]).tolist() == [0.0, 0.0, -6378137.0]            # the reference's "invalid vector" answer
    from gpsiq.scenario import circle_track, llh_to_ecef
    xyz = circle_track(llh_to_ecef(35.681298, 139.766247, 10.0), 50)
    good = tmp_path / "good.csv"
    good.write_text("".join("%5.1f,%.3f,%.3f,%.3f\n" % (0.1 * k, *p) for k, p in enumerate(xyz)))
    damaged = tmp_path / "damaged.csv"
    damaged.write_text("0.0,1.5,2.5,3.5\n0.1,4.5\n\nnot a number\n0.4,7.0,8.0,9.0\n0.5, 10 , 11 ,12\n")
    for path in (good, damaged):
        want = np.zeros((3000, 3))
        n_ref = L.ref_read_user_motion(str(path).encode(), want.ctypes.data, 3000)
        got = gpsiq.motion_read_csv(str(path), 3000)
        assert len(got) == n_ref and got.tobytes() == want[:n_ref].tobytes(), path
    assert len(gpsiq.motion_read_csv(str(good), 7)) == 7
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.motion_read_csv(str(tmp_path / "missing.csv"))


def test_moving_in_the_local_frame_matches_reference(ref):
    """gpsiq_ecef_add_neu == ltcmat + the three lines of the reference's target offset / interactive step
    (gps.c:449-468, 2354-2356), bit for bit, incl. a chain of 0.1 s steps as the interactive mode makes them."""
    import ctypes as C
    import gpsiq
    L = ref.lib
    L.ref_add_neu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(31)
    for _ in range(200):
        llh = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-3.1, 3.1), rng.uniform(-100.0, 9000.0)])
        xyz = gpsiq.llh_to_ecef(*llh)
        want = xyz.copy()
        for _ in range(5):                                    # five steps in the frame of the start location
            neu = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-5, 5)])
            L.ref_add_neu(llh.ctypes.data, neu.ctypes.data, want.ctypes.data)
            xyz = gpsiq.ecef_add_neu(llh, neu, xyz)
            assert xyz.tobytes() == want.tobytes()
    # the -T form: distance and bearing (milli-degrees in the reference's struct) from the location
    llh = np.array([35.681298 / 57.2957795131, 139.766247 / 57.2957795131, 10.0])
    d, bearing_mdeg, h = 1500.0, 45000.0, 20.0
    neu = np.array([d * np.cos((bearing_mdeg / 1000) / 57.2957795131), d * np.sin((bearing_mdeg / 1000) / 57.2957795131), h])
    moved = gpsiq.ecef_add_neu(llh, neu, gpsiq.llh_to_ecef(*llh))
    back = gpsiq.ecef_to_llh(moved)
    assert abs(np.linalg.norm(moved - gpsiq.llh_to_ecef(*llh)) - np.hypot(d, h)) < 1e-6 and back[0] > llh[0] and back[1] > llh[1]


def test_time_conversions_match_reference(ref):
    """gpsiq_date_to_gps / gpsiq_gps_to_date == date2gps / gps2date (gps.c:315-355): random dates 1981-2080 (and the epoch itself) incl. leap
    days and week boundaries, exact doubles, and the round trip."""
    import ctypes as C
    import gpsiq
    L = ref.lib
    L.ref_date2gps.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    L.ref_gps2date.argtypes = [C.c_int, C.c_double] + [C.c_void_p] * 6
    rng = np.random.default_rng(15)
    cases = [(1980, 1, 6, 0, 0, 0.0), (2000, 2, 29, 23, 59, 59.999), (2019, 4, 7, 0, 0, 0.0), (2021, 6, 20, 0, 0, 0.0), (2024, 12, 31, 12, 30, 30.5)]
    for _ in range(2000):
        y, m = int(rng.integers(1981, 2081)), int(rng.integers(1, 13))
        d = int(rng.integers(1, 29 if m == 2 else 31))
        cases.append((y, m, d, int(rng.integers(0, 24)), int(rng.integers(0, 60)), float(rng.choice([0.0, 0.1, 29.9, 59.0]) + rng.integers(0, 2) * rng.random())))
    for y, m, d, hh, mm, sec in cases:
        w, s = C.c_int(0), C.c_double(0.0)
        L.ref_date2gps(y, m, d, hh, mm, sec, C.byref(w), C.byref(s))
        got = gpsiq.date_to_gps(y, m, d, hh, mm, sec)
        assert got == (w.value, s.value), (y, m, d, hh, mm, sec)
        v = [C.c_int(0) for _ in range(5)]
        fs = C.c_double(0.0)
        L.ref_gps2date(w.value, s.value, *[C.byref(x) for x in v], C.byref(fs))
        back = gpsiq.gps_to_date(*got)
        assert back == tuple(x.value for x in v) + (fs.value,), (y, m, d, hh, mm, sec)
        assert back[:5] == (y, m, d, hh, mm) and abs(back[5] - sec) < 1e-6
    assert gpsiq.date_to_gps(2021, 6, 20) == (2163, 0.0)          # a Sunday 00:00: the start of GPS week 2163
