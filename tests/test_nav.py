"""Navigation-message words (gpsiq_nav_*) against the reference's own
eph2sbf / generateNavMsg / computeChecksum lines and against the GPS parity equations."""
import os

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import NAV_ALM_DTYPE, NAV_EPH_DTYPE, NAV_STATE_DTYPE, NAV_UTC_DTYPE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_eph(rng, week=2190):
    e = np.zeros((), dtype=NAV_EPH_DTYPE)
    e["toe_week"] = week
    e["toe_sec"] = e["toc_sec"] = 7200.0 * rng.integers(0, 84)
    e["iode"] = rng.integers(0, 256)
    e["iodc"] = int(e["iode"]) + 256 * rng.integers(0, 4)
    e["deltan"] = rng.uniform(3e-9, 6e-9)
    for f, a in (("cuc", 5e-6), ("cus", 9e-6), ("cic", 3e-7), ("cis", 3e-7), ("crc", 350.0), ("crs", 120.0)):
        e[f] = rng.uniform(-a, a)
    e["ecc"] = rng.uniform(1e-4, 0.025)
    e["sqrta"] = rng.uniform(5153.0, 5154.5)
    for f in ("m0", "omg0", "aop"):
        e[f] = rng.uniform(-np.pi, np.pi)
    e["inc0"] = rng.uniform(0.93, 0.99)
    e["omgdot"] = rng.uniform(-9e-9, -7e-9)
    e["idot"] = rng.uniform(-6e-10, 6e-10)
    e["af0"], e["af1"], e["af2"] = rng.uniform(-8e-4, 8e-4), rng.uniform(-2e-11, 2e-11), 0.0
    e["tgd"] = rng.uniform(-2e-8, 2e-8)
    return e


def rand_utc(rng, vflg=1):
    u = np.zeros((), dtype=NAV_UTC_DTYPE)
    u["vflg"] = vflg
    u["alpha"] = [rng.uniform(5e-9, 3e-8), rng.uniform(-2e-8, 2e-8), rng.uniform(-1e-7, 1e-7), rng.uniform(-2e-7, 2e-7)]
    u["beta"] = [rng.uniform(8e4, 1.4e5), rng.uniform(-3e5, 3e5), rng.uniform(-2e5, 2e5), rng.uniform(-1e6, 1.2e6)]
    u["A0"], u["A1"] = rng.uniform(-3e-8, 3e-8), rng.uniform(-5e-14, 5e-14)
    u["dtls"], u["tot"], u["wnt"] = 18, 4096 * int(rng.integers(0, 147)), 2190
    return u


def rand_alm(rng):
    a = np.zeros(32, dtype=NAV_ALM_DTYPE)
    for sv in range(32):
        if rng.random() < 0.15:
            continue
        a[sv]["svid"], a[sv]["valid"] = sv + 1, 1
        a[sv]["toa_week"], a[sv]["toa_sec"] = 2190, 4096.0 * rng.integers(0, 147)
        a[sv]["e"], a[sv]["delta_i"] = rng.uniform(1e-3, 0.02), rng.uniform(-0.02, 0.02)
        a[sv]["omegadot"], a[sv]["sqrta"] = rng.uniform(-9e-9, -7e-9), rng.uniform(5153.0, 5154.5)
        for f in ("omega0", "aop", "m0"):
            a[sv][f] = rng.uniform(-np.pi, np.pi)
        a[sv]["af0"], a[sv]["af1"] = rng.uniform(-8e-4, 8e-4), rng.uniform(-3e-11, 3e-11)
    return a


def spec_parity_ok(word, prev):
    """IS-GPS-200 20.3.5.2 written out independently: the six parity equations over the
    source bits d1..d24 (recovered with D30*) and D29*, D30* of the previous word."""
    D29s, D30s = (prev >> 1) & 1, prev & 1
    D = [(word >> (29 - i)) & 1 for i in range(30)]              # D[0] = D1 ... D[29] = D30
    d = [b ^ D30s for b in D[:24]]
    eq = {25: (D29s, [1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23]),
          26: (D30s, [2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24]),
          27: (D29s, [1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22]),
          28: (D30s, [2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23]),
          29: (D30s, [1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24]),
          30: (D29s, [3, 5, 6, 8, 9, 10, 11, 13, 15, 19, 22, 23, 24])}
    for n, (star, idx) in eq.items():
        p = star
        for i in idx:
            p ^= d[i - 1]
        if p != D[n - 1]:
            return False
    return True


def test_parity_matches_reference_and_spec(ref):
    rng = np.random.default_rng(1)
    before = ref.parity_complaints()
    for _ in range(3000):
        src = int(rng.integers(0, 1 << 32))
        nib = bool(rng.integers(0, 2))
        w = gpsiq.nav_parity(src, nib)
        assert w == ref.nav_parity(src, nib)
        assert spec_parity_ok(w & 0x3FFFFFFF, src >> 30)
        if nib:
            assert (w & 3) == 0                                   # words 2 and 10 end in 00
    assert ref.parity_complaints() == before                      # the reference's own checkers agree


@pytest.mark.parametrize("with_alm,vflg", [(False, 1), (True, 1), (True, 0), (False, 0)])
def test_subframes_match_reference(ref, with_alm, vflg):
    rng = np.random.default_rng(7 + 2 * with_alm + vflg)
    for _ in range(25):
        e, u = rand_eph(rng), rand_utc(rng, vflg)
        alm = rand_alm(rng) if with_alm else None
        assert np.array_equal(gpsiq.nav_subframes(e, u, alm), ref.nav_subframes(e, u, alm))


def test_message_roll_matches_reference_over_an_hour(ref):
    """init at allocation, then a 30 s refresh 120 times (all 25 pages, twice and more):
    identical 60-word buffers, page counter and reference time; every word passes the spec
    parity with its predecessor; TOW in each HOW counts 6 s steps."""
    rng = np.random.default_rng(11)
    e, u, alm = rand_eph(rng), rand_utc(rng), rand_alm(rng)
    sbf = gpsiq.nav_subframes(e, u, alm)
    week, sec = 2190, 345612.3
    a = np.zeros(1, dtype=NAV_STATE_DTYPE)
    b = np.zeros(1, dtype=NAV_STATE_DTYPE)
    gpsiq.nav_message(sbf, week, sec, True, a)
    ref.nav_message(sbf, week, sec, True, b)
    assert a.tobytes() == b.tobytes()
    for k in range(120):
        sec += 30.0
        gpsiq.nav_message(sbf, week, sec, False, a)
        ref.nav_message(sbf, week, sec, False, b)
        assert a.tobytes() == b.tobytes(), k
        d = a[0]["dwrd"]
        for i in range(1, 60):
            assert spec_parity_ok(int(d[i]) & 0x3FFFFFFF, int(d[i - 1]) & 3), (k, i)
        tows = [(int(d[10 * s + 1]) >> 13) & 0x1FFFF for s in range(1, 6)]
        if not any(int(d[10 * s]) & 1 for s in range(1, 6)):      # HOW not inverted by D30*
            assert tows == list(range(tows[0], tows[0] + 5))
        assert (int(d[10]) >> 22) & 0xFF == 0x8B or (int(d[10]) >> 22) & 0xFF == 0x74   # preamble (or inverted)
    assert ref.parity_complaints() == 0


def test_golden_nav_capture():
    z = np.load(os.path.join(GOLD, "nav_words.npz"))
    e = np.ascontiguousarray(z["eph"]).view(NAV_EPH_DTYPE).reshape(())
    u = np.ascontiguousarray(z["utc"]).view(NAV_UTC_DTYPE).reshape(())
    alm = np.ascontiguousarray(z["alm"]).view(NAV_ALM_DTYPE).reshape(32)
    sbf = gpsiq.nav_subframes(e, u, alm)
    assert np.array_equal(sbf, z["sbf"])
    st = np.zeros(1, dtype=NAV_STATE_DTYPE)
    gpsiq.nav_message(sbf, int(z["week"]), float(z["sec"]), True, st)
    assert np.array_equal(st[0]["dwrd"], z["dwrd_seq"][0])
    for k in range(1, len(z["dwrd_seq"])):
        gpsiq.nav_message(sbf, int(z["week"]), float(z["sec"]) + 30.0 * k, False, st)
        assert np.array_equal(st[0]["dwrd"], z["dwrd_seq"][k])


def _sem_text(records, announced=None, week=2189, sec=405504, blank=True, svn_blank=()):
    lines = ["%d CURRENT.ALM" % (len(records) if announced is None else announced), " %d %d" % (week, sec)]
    for k, r in enumerate(records):
        if blank:
            lines.append("")
        lines += ["%d" % r["id"], "" if k in svn_blank else "%d" % (40 + r["id"]), "%d" % r.get("ura", 0),
                  " %.14E %.14E %.14E" % (r["e"], r["di"], r["od"]), " %.14E %.14E %.14E" % (r["sq"], r["o0"], r["w"]),
                  " %.14E %.14E %.14E" % (r["m0"], r["af0"], r["af1"]), "%d" % r.get("health", 0), "%d" % r.get("cfg", 11)]
    return "\n".join(lines) + "\n"


def test_sem_almanac_reader_matches_reference(ref, tmp_path):
    """gpsiq_almanac_read_sem == almanac_read_file (almanac.c:73-184), entry for entry: a well-formed file, blank and
    missing optional lines, ids 0 and > 32, more records than announced and fewer (end of file inside a record), a
    damaged number in the middle (everything dropped), an empty file, no file."""
    import ctypes as C
    import os
    from gpsiq.abi import NAV_ALM_DTYPE
    L = ref.lib
    L.ref_almanac_read.argtypes = [C.c_void_p]
    rng = np.random.default_rng(21)

    def rec(i):
        return dict(id=i, e=rng.uniform(0, 0.02), di=rng.uniform(-0.01, 0.01), od=rng.uniform(-3e-9, -2e-9), sq=rng.uniform(5153.0, 5154.0),
                    o0=rng.uniform(-1, 1), w=rng.uniform(-1, 1), m0=rng.uniform(-1, 1), af0=rng.uniform(-1e-4, 1e-4), af1=rng.uniform(-1e-11, 1e-11),
                    ura=int(rng.integers(0, 20)), health=int(rng.integers(0, 70)), cfg=int(rng.integers(0, 20)))
    full = [rec(i) for i in range(1, 32)]
    texts = {
        "full": _sem_text(full),
        "no_blank_lines": _sem_text(full[:9], blank=False),
        "svn_blank": _sem_text(full[:6], svn_blank=(1, 4)),
        "odd_ids": _sem_text([rec(0), rec(33), rec(7), rec(7)]),
        "more_than_announced": _sem_text(full[:12], announced=5),
        "fewer_than_announced": _sem_text(full[:4], announced=31),
        "announced_zero": _sem_text(full[:3], announced=0),
        "cut_inside_a_record": _sem_text(full[:5])[:-60],
        "damaged_number": _sem_text(full[:8]).replace("E-", "X-", 7).replace("X-", "E-", 6),
        "empty": "",
        "header_only": "3 X.ALM\n",
    }
    cwd = os.getcwd()
    try:
        for name, text in texts.items():
            d = tmp_path / name
            d.mkdir()
            (d / "almanac.sem").write_text(text)
            os.chdir(d)                                            # the reference opens "almanac.sem" where it runs
            want = np.zeros(32, dtype=NAV_ALM_DTYPE)
            L.ref_almanac_read(want.ctypes.data)
            got, n = gpsiq.almanac_read_sem(d / "almanac.sem")
            assert got.tobytes() == want.tobytes(), name
            assert n == int((want["valid"] != 0).sum()), name
        assert gpsiq.almanac_read_sem(tmp_path / "full" / "almanac.sem")[1] == 31
        assert gpsiq.almanac_read_sem(tmp_path / "damaged_number" / "almanac.sem")[1] == 0
        assert gpsiq.almanac_read_sem(tmp_path / "cut_inside_a_record" / "almanac.sem")[1] == 4
    finally:
        os.chdir(cwd)
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.almanac_read_sem(tmp_path / "nothing.sem")
