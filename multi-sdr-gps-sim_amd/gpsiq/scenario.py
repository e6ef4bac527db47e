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


# ---- scenarios for the host refresh (gpsiq_refresh_batch) -------------------------------
GM_EARTH = 3.986005e14          # reference gps.h:89-90
OMEGA_EARTH = 7.2921151467e-5
WGS84_A, WGS84_E = 6378137.0, 0.0818191908426


def llh_to_ecef(lat_deg, lon_deg, h):
    lat, lon = np.radians(lat_deg), np.radians(lon_deg)
    n = WGS84_A / np.sqrt(1.0 - (WGS84_E * np.sin(lat)) ** 2)
    return np.array([(n + h) * np.cos(lat) * np.cos(lon), (n + h) * np.cos(lat) * np.sin(lon),
                     ((1.0 - WGS84_E ** 2) * n + h) * np.sin(lat)])


def _elevation_deg(e, sec, xyz):
    """Plain Kepler orbit -> elevation; only used to pick visible satellites."""
    tk = sec - e["toe_sec"]
    mk = e["m0"] + e["n"] * tk
    ek = mk
    for _ in range(12):
        ek = ek + (mk - ek + e["ecc"] * np.sin(ek)) / (1.0 - e["ecc"] * np.cos(ek))
    pk = np.arctan2(e["sq1e2"] * np.sin(ek), np.cos(ek) - e["ecc"]) + e["aop"]
    rk = e["A"] * (1.0 - e["ecc"] * np.cos(ek))
    ok = e["omg0"] + tk * e["omgkdot"] - OMEGA_EARTH * e["toe_sec"]
    xp, yp = rk * np.cos(pk), rk * np.sin(pk)
    pos = np.array([xp * np.cos(ok) - yp * np.cos(e["inc0"]) * np.sin(ok),
                    xp * np.sin(ok) + yp * np.cos(e["inc0"]) * np.cos(ok), yp * np.sin(e["inc0"])])
    los = pos - xyz
    up = xyz / np.linalg.norm(xyz)
    return float(np.degrees(np.arcsin(np.dot(los, up) / np.linalg.norm(los))))


def synth_constellation(nsat, xyz, sec, seed=1, min_elev_deg=8.0, max_elev_deg=None):
    """nsat broadcast-ephemeris sets (EPHEM_DTYPE) of GPS-like orbits that are all above
    min_elev_deg (and, if given, below max_elev_deg: satellites about to rise or set) at
    receiver position xyz and time-of-week sec, with the working variables the reference
    derives when it reads RINEX (A, n, sq1e2, omgkdot: gps.c:1456-1461)."""
    from .abi import EPHEM_DTYPE
    rng = SplitMix64(seed)
    out = np.zeros(nsat, dtype=EPHEM_DTYPE)
    k = 0
    toe = float(int(sec / 7200.0) * 7200)
    while k < nsat:
        e = np.zeros((), dtype=EPHEM_DTYPE)
        e["sqrta"] = 5153.6 + (rng.uniform() - 0.5)
        e["A"] = e["sqrta"] * e["sqrta"]
        e["ecc"] = 0.001 + 0.019 * rng.uniform()
        e["sq1e2"] = np.sqrt(1.0 - e["ecc"] * e["ecc"])
        e["n"] = np.sqrt(GM_EARTH / (e["A"] * e["A"] * e["A"])) + (3.0 + 2.0 * rng.uniform()) * 1e-9
        e["inc0"] = np.radians(55.0 + 2.0 * (rng.uniform() - 0.5))
        e["idot"] = (rng.uniform() - 0.5) * 6e-10
        e["omg0"] = 2.0 * np.pi * rng.uniform() - np.pi
        e["omgkdot"] = -(7.5 + rng.uniform()) * 1e-9 - OMEGA_EARTH
        e["aop"] = 2.0 * np.pi * rng.uniform() - np.pi
        e["m0"] = 2.0 * np.pi * rng.uniform() - np.pi
        for f, a in (("cuc", 3e-6), ("cus", 8e-6), ("cic", 2e-7), ("cis", 2e-7), ("crc", 300.0), ("crs", 80.0)):
            e[f] = (rng.uniform() - 0.5) * 2.0 * a
        e["af0"] = (rng.uniform() - 0.5) * 4e-4
        e["af1"] = (rng.uniform() - 0.5) * 2e-11
        e["af2"] = 0.0
        e["tgd"] = -(0.5 + rng.uniform()) * 1e-8
        e["toe_sec"] = toe
        e["toc_sec"] = toe
        el = _elevation_deg(e, sec, xyz)
        if el > min_elev_deg and (max_elev_deg is None or el < max_elev_deg):
            out[k] = e
            k += 1
    return out


def synth_iono(kind="klobuchar"):
    from .abi import IONO_DTYPE
    io = np.zeros((), dtype=IONO_DTYPE)
    if kind == "off":
        return io
    io["enable"] = 1
    if kind == "klobuchar":
        io["vflg"] = 1
        io["alpha"] = [0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06]
        io["beta"] = [0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07]
    return io


def synth_tracks(nsat, week, sec, seed=1, prn0=1):
    """Per-channel host state before gpsiq_track_init: prn, nav-word buffer (random 30-bit
    words) and its start time g0 (the start of the 30 s frame that contains sec)."""
    from .abi import TRACK_DTYPE
    rng = SplitMix64(seed + 77)
    t = np.zeros(nsat, dtype=TRACK_DTYPE)
    for c in range(nsat):
        t[c]["prn"] = (prn0 - 1 + c) % 32 + 1
        t[c]["g0_week"] = week
        t[c]["g0_sec"] = float(int(sec / 30.0) * 30)
        t[c]["dwrd"] = [rng.next() & 0x3FFFFFFF for _ in range(N_DWRD)]
    return t


def circle_track(centre_xyz, nblocks, radius_m=100.0, period_s=60.0, dt=0.1):
    """Receiver positions for nblocks+1 epochs on a horizontal circle around centre_xyz."""
    up = centre_xyz / np.linalg.norm(centre_xyz)
    east = np.cross([0.0, 0.0, 1.0], up)
    east /= np.linalg.norm(east)
    north = np.cross(up, east)
    t = np.arange(nblocks + 1) * dt
    ang = 2.0 * np.pi * t / period_s
    return centre_xyz[None, :] + radius_m * (np.cos(ang)[:, None] * north[None, :] + np.sin(ang)[:, None] * east[None, :])


# ---- synthetic RINEX navigation files (the reference ships none) ---------------------------
def _d19(x):
    return ("%19.12E" % x).replace("E", "D")


def _d12(x):
    return ("%12.4E" % x).replace("E", "D")


def gps_to_calendar(week, sec):
    import datetime
    t = datetime.datetime(1980, 1, 6) + datetime.timedelta(weeks=int(week), seconds=float(sec))
    return t.year, t.month, t.day, t.hour, t.minute, t.second


def write_rinex_nav(path, records, utc=None, version=2, gzip_it=False, extra_comment=True):
    """records: list of dicts with keys prn, week, toc_sec + the broadcast parameters
    (af0 af1 af2 iode crs deltan m0 cuc ecc cus sqrta toe_sec cic omg0 cis inc0 crc aop omgdot idot
    code toe_week flag sva svh tgd iodc tx fit), written in file order.  utc: dict with alpha[4],
    beta[4], A0, A1, tot, wnt, dtls or None (then the header carries no iono/UTC records).
    Column layout per the RINEX 2.10 / 3.0x navigation message file definition."""
    v3 = version == 3
    L = []

    def hdr(body, label):
        L.append("%-60s%s" % (body, label))

    if v3:
        hdr("     3.02           N: GNSS NAV DATA    G: GPS", "RINEX VERSION / TYPE")
    else:
        hdr("     2.10           N: GPS NAV DATA", "RINEX VERSION / TYPE")
    hdr("gpsiq scenario      synthetic           20250215 000000 UTC", "PGM / RUN BY / DATE")
    if extra_comment:
        hdr("synthetic constellation for parity tests", "COMMENT")
    if utc is not None:
        if v3:
            hdr("GPSA " + "".join(_d12(a) for a in utc["alpha"]), "IONOSPHERIC CORR")
            hdr("GPSB " + "".join(_d12(b) for b in utc["beta"]), "IONOSPHERIC CORR")
            hdr("GPUT " + ("%17.10E" % utc["A0"]).replace("E", "D") + ("%16.9E" % utc["A1"]).replace("E", "D")
                + "%7d%5d" % (utc["tot"], utc["wnt"]), "TIME SYSTEM CORR")
        else:
            hdr("  " + "".join(_d12(a) for a in utc["alpha"]), "ION ALPHA")
            hdr("  " + "".join(_d12(b) for b in utc["beta"]), "ION BETA")
            hdr("   " + _d19(utc["A0"]) + _d19(utc["A1"]) + "%9d%9d" % (utc["tot"], utc["wnt"]), "DELTA-UTC: A0,A1,T,W")
        hdr("%6d" % utc["dtls"], "LEAP SECONDS")
    hdr("", "END OF HEADER")
    pad = "    " if v3 else "   "
    for r in records:
        y, mo, d, hh, mi, ss = gps_to_calendar(r["week"], r["toc_sec"])
        if v3:
            head = "G%02d %04d %02d %02d %02d %02d %02d" % (r["prn"], y, mo, d, hh, mi, ss)
        else:
            head = "%2d %02d %2d %2d %2d %2d%5.1f" % (r["prn"], y % 100, mo, d, hh, mi, float(ss))
        L.append(head + _d19(r["af0"]) + _d19(r["af1"]) + _d19(r["af2"]))
        for keys in (("iode", "crs", "deltan", "m0"), ("cuc", "ecc", "cus", "sqrta"), ("toe_sec", "cic", "omg0", "cis"),
                     ("inc0", "crc", "aop", "omgdot"), ("idot", "code", "toe_week", "flag"), ("sva", "svh", "tgd", "iodc"),
                     ("tx", "fit")):
            L.append(pad + "".join(_d19(float(r[k])) for k in keys))
    data = ("\n".join(L) + "\n").encode("ascii")
    if gzip_it:
        import gzip
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)
    return path


def synth_rinex_records(nsat, xyz, week, sec, seed=1, sets=2, eph=None):
    """Broadcast records for nsat satellites visible at (week, sec), repeated for `sets`
    consecutive two-hour issues (so the reader has to group them into hourly sets).
    eph: optional ready-made orbits (EPHEM_DTYPE[nsat]) instead of synth_constellation's."""
    if eph is None:
        eph = synth_constellation(nsat, xyz, sec, seed=seed)
    rng = SplitMix64(seed + 5)
    recs = []
    for k in range(sets):
        toc = float(int(sec / 7200.0) * 7200 + 7200 * k)
        for c in range(nsat):
            e = eph[c]
            # keep the orbit continuous across issues: advance the mean anomaly to the new toe
            m0 = float(e["m0"] + e["n"] * (toc - e["toe_sec"]))
            r = dict(prn=c + 1, week=week, toc_sec=toc, af0=float(e["af0"]), af1=float(e["af1"]), af2=0.0,
                     iode=(17 + c + k) % 256, crs=float(e["crs"]), deltan=float(e["n"] - np.sqrt(GM_EARTH / e["A"] ** 3)),
                     m0=m0, cuc=float(e["cuc"]), ecc=float(e["ecc"]), cus=float(e["cus"]), sqrta=float(e["sqrta"]),
                     toe_sec=toc, cic=float(e["cic"]), omg0=float(e["omg0"]), cis=float(e["cis"]), inc0=float(e["inc0"]),
                     crc=float(e["crc"]), aop=float(e["aop"]), omgdot=float(e["omgkdot"] + OMEGA_EARTH), idot=float(e["idot"]),
                     code=1, toe_week=week, flag=0, sva=int(rng.below(4)), svh=0 if c != 3 else 5, tgd=float(e["tgd"]),
                     iodc=(17 + c + k) % 256 + 256, tx=toc - 18.0, fit=4.0)
            recs.append(r)
    return recs
