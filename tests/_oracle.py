"""ctypes bindings of the CPU checker (oracle/liboracle.so, oracle/_ref/libgpsref.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

import sys
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
from gpsiq.abi import (CHAN_DTYPE, QCHAN_DTYPE, SC08, SC16, elem_dtype,  # noqa: E402
                       EPHEM_DTYPE, IONO_DTYPE, TRACK_DTYPE, NAV_EPH_DTYPE, NAV_UTC_DTYPE, NAV_ALM_DTYPE,
                       NAV_STATE_DTYPE, RINEX_EPH_DTYPE)


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path):
        L = self.lib = C.CDLL(path)
        L.oracle_sin512.restype = C.c_int
        L.oracle_cos512.restype = C.c_int
        L.oracle_codegen.argtypes = [C.c_int, C.c_void_p]
        L.oracle_block_float.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_block_float_closed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_quantize.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_block_fixed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_block_fixed_seq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_block_fixed_range.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_int, C.c_void_p]
        L.oracle_carrier_chain.argtypes = [C.c_double, C.c_double, C.c_long]
        L.oracle_carrier_chain.restype = C.c_double
        L.oracle_chunk_plan.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_int]

    def tables(self):
        s = np.array([self.lib.oracle_sin512(k) for k in range(512)], dtype=np.int32)
        c = np.array([self.lib.oracle_cos512(k) for k in range(512)], dtype=np.int32)
        return s, c

    def codegen(self, prn):
        ca = np.zeros(1023, dtype=np.uint8)
        rc = self.lib.oracle_codegen(prn, _ptr(ca))
        if rc:
            raise ValueError(rc)
        return ca

    def block_float(self, ch, nsamp, fs, sample_size):
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        out = np.zeros(2 * nsamp, dtype=elem_dtype(sample_size))
        carr = np.zeros(len(ch), dtype=np.float64)
        rc = self.lib.oracle_block_float(_ptr(ch), len(ch), nsamp, fs, sample_size, _ptr(out), _ptr(carr))
        if rc:
            raise ValueError(rc)
        return out, carr

    def block_float_closed(self, ch, nsamp, fs, sample_size):
        """The float loop evaluated piecewise in closed form (no per-sample recurrence)."""
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        out = np.zeros(2 * nsamp, dtype=elem_dtype(sample_size))
        carr = np.zeros(len(ch), dtype=np.float64)
        rc = self.lib.oracle_block_float_closed(_ptr(ch), len(ch), nsamp, fs, sample_size, _ptr(out), _ptr(carr))
        if rc:
            raise ValueError(rc)
        return out, carr

    def carrier_chain(self, x0, inc, nsamp):
        """carr_phase after nsamp passes of gps.c:2821-2826."""
        return self.lib.oracle_carrier_chain(float(x0), float(inc), int(nsamp))

    def quantize(self, ch, fs, nsamp, carry_in=None):
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        q = np.zeros(len(ch), dtype=QCHAN_DTYPE)
        cout = np.zeros(len(ch), dtype=np.uint64)
        cin = None if carry_in is None else np.ascontiguousarray(carry_in, dtype=np.uint64)
        rc = self.lib.oracle_quantize(_ptr(ch), len(ch), fs, nsamp, _ptr(q),
                                      None if cin is None else _ptr(cin), _ptr(cout))
        if rc:
            raise ValueError(rc)
        return q, cout

    def quantize_blocks(self, desc, fs, nsamp):
        """[nblocks][nchan] descriptors -> quantised, carrier carried exactly between
        blocks (re-seeded when a slot's prn changes), as gpsiq_generate_batch does."""
        nb, nc = desc.shape
        q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
        carry = None
        for b in range(nb):
            cin = None
            if b > 0:
                cin = carry.copy()
                for c in range(nc):
                    if desc[b, c]["prn"] != desc[b - 1, c]["prn"]:
                        cin[c] = np.uint64(int(np.floor(np.ldexp(desc[b, c]["carr_phase"], 59))))
            q[b], carry = self.quantize(desc[b], fs, nsamp, cin)
        return q

    def block_fixed(self, q, nsamp, sample_size, seq=False):
        q = np.ascontiguousarray(q, dtype=QCHAN_DTYPE)
        out = np.zeros(2 * nsamp, dtype=elem_dtype(sample_size))
        fn = self.lib.oracle_block_fixed_seq if seq else self.lib.oracle_block_fixed
        rc = fn(_ptr(q), len(q), nsamp, sample_size, _ptr(out))
        if rc:
            raise ValueError(rc)
        return out

    def block_fixed_range(self, q, n0, cnt, sample_size):
        q = np.ascontiguousarray(q, dtype=QCHAN_DTYPE)
        out = np.zeros(2 * cnt, dtype=elem_dtype(sample_size))
        rc = self.lib.oracle_block_fixed_range(_ptr(q), len(q), n0, cnt, sample_size, _ptr(out))
        if rc:
            raise ValueError(rc)
        return out

    def chunk_plan(self, sink_kind, nelem, nblocks):
        buf = np.zeros(4 * (nblocks + 1) + (nelem * nblocks) // 262144 + 8, dtype=np.uint64)
        n = self.lib.oracle_chunk_plan(sink_kind, nelem, nblocks, _ptr(buf), len(buf))
        if n < 0:
            raise ValueError(n)
        return buf[:n].astype(np.int64)


class Ref:
    """The reference's own lines (oracle/ref_slice.c)."""

    def __init__(self, path):
        L = self.lib = C.CDLL(path)
        L.ref_run_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p]
        L.ref_compute_code_phase.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, C.c_double,
                                             C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p]

    def tables(self):
        s = np.zeros(512, dtype=np.int32)
        c = np.zeros(512, dtype=np.int32)
        assert self.lib.ref_tables(_ptr(s), _ptr(c)) == 0
        return s, c

    def codegen(self, prn):
        ca = np.zeros(1023, dtype=np.int32)
        self.lib.ref_codegen(prn, _ptr(ca))
        return ca.astype(np.uint8)

    def run_blocks(self, desc, fs, sample_size, sdr_type=1):
        """desc [nblocks][nchan] -> (elements in enqueue order, chunk lengths, carr_out)"""
        desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
        nb, nc = desc.shape
        nelem = 2 * (fs // 10) * nb
        out = np.zeros(nelem, dtype=elem_dtype(sample_size))
        chunks = np.zeros(nelem // 262144 + nb + 8, dtype=np.uint64)
        carr = np.zeros((nb, nc), dtype=np.float64)
        n_out = C.c_size_t(0)
        n_chunks = C.c_int(0)
        rc = self.lib.ref_run_blocks(_ptr(desc), nb, nc, int(fs), sample_size, sdr_type,
                                     _ptr(out), nelem, C.byref(n_out), _ptr(chunks), len(chunks),
                                     C.byref(n_chunks), _ptr(carr))
        if rc:
            raise RuntimeError(rc)
        return out[: n_out.value], chunks[: n_chunks.value].astype(np.int64), carr

    def compute_range(self, eph, iono, week, sec, xyz):
        """satpos + computeRange: dict(pos, vel, clk, range, rate, d, az, el, iono)"""
        eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
        iono = np.ascontiguousarray(iono, dtype=IONO_DTYPE)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        o = np.zeros(14)
        self.lib.ref_compute_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        assert self.lib.ref_compute_range(_ptr(eph), _ptr(iono), int(week), float(sec), _ptr(xyz), _ptr(o)) == 0
        return dict(pos=o[0:3], vel=o[3:6], clk=o[6:8], range=o[8], rate=o[9], d=o[10], az=o[11], el=o[12], iono=o[13])

    def inc_gps_time(self, week, sec, dt):
        w, s = C.c_int(week), C.c_double(sec)
        self.lib.ref_inc_gps_time.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        self.lib.ref_inc_gps_time(C.byref(w), C.byref(s), dt)
        return w.value, s.value

    def refresh_blocks(self, eph, iono, week, sec, xyz, trk, sdr_type=1):
        """xyz [nblocks+1][3] (xyz[0] = allocation position) -> (descriptors [nblocks][nchan], carr_init[nchan])"""
        eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
        iono = np.ascontiguousarray(iono, dtype=IONO_DTYPE)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        trk = np.ascontiguousarray(trk, dtype=TRACK_DTYPE)
        nb, nc = len(xyz) - 1, len(trk)
        out = np.zeros((nb, nc), dtype=CHAN_DTYPE)
        carr = np.zeros(nc)
        self.lib.ref_refresh_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = self.lib.ref_refresh_blocks(_ptr(eph), _ptr(iono), int(week), float(sec), _ptr(xyz), nb, nc, sdr_type,
                                         _ptr(trk), _ptr(out), _ptr(carr))
        if rc:
            raise RuntimeError(rc)
        return out, carr

    def refresh_epochs(self, eph, iono, week, sec, xyz, trk, sbf, ipage, sdr_type=1):
        """refresh_blocks with the reference's 30 s navigation-message refresh in the loop
        (gps.c:2870, 2878-2885).  sbf [nchan][53][10], ipage [nchan] as left by generateNavMsg(init)."""
        eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
        iono = np.ascontiguousarray(iono, dtype=IONO_DTYPE)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        trk = np.ascontiguousarray(trk, dtype=TRACK_DTYPE)
        sbf = np.ascontiguousarray(sbf, dtype=np.uint32)
        ipage = np.ascontiguousarray(ipage, dtype=np.int32)
        nb, nc = len(xyz) - 1, len(trk)
        assert sbf.shape == (nc, 53, 10) and ipage.shape == (nc,)
        out = np.zeros((nb, nc), dtype=CHAN_DTYPE)
        carr = np.zeros(nc)
        self.lib.ref_refresh_epochs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = self.lib.ref_refresh_epochs(_ptr(eph), _ptr(iono), int(week), float(sec), _ptr(xyz), nb, nc, sdr_type,
                                         _ptr(trk), _ptr(sbf), _ptr(ipage), _ptr(out), _ptr(carr))
        if rc:
            raise RuntimeError(rc)
        return out, carr

    def run_host(self, eph_sets, ieph, utc, week, sec, xyz, nchan, sdr_type=1):
        """The reference's host side of the block loop with channel allocation in it
        (allocateChannel at the start and at every 30 s refresh, the nav-message refresh and the
        switch to the next ephemeris set: gps.c:2663-2675, 2731-2765, 2870, 2878-2909).
        eph_sets: gpsiq_rinex_eph_t[nsets][32], ieph the set to start with; xyz [nblocks+1][3].
        Returns (descriptors [nblocks][nchan], nsat of every allocateChannel call, final ieph)."""
        eph_sets = np.ascontiguousarray(eph_sets, dtype=RINEX_EPH_DTYPE)
        utc = np.ascontiguousarray(utc, dtype=NAV_UTC_DTYPE)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        assert eph_sets.ndim == 2 and eph_sets.shape[1] == 32
        nb = len(xyz) - 1
        out = np.zeros((nb, nchan), dtype=CHAN_DTYPE)
        nsat = np.zeros(nb // 300 + 3, dtype=np.int32)
        ieph_end = C.c_int(-1)
        self.lib.ref_run_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        rc = self.lib.ref_run_host(_ptr(eph_sets), len(eph_sets), int(ieph), _ptr(utc), int(week), float(sec), _ptr(xyz), nb,
                                   int(nchan), sdr_type, _ptr(out), _ptr(nsat), len(nsat), C.byref(ieph_end))
        if rc < 0:
            raise RuntimeError(rc)
        return out, nsat[:rc], ieph_end.value

    def nav_parity(self, source, nib=False):
        self.lib.ref_nav_parity.restype = C.c_uint
        return int(self.lib.ref_nav_parity(C.c_uint(int(source) & 0xFFFFFFFF), int(bool(nib))))

    def parity_complaints(self):
        return int(self.lib.ref_parity_complaints())

    def nav_subframes(self, eph, utc, alm=None):
        eph = np.ascontiguousarray(eph, dtype=NAV_EPH_DTYPE)
        utc = np.ascontiguousarray(utc, dtype=NAV_UTC_DTYPE)
        a = None if alm is None else np.ascontiguousarray(alm, dtype=NAV_ALM_DTYPE)
        sbf = np.zeros((53, 10), dtype=np.uint32)
        self.lib.ref_nav_subframes.argtypes = [C.c_void_p] * 4
        rc = self.lib.ref_nav_subframes(_ptr(eph), _ptr(utc), None if a is None else _ptr(a), _ptr(sbf))
        if rc:
            raise RuntimeError(rc)
        return sbf

    def nav_message(self, sbf, week, sec, init, state):
        sbf = np.ascontiguousarray(sbf, dtype=np.uint32)
        self.lib.ref_nav_message.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
        rc = self.lib.ref_nav_message(_ptr(sbf), int(week), float(sec), int(bool(init)), _ptr(state))
        if rc:
            raise RuntimeError(rc)
        return state

    def read_rinex(self, path, version=2):
        eph = np.zeros((13, 32), dtype=RINEX_EPH_DTYPE)
        utc = np.zeros(1, dtype=NAV_UTC_DTYPE)
        date = C.create_string_buffer(21)
        self.lib.ref_read_rinex.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_char_p]
        n = self.lib.ref_read_rinex(int(version), os.fsencode(path), _ptr(eph), _ptr(utc), date)
        return eph, utc[0], n

    def compute_code_phase(self, rho0_range, rho0_g, g0, rho1_range, dt, dwrd, prn):
        out = np.zeros(1, dtype=CHAN_DTYPE)
        dwrd = np.ascontiguousarray(dwrd, dtype=np.uint32)
        rc = self.lib.ref_compute_code_phase(rho0_range, rho0_g[0], rho0_g[1], g0[0], g0[1],
                                             rho1_range, dt, _ptr(dwrd), prn, _ptr(out))
        if rc:
            raise RuntimeError(rc)
        return out[0]


def apply_patches(orc, q, out, patches, sample_size):
    """CPU stand-in for the device fix-up of GPSIQ_NCO_REFERENCE (apply_patches in gpsiq_kernels.hip):
    q = one block's quantised descriptors, out = its 2*nsamp fixed-point elements (modified in place),
    patches = that block's gpsiq_patch_t entries.  Every patched sample is recomputed whole from the
    closed form of include/gpsiq.h with the patched channels' (lut, neg) substituted."""
    sin512, cos512 = orc.tables()
    active = [c for c in range(len(q)) if q[c]["prn"] > 0]           # device order: active channels first
    codes = {c: orc.codegen(int(q[c]["prn"])) for c in active}
    by_sample = {}
    for p in patches:
        by_sample.setdefault(int(p["sample"]), {})[int(p["slot"])] = (int(p["lut"]), int(p["neg"]))

    def s16(v):
        v &= 0xFFFF
        return v - 65536 if v >= 32768 else v

    for n, over in by_sample.items():
        acc_i = acc_q = 0
        for slot, c in enumerate(active):
            d = q[c]
            if slot in over:
                idx, neg = over[slot]
            else:
                idx = ((int(d["carr_phase"]) + int(d["carr_step"]) * n) % (1 << 59)) >> 50
                a = int(d["chip0"]) + ((int(d["code_frac"]) + int(d["code_step"]) * n) >> 56)
                neg = int(codes[c][a % 1023]) ^ ((int(d["nav_bits"]) >> ((int(d["icode"]) + a // 1023) // 20)) & 1)
            tc, ts = int(int(cos512[idx]) * float(d["gain"])), int(int(sin512[idx]) * float(d["gain"]))
            acc_i += -tc if neg else tc
            acc_q += -ts if neg else ts
        for k, v in ((2 * n, acc_i), (2 * n + 1, acc_q)):
            v = s16(v)
            if sample_size == SC08:
                v = (v >> 4) & 0xFF
                v = v - 256 if v >= 128 else v
            out[k] = v
    return out


def load_oracle():
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    return Oracle(path)


def load_ref():
    path = os.path.join(ORACLE_DIR, "_ref", "libgpsref.so")
    if not os.path.exists(path) and os.path.exists("/root/reference/gps.c"):
        build()
    return Ref(path) if os.path.exists(path) else None
