"""numpy views of the C-ABI structs in include/gpsiq.h (layout checked by tests)."""
import numpy as np

MAX_CHAN = 16
N_DWRD = 60
CA_SEQ_LEN = 1023
CARR_FRAC_BITS = 59
CODE_FRAC_BITS = 56
SC08, SC16 = 1, 2
SINK_IQFILE, SINK_HACKRF, SINK_PLUTOSDR = 1, 2, 3
HACKRF_CHUNK = 262144

# gpsiq_chan_t (mirrors the channel_t fields the loop reads, reference gps.h:213-236)
CHAN_DTYPE = np.dtype([
    ("prn", "<i4"), ("iword", "<i4"), ("ibit", "<i4"), ("icode", "<i4"),
    ("f_carr", "<f8"), ("f_code", "<f8"), ("carr_phase", "<f8"), ("code_phase", "<f8"),
    ("gain", "<f8"), ("dwrd", "<u4", (N_DWRD,)),
], align=True)
assert CHAN_DTYPE.itemsize == 296

# gpsiq_qchan_t
QCHAN_DTYPE = np.dtype([
    ("carr_phase", "<u8"), ("carr_step", "<i8"), ("code_frac", "<u8"), ("code_step", "<u8"),
    ("gain", "<f8"), ("nav_bits", "<u4"), ("chip0", "<u2"), ("icode", "u1"), ("prn", "u1"),
], align=True)
assert QCHAN_DTYPE.itemsize == 48

# gpsiq_patch_t (GPSIQ_NCO_REFERENCE)
PATCH_DTYPE = np.dtype([("block", "<u4"), ("sample", "<u4"), ("slot", "u1"), ("neg", "u1"), ("lut", "<u2")], align=True)
assert PATCH_DTYPE.itemsize == 12
NCO_FIXED, NCO_REFERENCE = 0, 1
# gpsiq_chain_in_t (the three fields of gpsiq_chan_t the serial carrier chain of GPSIQ_NCO_REFERENCE reads)
CHAIN_IN_DTYPE = np.dtype([("f_carr", "<f8"), ("carr_phase", "<f8"), ("prn", "<i4"), ("reserved", "<i4")], align=True)
assert CHAIN_IN_DTYPE.itemsize == 24
# gpsiq_chain_est_t / gpsiq_chain_map_t (the time-parallel carrier chain)
CHAIN_EST_DTYPE = np.dtype([("r_hi", "<u8"), ("r_lo", "<u8"), ("drift", "<f8"), ("carr", "<f8"), ("f_carr", "<f8"),
                            ("prn", "<i4"), ("flags", "<i4"), ("first_prn", "<i4"), ("reserved", "<i4")], align=True)
assert CHAIN_EST_DTYPE.itemsize == 56
CHAIN_MAP_DTYPE = np.dtype([("xs", "<f8"), ("e", "<f8"), ("cum", "<i8", (2,)), ("lo", "<i8"), ("hi", "<i8"), ("ok", "<i4"), ("info", "<i4")], align=True)
assert CHAIN_MAP_DTYPE.itemsize == 56
# gpsiq_chain_range_t (csrc/gpsiq_plumbing.h): what a range of blocks does to one slot's accumulator
CHAIN_RANGE_DTYPE = np.dtype([("xs", "<f8"), ("e", "<f8"), ("t", "<i8", (4,)), ("lo", "<i8", (4,)), ("hi", "<i8", (4,)), ("cum_last", "<i8", (2,)),
                              ("abs_end", "<f8"), ("first_prn", "<i4"), ("last_prn", "<i4"), ("ok", "<i4"), ("restart", "<i4"), ("nblocks", "<i4"),
                              ("grid_last", "<i4")], align=True)
assert CHAIN_RANGE_DTYPE.itemsize == 160
CHAIN_EXACT, CHAIN_RESEEDED, CHAIN_EMPTY = 1, 2, 4
# gpsiq_shard_carry_t
SHARD_CARRY_DTYPE = np.dtype([("end_phase", "<u8"), ("advance", "<u8"), ("first_prn", "<i4"), ("last_prn", "<i4"),
                              ("reseeded", "<i4"), ("nblocks", "<i4")], align=True)
assert SHARD_CARRY_DTYPE.itemsize == 32


def elem_dtype(sample_size):
    return np.int8 if sample_size == SC08 else np.int16


# gpsiq_ephem_t / gpsiq_iono_t / gpsiq_track_t (host refresh, include/gpsiq.h)
EPHEM_FIELDS = ["toe_sec", "toc_sec", "m0", "n", "ecc", "sqrta", "sq1e2", "A", "aop", "omg0", "omgkdot", "inc0", "idot",
                "cuc", "cus", "cic", "cis", "crc", "crs", "af0", "af1", "af2", "tgd"]
EPHEM_DTYPE = np.dtype([(f, "<f8") for f in EPHEM_FIELDS], align=True)
assert EPHEM_DTYPE.itemsize == 23 * 8
IONO_DTYPE = np.dtype([("enable", "<i4"), ("vflg", "<i4"), ("alpha", "<f8", (4,)), ("beta", "<f8", (4,))], align=True)
assert IONO_DTYPE.itemsize == 72
TRACK_DTYPE = np.dtype([("prn", "<i4"), ("g0_week", "<i4"), ("g0_sec", "<f8"), ("rho0_week", "<i4"), ("rho0_sec", "<f8"),
                        ("rho0_range", "<f8"), ("carr_phase", "<f8"), ("dwrd", "<u4", (N_DWRD,))], align=True)
assert TRACK_DTYPE.itemsize == 288

# navigation message structs (include/gpsiq.h)
NAV_EPH_DTYPE = np.dtype([("toe_week", "<i4"), ("iode", "<i4"), ("iodc", "<i4"), ("reserved", "<i4"),
                          ("toe_sec", "<f8"), ("toc_sec", "<f8")] +
                         [(f, "<f8") for f in ["deltan", "cuc", "cus", "cic", "cis", "crc", "crs", "ecc", "sqrta", "m0", "omg0",
                                                "inc0", "aop", "omgdot", "idot", "af0", "af1", "af2", "tgd"]], align=True)
assert NAV_EPH_DTYPE.itemsize == 184
NAV_UTC_DTYPE = np.dtype([("vflg", "<i4"), ("dtls", "<i4"), ("tot", "<i4"), ("wnt", "<i4"),
                          ("alpha", "<f8", (4,)), ("beta", "<f8", (4,)), ("A0", "<f8"), ("A1", "<f8")], align=True)
assert NAV_UTC_DTYPE.itemsize == 96
NAV_ALM_DTYPE = np.dtype([("svid", "<u4"), ("valid", "<u4"), ("toa_week", "<i4"), ("reserved", "<i4"), ("toa_sec", "<f8")] +
                         [(f, "<f8") for f in ["e", "delta_i", "omegadot", "sqrta", "omega0", "aop", "m0", "af0", "af1"]], align=True)
assert NAV_ALM_DTYPE.itemsize == 96
NAV_STATE_DTYPE = np.dtype([("dwrd", "<u4", (N_DWRD,)), ("ipage", "<i4"), ("g0_week", "<i4"), ("g0_sec", "<f8")], align=True)
assert NAV_STATE_DTYPE.itemsize == 256

RINEX_EPH_DTYPE = np.dtype([("vflg", "<i4"), ("sva", "<i4"), ("svh", "<i4"), ("code", "<i4"), ("flag", "<i4"),
                            ("t_y", "<i4"), ("t_m", "<i4"), ("t_d", "<i4"), ("t_hh", "<i4"), ("t_mm", "<i4"),
                            ("t_sec", "<f8"), ("fit", "<f8"), ("toc_week", "<i4"), ("reserved", "<i4"),
                            ("orbit", EPHEM_DTYPE), ("nav", NAV_EPH_DTYPE)], align=True)
assert RINEX_EPH_DTYPE.itemsize == 432
EPHEM_SETS, MAX_SAT = 13, 32
