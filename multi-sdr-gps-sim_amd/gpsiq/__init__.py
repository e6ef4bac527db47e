"""gpsiq — Python host binding of libgpsiq (the C-ABI in include/gpsiq.h).

The shared library holds the hand-written gfx950 kernels; this module only marshals
numpy descriptors and raw device pointers (e.g. ``torch.Tensor.data_ptr()``) into it.
There is no fallback: if libgpsiq.so is missing the import fails, and without a GPU
``Context()`` raises.
"""
import ctypes as C
import os

import numpy as np

from .abi import (CHAN_DTYPE, QCHAN_DTYPE, SC08, SC16, SINK_HACKRF, SINK_IQFILE,  # noqa: F401
                  SINK_PLUTOSDR, HACKRF_CHUNK, MAX_CHAN, elem_dtype, EPHEM_DTYPE, IONO_DTYPE, TRACK_DTYPE,
                  NAV_EPH_DTYPE, NAV_UTC_DTYPE, NAV_ALM_DTYPE, NAV_STATE_DTYPE, RINEX_EPH_DTYPE, PATCH_DTYPE,
                  NCO_FIXED, NCO_REFERENCE, SHARD_CARRY_DTYPE)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPSIQ_LIB: another build of the library (tests/test_gpu_build.py loads the one it has just compiled on the GPU box)
LIB_PATH = os.environ.get("GPSIQ_LIB") or os.path.join(_HERE, "libgpsiq.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or make -C multi-sdr-gps-sim_amd/csrc); gpsiq has no non-HIP path")



def _preload_shared_hip_runtime():
    """PyTorch wheels carry their own libamdhip64.so (SONAME libamdhip64.so.7) and load it by
    file name, so a process that loaded /opt/rocm's copy first ends up with two HIP/HSA
    runtimes and the second one to initialise sees no device.  When torch is installed,
    load ITS runtime first (without importing torch) so libgpsiq.so binds to the same one."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


_preload_shared_hip_runtime()
_lib = C.CDLL(LIB_PATH)
# the rows either side of the path (include/gpsiq_rows.h, gpsiq_extras.h): a host-only library of its own that runs on the first
# one's worker pool and error text (its NEEDED entry "libgpsiq.so" is satisfied by the library just loaded, whatever its path)
ROWS_PATH = os.path.join(os.path.dirname(LIB_PATH), "libgpsiq_rows.so")
if not os.path.exists(ROWS_PATH):
    ROWS_PATH = os.path.join(_HERE, "libgpsiq_rows.so")
if not os.path.exists(ROWS_PATH):
    raise ImportError(f"{ROWS_PATH} not built: make -C multi-sdr-gps-sim_amd/csrc builds it next to libgpsiq.so")
_rows = C.CDLL(ROWS_PATH)


class GpsiqError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"gpsiq error {code}: {text}")
        self.code = code


def _sig(name, restype, *argtypes):
    """A function of the C-ABI (an exported symbol: include/gpsiq.h in libgpsiq.so, include/gpsiq_rows.h and gpsiq_extras.h in
    libgpsiq_rows.so), or of the library's plumbing (csrc/gpsiq_plumbing.h: hidden symbols, resolved through the one exported
    entry gpsiq_plumbing(name))."""
    f = getattr(_lib, name, None) or getattr(_rows, name, None)
    if f is None:
        _lib.gpsiq_plumbing.restype = C.c_void_p
        _lib.gpsiq_plumbing.argtypes = [C.c_char_p]
        addr = _lib.gpsiq_plumbing(name.encode())
        if not addr:
            raise AttributeError(f"{name}: exported by neither libgpsiq.so nor libgpsiq_rows.so, and no plumbing entry of that name")
        return C.CFUNCTYPE(restype, *argtypes)(addr)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_version = _sig("gpsiq_version", C.c_char_p)
_last_error = _sig("gpsiq_last_error", C.c_char_p)
_kernels_id = _sig("gpsiq_kernels_id", C.c_char_p)
_prn_code = _sig("gpsiq_prn_code", _i, _i, _vp)
_carrier_table = _sig("gpsiq_carrier_table", None, _vp, _vp)
_quantize = _sig("gpsiq_quantize", _i, _vp, _i, _d, _i, _vp, _vp, _vp)
_create = _sig("gpsiq_create", _i, C.POINTER(_vp), _i)
_destroy = _sig("gpsiq_destroy", None, _vp)
_generate_block = _sig("gpsiq_generate_block", _i, _vp, _vp, _i, _i, _d, _i, _vp, _vp)
_generate_block_async = _sig("gpsiq_generate_block_async", _i, _vp, _vp, _i, _i, _d, _i, _vp, _vp)
_wait = _sig("gpsiq_wait", _i, _vp)
_generate_batch = _sig("gpsiq_generate_batch", _i, _vp, _vp, _i, _i, _i, _d, _i, _vp, _i, _vp)
_generate_quantized = _sig("gpsiq_generate_quantized", _i, _vp, _vp, _i, _i, _i, _i, _vp, _i)
_generate_seeded = _sig("gpsiq_generate_seeded", _i, _vp, _vp, _i, _i, _i, _d, _i, _vp, _vp, _i)
_quantize_batch = _sig("gpsiq_quantize_batch", _i, _vp, _i, _i, _d, _i, _vp, _vp, _vp)
_reference_batch = _sig("gpsiq_reference_batch", _i, _vp, _i, _i, _d, _i, _vp, _vp, _i, C.POINTER(C.c_int), _vp)
_reference_chain = _sig("gpsiq_reference_chain", _i, _vp, _i, _i, _d, _i, _vp, _vp, _vp, _vp, _vp)
_reference_seeded = _sig("gpsiq_reference_seeded", _i, _vp, _i, _i, _d, _i, _vp, _vp, _vp, _i, C.POINTER(C.c_int))
_reference_stats = _sig("gpsiq_reference_stats", None, _vp)
_chain_inputs = _sig("gpsiq_chain_inputs", None, _vp, _i, _vp)
_chain_maps = _sig("gpsiq_chain_maps", _i, _vp, _i, _i, _d, _i, _vp, _i, _vp, _vp)
_chain_maps_device = _sig("gpsiq_chain_maps_device", _i, _vp, _vp, _i, _i, _d, _i, _vp, _i, _vp, _vp, C.POINTER(C.c_float))
_chain_link = _sig("gpsiq_chain_link", _i, _vp, _vp, _i, _i, _d, _i, _vp, _vp, _vp, _vp, _vp)
_chain_summary = _sig("gpsiq_chain_summary", _i, _vp, _i, _i, _d, _i, _vp, _vp)
_chain_fold = _sig("gpsiq_chain_fold", _i, _vp, _i, _i, _vp)
_chain_range = _sig("gpsiq_chain_range", _i, _vp, _vp, _i, _i, _d, _i, _vp)
_chain_range_fold = _sig("gpsiq_chain_range_fold", _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp)
_chain_stats = _sig("gpsiq_chain_stats", None, _vp)
_set_patches = _sig("gpsiq_set_patches", _i, _vp, _vp, _i)
_set_nco_mode = _sig("gpsiq_set_nco_mode", _i, _vp, _i)
_generate_batch_multi = _sig("gpsiq_generate_batch_multi", _i, _vp, _i, _vp, _i, _i, _i, _d, _i, _vp, _vp, _vp)
_shard_carry = _sig("gpsiq_shard_carry", _i, _vp, _i, _i, _i, _vp)
_shard_seed = _sig("gpsiq_shard_seed", _i, _vp, _i, _i, _i, _vp, _i)
_shard_range = _sig("gpsiq_shard_range", _i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int))
_set_descriptors = _sig("gpsiq_set_descriptors", _i, _vp, _vp, _i, _i)
_launch = _sig("gpsiq_launch", _i, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _i)
_synchronize = _sig("gpsiq_synchronize", _i, _vp, _vp)
_time_launches = _sig("gpsiq_time_launches", _i, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _i, _i, C.POINTER(C.c_float))
_host_alloc = _sig("gpsiq_host_alloc", _vp, _sz)
_host_free = _sig("gpsiq_host_free", None, _vp)
_track_init = _sig("gpsiq_track_init", _i, _vp, _vp, _i, _d, _vp, _vp, _i)
_sat_visibility = _sig("gpsiq_sat_visibility", _i, _vp, _i, _d, _vp, _d, _vp)
_refresh_batch = _sig("gpsiq_refresh_batch", _i, _vp, _vp, _i, _d, _vp, _i, _i, _i, _vp, _vp, _i)
_refresh_epochs = _sig("gpsiq_refresh_epochs", _i, _vp, _vp, _i, _d, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i)
_almanac_read_sem = _sig("gpsiq_almanac_read_sem", _i, C.c_char_p, _vp)
_rinex_overwrite_time = _sig("gpsiq_rinex_overwrite_time", _i, _vp, _i, _vp, _i, _d)
_ecef_add_neu = _sig("gpsiq_ecef_add_neu", None, _vp, _vp, _vp)
_date_to_gps = _sig("gpsiq_date_to_gps", None, _i, _i, _i, _i, _i, _d, _vp, _vp)
_gps_to_date = _sig("gpsiq_gps_to_date", None, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp)
_refresh_epochs_q = _sig("gpsiq_refresh_epochs_quantized", _i, _vp, _vp, _i, _d, _vp, _i, _i, _i, _vp, _vp, _i, _d, _i, _vp, _i)
_llh_to_ecef = _sig("gpsiq_llh_to_ecef", None, _vp, _vp)
_ecef_to_llh = _sig("gpsiq_ecef_to_llh", None, _vp, _vp)
_motion_read_csv = _sig("gpsiq_motion_read_csv", _i, C.c_char_p, _vp, _i)
_nav_parity = _sig("gpsiq_nav_parity", C.c_uint32, C.c_uint32, _i)
_nav_subframes = _sig("gpsiq_nav_subframes", _i, _vp, _vp, _vp, _vp)
_nav_message = _sig("gpsiq_nav_message", _i, _vp, _i, _d, _i, _vp)
_nav_roll = _sig("gpsiq_nav_roll", _i, _vp, _i, _i, _d, _vp)
_rinex_read = _sig("gpsiq_rinex_read", _i, C.c_char_p, _i, _vp, _vp)
_rinex_select = _sig("gpsiq_rinex_select", _i, _vp, _i, _i, _d)
_device_eval_stats = _sig("gpsiq_device_eval_stats", None, _vp)
_device_eval_host_ms = _sig("gpsiq_device_eval_host_ms", _d, _vp)
_num_variants = _sig("gpsiq_num_variants", _i)
_variant_name = _sig("gpsiq_variant_name", C.c_char_p, _i)


def _check(rc):
    if rc < 0:
        raise GpsiqError(rc, _last_error().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(_vp)


def version():
    return _version().decode()


def kernels_id():
    """Identity of the device code in the loaded library (see gpsiq_kernels_id in include/gpsiq.h)."""
    return _kernels_id().decode()


def variants():
    return {_variant_name(v).decode(): v for v in range(_num_variants())}


def prn_code(prn):
    ca = np.zeros(1023, dtype=np.uint8)
    _check(_prn_code(int(prn), _p(ca)))
    return ca


def carrier_table():
    cos = np.zeros(512, dtype=np.int16)
    sin = np.zeros(512, dtype=np.int16)
    _carrier_table(_p(cos), _p(sin))
    return cos, sin


def quantize(ch, fs, nsamp, carry_in=None):
    """One block: gpsiq_chan_t[nchan] -> (gpsiq_qchan_t[nchan], carry_out[nchan])."""
    ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
    q = np.zeros(len(ch), dtype=QCHAN_DTYPE)
    cout = np.zeros(len(ch), dtype=np.uint64)
    cin = None if carry_in is None else np.ascontiguousarray(carry_in, dtype=np.uint64)
    _check(_quantize(_p(ch), len(ch), float(fs), int(nsamp), _p(q), None if cin is None else _p(cin), _p(cout)))
    return q, cout


def quantize_blocks(desc, fs, nsamp, carry0=None):
    """[nblocks][nchan] descriptors -> quantised descriptors with the carrier carried
    exactly from block to block (the rule of gpsiq_generate_batch).  Returns (q, carry_end)."""
    desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
    nb, nc = desc.shape
    q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
    cin = None if carry0 is None else np.ascontiguousarray(carry0, dtype=np.uint64)
    cout = np.zeros(nc, dtype=np.uint64)
    _check(_quantize_batch(_p(desc), nb, nc, float(fs), int(nsamp), _p(q), None if cin is None else _p(cin), _p(cout)))
    return q, cout


def reference_blocks(desc, fs, nsamp):
    """GPSIQ_NCO_REFERENCE form of quantize_blocks: (q, patches, carr_phase_end).  Every block is seeded from
    the carrier phase the reference's double accumulator holds at its start; patches lists the samples where
    the double path differs from the closed form."""
    desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
    nb, nc = desc.shape
    q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
    carr = np.zeros(nc, dtype=np.float64)
    n = C.c_int(0)
    cap = 64 + 8 * nb
    while True:
        pt = np.zeros(cap, dtype=PATCH_DTYPE)
        rc = _reference_batch(_p(desc), nb, nc, float(fs), int(nsamp), _p(q), _p(pt), cap, C.byref(n), _p(carr))
        if rc == -2 and n.value > cap:           # GPSIQ_E_RANGE: not enough room
            cap = n.value
            continue
        _check(rc)
        return q, pt[: n.value].copy(), carr


def chain_inputs(desc):
    """gpsiq_chain_inputs: the three fields per channel and block the serial carrier chain reads, [nblocks][nchan] x 24 bytes."""
    from .abi import CHAIN_IN_DTYPE
    desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
    out = np.zeros(desc.shape, dtype=CHAIN_IN_DTYPE)
    _chain_inputs(_p(desc), desc.size, _p(out))
    return out


def reference_chain(cin, fs, nsamp, carr_in=None, prn_in=None):
    """gpsiq_reference_chain: the serial half of GPSIQ_NCO_REFERENCE alone.  cin [nblocks][nchan] (chain_inputs, or any
    subset of its columns: the channels are independent) -> (carr_start[nblocks][nchan], carr_end[nchan], last_prn[nchan])."""
    from .abi import CHAIN_IN_DTYPE
    cin = np.ascontiguousarray(cin, dtype=CHAIN_IN_DTYPE)
    nb, nc = cin.shape
    start = np.zeros((nb, nc), dtype=np.float64)
    end = np.zeros(nc, dtype=np.float64)
    last = np.zeros(nc, dtype=np.int32)
    ci = None if carr_in is None else np.ascontiguousarray(carr_in, dtype=np.float64)
    pi = None if prn_in is None else np.ascontiguousarray(prn_in, dtype=np.int32)
    _check(_reference_chain(_p(cin), nb, nc, float(fs), int(nsamp), None if ci is None else _p(ci), None if pi is None else _p(pi),
                            _p(start), _p(end), _p(last)))
    return start, end, last


def chain_maps(cin, fs, nsamp, start=None, max_stretches=0, ctx=None):
    """gpsiq_chain_maps (ctx given: gpsiq_chain_maps_device): level 1 of the time-parallel carrier chain, the certified map of
    every block -> (maps[nblocks][nchan], end[nchan]) (+ the kernels' milliseconds on a device)."""
    from .abi import CHAIN_IN_DTYPE, CHAIN_EST_DTYPE, CHAIN_MAP_DTYPE
    cin = np.ascontiguousarray(cin, dtype=CHAIN_IN_DTYPE)
    nb, nc = cin.shape
    maps = np.zeros((nb, nc), dtype=CHAIN_MAP_DTYPE)
    end = np.zeros(nc, dtype=CHAIN_EST_DTYPE)
    st = None if start is None else np.ascontiguousarray(start, dtype=CHAIN_EST_DTYPE)
    if ctx is None:
        _check(_chain_maps(_p(cin), nb, nc, float(fs), int(nsamp), None if st is None else _p(st), int(max_stretches), _p(maps), _p(end)))
        return maps, end
    ms = C.c_float(0.0)
    _check(_chain_maps_device(ctx._h, _p(cin), nb, nc, float(fs), int(nsamp), None if st is None else _p(st), int(max_stretches),
                              _p(maps), _p(end), C.byref(ms)))
    return maps, end, float(ms.value)


def chain_link(cin, maps, fs, nsamp, carr_in=None, prn_in=None):
    """gpsiq_chain_link: level 2, the chain itself -> (carr_start[nblocks][nchan], carr_end[nchan], last_prn[nchan]), equal to
    reference_chain's."""
    from .abi import CHAIN_IN_DTYPE, CHAIN_MAP_DTYPE
    cin = np.ascontiguousarray(cin, dtype=CHAIN_IN_DTYPE)
    maps = np.ascontiguousarray(maps, dtype=CHAIN_MAP_DTYPE)
    nb, nc = cin.shape
    assert maps.shape == (nb, nc)
    start = np.zeros((nb, nc), dtype=np.float64)
    end = np.zeros(nc, dtype=np.float64)
    last = np.zeros(nc, dtype=np.int32)
    ci = None if carr_in is None else np.ascontiguousarray(carr_in, dtype=np.float64)
    pi = None if prn_in is None else np.ascontiguousarray(prn_in, dtype=np.int32)
    _check(_chain_link(_p(cin), _p(maps), nb, nc, float(fs), int(nsamp), None if ci is None else _p(ci), None if pi is None else _p(pi),
                       _p(start), _p(end), _p(last)))
    return start, end, last


def chain_summary(cin, fs, nsamp, start=None):
    """gpsiq_chain_summary: what a range of blocks does to every slot's estimator (start None: the phase pass)."""
    from .abi import CHAIN_IN_DTYPE, CHAIN_EST_DTYPE
    cin = np.ascontiguousarray(cin, dtype=CHAIN_IN_DTYPE)
    nb, nc = cin.shape
    out = np.zeros(nc, dtype=CHAIN_EST_DTYPE)
    st = None if start is None else np.ascontiguousarray(start, dtype=CHAIN_EST_DTYPE)
    _check(_chain_summary(_p(cin), nb, nc, float(fs), int(nsamp), None if st is None else _p(st), _p(out)))
    return out


def chain_fold(sums):
    """gpsiq_chain_fold: the estimator state after the ranges sums[0 .. n) ([n][nchan]), i.e. where range n starts."""
    from .abi import CHAIN_EST_DTYPE
    sums = np.ascontiguousarray(sums, dtype=CHAIN_EST_DTYPE)
    n, nc = sums.shape
    out = np.zeros(nc, dtype=CHAIN_EST_DTYPE)
    _check(_chain_fold(_p(sums) if n else None, n, nc, _p(out)))
    return out


def chain_range(cin, maps, fs, nsamp):
    """gpsiq_chain_range: what this range of blocks does to every slot's accumulator as a function of the state it is entered
    with (its maps composed) -> CHAIN_RANGE_DTYPE[nchan]."""
    from .abi import CHAIN_IN_DTYPE, CHAIN_MAP_DTYPE, CHAIN_RANGE_DTYPE
    cin = np.ascontiguousarray(cin, dtype=CHAIN_IN_DTYPE)
    maps = np.ascontiguousarray(maps, dtype=CHAIN_MAP_DTYPE)
    nb, nc = cin.shape
    out = np.zeros(nc, dtype=CHAIN_RANGE_DTYPE)
    _check(_chain_range(_p(cin) if nb else None, _p(maps) if nb else None, nb, nc, float(fs), int(nsamp), _p(out)))
    return out


def chain_range_fold(ranges, true_end=None, true_prn=None, true_known=None):
    """gpsiq_chain_range_fold over ranges[nranges][nchan]: (every state known, carr, prn, known -- each [nranges + 1][nchan]) --
    the accumulator and satellite every range is entered with, the last row: after the whole timeline.  true_end / true_prn
    [nranges][nchan] with true_known[nranges]: the states ranks that have linked their range ended on."""
    from .abi import CHAIN_RANGE_DTYPE
    ranges = np.ascontiguousarray(ranges, dtype=CHAIN_RANGE_DTYPE)
    n, nc = ranges.shape
    carr = np.zeros((n + 1, nc), dtype=np.float64)
    prn = np.zeros((n + 1, nc), dtype=np.int32)
    known = np.zeros((n + 1, nc), dtype=np.uint8)
    te = tp = tk = None
    if true_known is not None:
        te = np.ascontiguousarray(true_end, dtype=np.float64)
        tp = np.ascontiguousarray(true_prn, dtype=np.int32)
        tk = np.ascontiguousarray(true_known, dtype=np.uint8)
        assert te.shape == (n, nc) and tp.shape == (n, nc) and tk.shape == (n,)
    every = _check(_chain_range_fold(_p(ranges) if n else None, n, nc, None if tk is None else _p(te), None if tk is None else _p(tp),
                                     None if tk is None else _p(tk), _p(carr), _p(prn), _p(known)))
    return bool(every), carr, prn, known.astype(bool)


def chain_stats():
    """gpsiq_chain_stats: (blocks linked through their map, blocks walked from their true start) since the process started."""
    out = np.zeros(2, dtype=np.uint64)
    _chain_stats(_p(out))
    return int(out[0]), int(out[1])


def reference_seeded(desc, fs, nsamp, carr_start):
    """gpsiq_reference_seeded: the parallel half -- (q, patches) of blocks whose start states are known."""
    desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
    nb, nc = desc.shape
    st = np.ascontiguousarray(carr_start, dtype=np.float64)
    assert st.shape == (nb, nc)
    q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
    n = C.c_int(0)
    cap = 64 + 8 * nb
    while True:
        pt = np.zeros(cap, dtype=PATCH_DTYPE)
        rc = _reference_seeded(_p(desc), nb, nc, float(fs), int(nsamp), _p(st), _p(q), _p(pt), cap, C.byref(n))
        if rc == -2 and n.value > cap:
            cap = n.value
            continue
        _check(rc)
        return q, pt[: n.value].copy()


def reference_stats():
    """gpsiq_reference_stats: (states needed, decided from the start state, carrier walks, code walks) since the process started."""
    out = np.zeros(4, dtype=np.uint64)
    _reference_stats(_p(out))
    return tuple(int(v) for v in out)


def device_eval_stats():
    """Since the process started: batch calls taken by the device evaluation, (block, channel) pairs it evaluated, pairs handed
    to the host walker, slots whose chain the host repaired, patches, calls that fell back to the host path."""
    out = np.zeros(6, dtype=np.uint64)
    _device_eval_stats(_p(out))
    return tuple(int(v) for v in out)


def shard_carry(q, nsamp):
    """gpsiq_shard_carry: what this rank's (self-seeded) block range does to each slot's carrier."""
    q = np.ascontiguousarray(q, dtype=QCHAN_DTYPE)
    nb, nc = q.shape
    out = np.zeros(nc, dtype=SHARD_CARRY_DTYPE)
    _check(_shard_carry(_p(q), nb, nc, int(nsamp), _p(out)))
    return out


def shard_seed(q, nsamp, all_carry, rank):
    """gpsiq_shard_seed: add the exact carrier prefix of the ranks before `rank` to q (in place)."""
    assert q.dtype == QCHAN_DTYPE and q.flags.c_contiguous
    nb, nc = q.shape
    allc = np.ascontiguousarray(all_carry, dtype=SHARD_CARRY_DTYPE).reshape(-1, nc)
    _check(_shard_seed(_p(q), nb, nc, int(nsamp), _p(allc), int(rank)))
    return q


def shard_range(nblocks, rank, world):
    """gpsiq_shard_range: the contiguous block range [begin, end) of `rank`."""
    b, e = C.c_int(0), C.c_int(0)
    _check(_shard_range(int(nblocks), int(rank), int(world), C.byref(b), C.byref(e)))
    return b.value, e.value


def track_init(eph, iono, week, sec, xyz, trk):
    """allocateChannel()'s range / carrier-phase initialisation (reference gps.c:2199-2214); trk is updated in place."""
    eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
    iono = np.ascontiguousarray(iono, dtype=IONO_DTYPE)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    assert trk.dtype == TRACK_DTYPE and trk.flags.c_contiguous
    _check(_track_init(_p(eph), _p(iono), int(week), float(sec), _p(xyz), _p(trk), len(trk)))
    return trk


def sat_visibility(eph, week, sec, xyz, elv_mask_deg=0.0):
    """checkSatVisibility() (reference gps.c:2142-2162) for one ephemeris: (visible, azel[2] in radians)."""
    eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    azel = np.zeros(2)
    return bool(_check(_sat_visibility(_p(eph), int(week), float(sec), _p(xyz), float(elv_mask_deg), _p(azel)))), azel


def refresh_batch(eph, iono, week, sec, xyz, trk, gain_x2=False, nthreads=0, out=None):
    """The per-block host refresh (reference gps.c:2731-2765) for len(xyz) blocks -> gpsiq_chan_t[nblocks][nchan].
    out: optional preallocated C-contiguous [len(xyz)][len(trk)] array (every field is written)."""
    eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
    iono = np.ascontiguousarray(iono, dtype=IONO_DTYPE)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    assert trk.dtype == TRACK_DTYPE and trk.flags.c_contiguous
    if out is None:
        out = np.zeros((len(xyz), len(trk)), dtype=CHAN_DTYPE)
    assert out.dtype == CHAN_DTYPE and out.shape == (len(xyz), len(trk)) and out.flags.c_contiguous
    _check(_refresh_batch(_p(eph), _p(iono), int(week), float(sec), _p(xyz), len(xyz), len(trk), int(bool(gain_x2)),
                          _p(trk), _p(out), int(nthreads)))
    return out


def refresh_epochs(eph, iono, week, sec, xyz, trk_epochs, first_block, gain_x2=False, nthreads=0, out=None):
    """gpsiq_refresh_epochs: refresh_batch over several navigation-message epochs in one threaded pass.
    trk_epochs [nepochs][nchan] (TRACK_DTYPE): epoch e's word buffer and g0; first_block [nepochs]."""
    eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
    iono = np.ascontiguousarray(iono, dtype=IONO_DTYPE)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    assert trk_epochs.dtype == TRACK_DTYPE and trk_epochs.flags.c_contiguous and trk_epochs.ndim == 2
    first = np.ascontiguousarray(first_block, dtype=np.int32)
    ne, nc = trk_epochs.shape
    assert len(first) == ne
    if out is None:
        out = np.empty((len(xyz), nc), dtype=CHAN_DTYPE)
    assert out.dtype == CHAN_DTYPE and out.shape == (len(xyz), nc) and out.flags.c_contiguous
    _check(_refresh_epochs(_p(eph), _p(iono), int(week), float(sec), _p(xyz), len(xyz), nc, int(bool(gain_x2)),
                           _p(trk_epochs), _p(first), ne, _p(out), int(nthreads)))
    return out


def refresh_epochs_quantized(eph, iono, week, sec, xyz, trk_epochs, first_block, fs, nsamp, gain_x2=False, nthreads=0, out=None):
    """gpsiq_refresh_epochs_quantized: refresh_epochs + quantize_blocks(carry0=None) in one pass -> QCHAN_DTYPE[nblocks][nchan]."""
    eph = np.ascontiguousarray(eph, dtype=EPHEM_DTYPE)
    iono = np.ascontiguousarray(iono, dtype=IONO_DTYPE)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    assert trk_epochs.dtype == TRACK_DTYPE and trk_epochs.flags.c_contiguous and trk_epochs.ndim == 2
    first = np.ascontiguousarray(first_block, dtype=np.int32)
    ne, nc = trk_epochs.shape
    assert len(first) == ne
    if out is None:
        out = np.empty((len(xyz), nc), dtype=QCHAN_DTYPE)
    assert out.dtype == QCHAN_DTYPE and out.shape == (len(xyz), nc) and out.flags.c_contiguous
    _check(_refresh_epochs_q(_p(eph), _p(iono), int(week), float(sec), _p(xyz), len(xyz), nc, int(bool(gain_x2)),
                             _p(trk_epochs), _p(first), ne, float(fs), int(nsamp), _p(out), int(nthreads)))
    return out


def llh_to_ecef(lat_rad, lon_rad, h):
    """llh2xyz() (reference gps.c:412-447): radians, metres -> ECEF metres."""
    llh = np.array([lat_rad, lon_rad, h], dtype=np.float64)
    xyz = np.zeros(3)
    _llh_to_ecef(_p(llh), _p(xyz))
    return xyz


def ecef_add_neu(llh_ref, neu, xyz):
    """xyz + ltcmat(llh_ref)^T * neu (reference gps.c:2354-2356, 2726-2728): a new array."""
    llh = np.ascontiguousarray(llh_ref, dtype=np.float64)
    d = np.ascontiguousarray(neu, dtype=np.float64)
    out = np.array(xyz, dtype=np.float64).copy()
    _ecef_add_neu(_p(llh), _p(d), _p(out))
    return out


def ecef_to_llh(xyz):
    """xyz2llh() (reference gps.c:361-410)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    llh = np.zeros(3)
    _ecef_to_llh(_p(xyz), _p(llh))
    return llh


def motion_read_csv(path, max_points=3000):
    """readUserMotion() (reference gps.c:2253-2277): -> xyz[n][3]; raises if the file cannot be opened."""
    xyz = np.zeros((max_points, 3), dtype=np.float64)
    n = _motion_read_csv(os.fsencode(path), _p(xyz), int(max_points))
    if n < 0:
        raise GpsiqError(n, _last_error().decode())
    return xyz[:n].copy()


def date_to_gps(year, month, day, hour=0, minute=0, second=0.0):
    """date2gps() (reference gps.c:315-337) -> (week, seconds of the week)."""
    w, sec = C.c_int(0), C.c_double(0.0)
    _date_to_gps(int(year), int(month), int(day), int(hour), int(minute), float(second), C.byref(w), C.byref(sec))
    return w.value, sec.value


def gps_to_date(week, sec):
    """gps2date() (reference gps.c:339-355) -> (year, month, day, hour, minute, second)."""
    v = [C.c_int(0) for _ in range(5)]
    s = C.c_double(0.0)
    _gps_to_date(int(week), float(sec), *[C.byref(x) for x in v], C.byref(s))
    return tuple(x.value for x in v) + (s.value,)


def nav_parity(source, nib=False):
    return int(_nav_parity(int(source) & 0xFFFFFFFF, int(bool(nib))))


def almanac_read_sem(path):
    """almanac_read_file() (reference almanac.c:73-184) -> (NAV_ALM_DTYPE[32], number of valid entries)."""
    alm = np.zeros(32, dtype=NAV_ALM_DTYPE)
    n = _almanac_read_sem(str(path).encode(), _p(alm))
    if n < 0:
        _check(n)
    return alm, n


def nav_subframes(eph, utc, alm=None):
    """eph2sbf(): -> uint32[53][10]"""
    eph = np.ascontiguousarray(eph, dtype=NAV_EPH_DTYPE)
    utc = np.ascontiguousarray(utc, dtype=NAV_UTC_DTYPE)
    sbf = np.zeros((53, 10), dtype=np.uint32)
    a = None if alm is None else np.ascontiguousarray(alm, dtype=NAV_ALM_DTYPE)
    assert a is None or len(a) == 32
    _check(_nav_subframes(_p(eph), _p(utc), None if a is None else _p(a), _p(sbf)))
    return sbf


def nav_message(sbf, week, sec, init, state):
    """generateNavMsg(): state (NAV_STATE_DTYPE scalar array of shape (1,)) is updated in place."""
    sbf = np.ascontiguousarray(sbf, dtype=np.uint32)
    assert sbf.shape == (53, 10) and state.dtype == NAV_STATE_DTYPE and state.size == 1
    _check(_nav_message(_p(sbf), int(week), float(sec), int(bool(init)), _p(state)))
    return state


def nav_roll(sbf, week, sec, states):
    """generateNavMsg(.., 0) for every channel in one call: sbf [nchan][53][10], states NAV_STATE_DTYPE[nchan] (in place)."""
    sbf = np.ascontiguousarray(sbf, dtype=np.uint32)
    assert sbf.ndim == 3 and sbf.shape[1:] == (53, 10) and states.dtype == NAV_STATE_DTYPE and len(states) == len(sbf) and states.flags.c_contiguous
    _check(_nav_roll(_p(sbf), len(sbf), int(week), float(sec), _p(states)))
    return states


def rinex_read(path, version=2):
    """readRinex2/readRinex3 -> (eph[13][32] RINEX_EPH_DTYPE, utc NAV_UTC_DTYPE, nsets).  nsets < 0:
    the reference's error codes (-1 open, -2 version, -3 file type)."""
    eph = np.zeros((13, 32), dtype=RINEX_EPH_DTYPE)
    utc = np.zeros(1, dtype=NAV_UTC_DTYPE)
    n = _rinex_read(os.fsencode(path), int(version), _p(eph), _p(utc))
    return eph, utc[0], n


def rinex_overwrite_time(eph, nsets, utc, week, sec):
    """The reference's -T option (gps.c:2534-2561): shift toc / toe / calendar time of every valid record so that the
    file serves the start time (week, sec); in place on eph[:nsets] and utc."""
    assert eph.dtype == RINEX_EPH_DTYPE and eph.flags.c_contiguous and utc.dtype == NAV_UTC_DTYPE
    _check(_rinex_overwrite_time(_p(eph), int(nsets), _p(utc), int(week), float(sec)))
    return eph, utc


def rinex_select(eph, nsets, week, sec):
    eph = np.ascontiguousarray(eph, dtype=RINEX_EPH_DTYPE)
    return int(_rinex_select(_p(eph), int(nsets), int(week), float(sec)))


def generate_batch_multi(contexts, desc, nsamp, fs, sample_size, host_ptr=None, device_ptrs=None, carr_out=None):
    """gpsiq_generate_batch_multi: one call, one contiguous block range per context.  host_ptr: one host buffer
    for the whole timeline; device_ptrs: one device pointer per context (its own range); neither: a numpy array."""
    desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
    nb, nc = desc.shape
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    co = None if carr_out is None else _p(carr_out)
    out = None
    if device_ptrs is not None:
        dp = (C.c_void_p * len(contexts))(*[int(p) for p in device_ptrs])
        _check(_generate_batch_multi(handles, len(contexts), _p(desc), nb, nc, int(nsamp), float(fs), int(sample_size), None, dp, co))
        return None
    if host_ptr is None:
        out = np.zeros((nb, 2 * nsamp), dtype=elem_dtype(sample_size))
        host_ptr = out.ctypes.data
    _check(_generate_batch_multi(handles, len(contexts), _p(desc), nb, nc, int(nsamp), float(fs), int(sample_size), _vp(host_ptr), None, co))
    return out


class Context:
    """One libgpsiq context = one GPU."""

    def __init__(self, device=0):
        h = _vp()
        _check(_create(C.byref(h), int(device)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            _destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_nco_mode(self, mode):
        """NCO_FIXED (default) or NCO_REFERENCE for generate_block / generate_batch."""
        _check(_set_nco_mode(self._h, int(mode)))

    def set_patches(self, patches):
        patches = np.ascontiguousarray(patches, dtype=PATCH_DTYPE)
        _check(_set_patches(self._h, _p(patches) if len(patches) else None, len(patches)))

    # -- drop-in entry points (host buffers) --
    def generate_block(self, ch, nsamp, fs, sample_size, host_ptr=None):
        """One 0.1 s block, the call that replaces gps.c:2767-2846.  host_ptr: optional
        caller-owned host buffer of 2*nsamp elements (e.g. a page-locked fifo buffer)."""
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        carr = np.zeros(len(ch), dtype=np.float64)
        if host_ptr is not None:
            _check(_generate_block(self._h, _p(ch), len(ch), int(nsamp), float(fs), int(sample_size), _vp(host_ptr), _p(carr)))
            return None, carr
        out = np.zeros(2 * nsamp, dtype=elem_dtype(sample_size))
        _check(_generate_block(self._h, _p(ch), len(ch), int(nsamp), float(fs), int(sample_size), _p(out), _p(carr)))
        return out, carr

    def generate_block_async(self, ch, nsamp, fs, sample_size, host_ptr):
        """gpsiq_generate_block_async: queue one block into the page-locked buffer at host_ptr; returns the carrier
        phases to hand back in (final at once).  The samples are there after wait()."""
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        carr = np.zeros(len(ch), dtype=np.float64)
        _check(_generate_block_async(self._h, _p(ch), len(ch), int(nsamp), float(fs), int(sample_size), _vp(host_ptr), _p(carr)))
        return carr

    def wait(self):
        _check(_wait(self._h))

    def generate_batch(self, desc, nsamp, fs, sample_size, device_ptr=None, host_ptr=None, carr_out=None):
        """carr_out: optional float64[nchan] array that receives the carrier phase after the
        last block (hand it back as block 0's carr_phase of the next batch to continue exactly).
        desc: a CHAN_DTYPE array [nblocks][nchan] (pageable or page-locked: a view is passed as it is), or a tuple
        (pointer, nblocks, nchan) for gpsiq_chan_t rows that lie in device or page-locked memory."""
        if isinstance(desc, tuple):
            dp, nb, nc = _vp(desc[0]), int(desc[1]), int(desc[2])
        else:
            desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
            nb, nc = desc.shape
            dp = _p(desc)
        co = None if carr_out is None else _p(carr_out)
        if host_ptr is not None:      # caller-owned host buffer (e.g. pinned), nb*2*nsamp elements
            _check(_generate_batch(self._h, dp, nb, nc, int(nsamp), float(fs), int(sample_size), _vp(host_ptr), 0, co))
            return None
        if device_ptr is not None:
            _check(_generate_batch(self._h, dp, nb, nc, int(nsamp), float(fs), int(sample_size), _vp(device_ptr), 1, co))
            return None
        out = np.zeros((nb, 2 * nsamp), dtype=elem_dtype(sample_size))
        _check(_generate_batch(self._h, dp, nb, nc, int(nsamp), float(fs), int(sample_size), _p(out), 0, co))
        return out

    def device_eval_host_ms(self):
        """Host time the last device-evaluated batch spent on descriptors (pack of pageable rows, repair, the host walker's share)."""
        return float(_device_eval_host_ms(self._h))

    def generate_quantized(self, q, nsamp, sample_size, device_ptr=None, host_ptr=None):
        """One shard of a time-sharded run: a slice of quantize_blocks()' output -> IQ elements."""
        q = np.ascontiguousarray(q, dtype=QCHAN_DTYPE)
        nb, nc = q.shape
        if host_ptr is not None:
            _check(_generate_quantized(self._h, _p(q), nb, nc, int(nsamp), int(sample_size), _vp(host_ptr), 0))
            return None
        if device_ptr is not None:
            _check(_generate_quantized(self._h, _p(q), nb, nc, int(nsamp), int(sample_size), _vp(device_ptr), 1))
            return None
        out = np.zeros((nb, 2 * nsamp), dtype=elem_dtype(sample_size))
        _check(_generate_quantized(self._h, _p(q), nb, nc, int(nsamp), int(sample_size), _p(out), 0))
        return out

    # -- resident-descriptor path (device buffers) --
    def generate_seeded(self, desc, nsamp, fs, sample_size, carr_start, device_ptr=None, host_ptr=None):
        """gpsiq_generate_seeded: GPSIQ_NCO_REFERENCE render of blocks whose start states are known (rows of reference_chain)."""
        desc = np.ascontiguousarray(desc, dtype=CHAN_DTYPE)
        nb, nc = desc.shape
        st = np.ascontiguousarray(carr_start, dtype=np.float64)
        assert st.shape == (nb, nc)
        if device_ptr is not None:
            _check(_generate_seeded(self._h, _p(desc), nb, nc, int(nsamp), float(fs), int(sample_size), _p(st), _vp(device_ptr), 1))
            return None
        out = None
        if host_ptr is None:
            out = np.zeros((nb, 2 * nsamp), dtype=elem_dtype(sample_size))
            host_ptr = out.ctypes.data
        _check(_generate_seeded(self._h, _p(desc), nb, nc, int(nsamp), float(fs), int(sample_size), _p(st), _vp(host_ptr), 0))
        return out

    def set_descriptors(self, q):
        q = np.ascontiguousarray(q, dtype=QCHAN_DTYPE)
        nb, nc = q.shape
        _check(_set_descriptors(self._h, _p(q), nb, nc))

    def launch(self, block0, nblocks, nsamp, sample_size, device_ptr, block_stride, stream=None, variant=0):
        _check(_launch(self._h, int(block0), int(nblocks), int(nsamp), int(sample_size), _vp(device_ptr),
                       int(block_stride), _vp(stream or 0), int(variant)))

    def synchronize(self, stream=None):
        _check(_synchronize(self._h, _vp(stream or 0)))

    def time_launches(self, block0, nblocks, nsamp, sample_size, device_ptr, block_stride, iters,
                      stream=None, variant=0):
        ms = C.c_float(0.0)
        _check(_time_launches(self._h, int(block0), int(nblocks), int(nsamp), int(sample_size), _vp(device_ptr),
                              int(block_stride), _vp(stream or 0), int(variant), int(iters),
                              C.byref(ms)))
        return ms.value
