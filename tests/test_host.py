"""Host half of libgpsiq (no GPU): tables, quantiser, fifo hand-off rules — against the
oracle and the committed reference captures."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import CHAN_DTYPE, QCHAN_DTYPE, SC08, SC16, SINK_HACKRF, SINK_IQFILE, SINK_PLUTOSDR
from gpsiq.scenario import synth_blocks

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tables_equal_oracle_and_golden(oracle):
    cos, sin = gpsiq.carrier_table()
    s, c = oracle.tables()
    assert (cos == c).all() and (sin == s).all()
    z = np.load(os.path.join(GOLD, "tables.npz"))
    assert (sin == z["sin512"]).all() and (cos == z["cos512"]).all()
    for prn in range(1, 33):
        ca = gpsiq.prn_code(prn)
        assert (ca == oracle.codegen(prn)).all()
        assert (np.packbits(ca, bitorder="little") == z["prn_packed"][prn - 1]).all()
    for bad in (0, 33, -1):
        with pytest.raises(gpsiq.GpsiqError):
            gpsiq.prn_code(bad)


@pytest.mark.parametrize("fs", [2.6e6, 3e6, 1e7, 2.5e7, 1.5e6])
def test_quantiser_equals_oracle(oracle, fs):
    d = synth_blocks(6, 16, seed=int(fs) % 1000)
    d["prn"][2:4, 3] = 0
    d["prn"][4:, 3] = 17
    ns = int(fs) // 10
    q, carry = gpsiq.quantize_blocks(d, fs, ns)
    assert q.tobytes() == oracle.quantize_blocks(d, fs, ns).tobytes()
    # exact carry: block k+1 starts where block k ended, modulo 2^59
    for b in range(1, 6):
        for c in range(16):
            if d[b, c]["prn"] > 0 and d[b, c]["prn"] == d[b - 1, c]["prn"]:
                want = (int(q[b - 1, c]["carr_phase"]) + int(q[b - 1, c]["carr_step"]) * ns) % (1 << 59)
                assert int(q[b, c]["carr_phase"]) == want


def test_quantiser_definition():
    """The rules of include/gpsiq.h spelled out with Python integers/Fractions."""
    from fractions import Fraction
    d = synth_blocks(1, 5, seed=9)[0]
    fs, ns = 2.6e6, 260000
    q, carry = gpsiq.quantize(d, fs, ns)
    delt = 1.0 / fs
    for c in range(5):
        assert int(q[c]["carr_step"]) == round(Fraction(float(d[c]["f_carr"]) * delt) * 2 ** 59)
        assert int(q[c]["code_step"]) == round(Fraction(float(d[c]["f_code"]) * delt) * 2 ** 56)
        assert int(q[c]["carr_phase"]) == int(Fraction(float(d[c]["carr_phase"])) * 2 ** 59)
        chip0 = int(d[c]["code_phase"])
        assert int(q[c]["chip0"]) == chip0
        assert int(q[c]["code_frac"]) == int((Fraction(float(d[c]["code_phase"])) - chip0) * 2 ** 56)
        bits = []
        w, b = int(d[c]["iword"]), int(d[c]["ibit"])
        for _ in range(8):
            bits.append((int(d[c]["dwrd"][w]) >> (29 - b)) & 1)
            b += 1
            if b == 30:
                b, w = 0, w + 1
        nb = (int(d[c]["icode"]) + (chip0 + ((int(q[c]["code_frac"]) + (ns - 1) * int(q[c]["code_step"])) >> 56)) // 1023) // 20 + 1
        assert int(q[c]["nav_bits"]) == sum(bit << i for i, bit in enumerate(bits[:nb]))
        assert int(carry[c]) == (int(q[c]["carr_phase"]) + ns * int(q[c]["carr_step"])) % (1 << 59)


def _chain_blocks_one_by_one(d, fs, ns, carry0=None):
    """gpsiq_quantize_batch's contract, restated with the single-block call."""
    nb, nc = d.shape
    q = np.zeros((nb, nc), dtype=QCHAN_DTYPE)
    carry = None if carry0 is None else np.array(carry0, dtype=np.uint64)
    for b in range(nb):
        cin = None
        if carry is not None:
            cin = carry.copy()
            if b > 0:
                for c in range(nc):
                    if d[b, c]["prn"] != d[b - 1, c]["prn"]:
                        cin[c] = np.uint64(int(np.floor(np.ldexp(float(d[b, c]["carr_phase"]), 59))))
        q[b], carry = gpsiq.quantize(d[b], fs, ns, cin)
    return q, carry


@pytest.mark.parametrize("nb,with_carry", [(1, False), (7, True), (3001, False), (3001, True)])
def test_batch_quantiser_is_the_block_quantiser_chained(nb, with_carry):
    """Threaded gpsiq_quantize_batch == gpsiq_quantize block after block (3001 blocks span
    many worker chunks), including slots that go idle and come back with another PRN."""
    fs, ns = 2.6e6, 260000
    d = synth_blocks(nb, 12, seed=77 + nb)
    if nb > 4:
        d["prn"][2:4, 3] = 0
        d["prn"][4:, 3] = 21
        d["prn"][nb // 2:, 7] = 30
    carry0 = (np.arange(12, dtype=np.uint64) * np.uint64(0x0123456789ABCDE)) % np.uint64(1 << 59) if with_carry else None
    q, end = gpsiq.quantize_blocks(d, fs, ns, carry0)
    q1, end1 = _chain_blocks_one_by_one(d, fs, ns, carry0)
    assert q.tobytes() == q1.tobytes()
    assert (end == end1).all()
    # and a second call gives the same bytes (the pool hands chunks out in a different order every time)
    q2, _ = gpsiq.quantize_blocks(d, fs, ns, carry0)
    assert q2.tobytes() == q.tobytes()


def test_batch_quantiser_reports_the_offending_block():
    d = synth_blocks(500, 4, seed=5)
    d["icode"][321, 2] = 20
    with pytest.raises(gpsiq.GpsiqError) as e:
        gpsiq.quantize_blocks(d, 2.6e6, 260000)
    assert e.value.code == -2 and "block 321" in str(e.value)
    q, end = gpsiq.quantize_blocks(d[:0], 2.6e6, 260000)        # an empty timeline is fine
    assert q.shape == (0, 4) and (end == 0).all()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("nblocks", [0, 1, 7, 8, 299, 35999])
def test_shard_range_is_a_contiguous_balanced_partition(nblocks, world):
    edges = [gpsiq.shard_range(nblocks, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == nblocks
    for r in range(1, world):
        assert edges[r][0] == edges[r - 1][1]
    sizes = [e - b for b, e in edges]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    for bad in [(nblocks, world, world), (nblocks, -1, world), (-1, 0, world), (nblocks, 0, 0)]:
        with pytest.raises(gpsiq.GpsiqError):
            gpsiq.shard_range(*bad)


@pytest.mark.parametrize("field,value", [("prn", 33), ("code_phase", 1023.0), ("code_phase", -0.5), ("carr_phase", 1.0),
                                         ("carr_phase", -0.1), ("iword", 60), ("ibit", 30), ("icode", 20),
                                         ("f_carr", 2.0e6), ("f_code", 6.0e6), ("f_code", 0.0), ("f_carr", float("nan")),
                                         ("gain", float("nan")), ("gain", float("inf")), ("gain", -float("inf")),
                                         ("gain", 1e300), ("gain", 4.0e6), ("gain", -4.0e6)])
def test_quantiser_rejects_out_of_range(field, value):
    d = synth_blocks(1, 3, seed=2)[0]
    d[field][1] = value
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.quantize(d, 2.6e6, 1000)


def test_quantiser_takes_gains_up_to_the_bound_and_every_entry_point_checks_it():
    d = synth_blocks(2, 3, seed=2)
    d["gain"][0, 1] = 3.9e6
    d["gain"][1, 2] = -3.9e6
    gpsiq.quantize(d[0], 2.6e6, 1000)
    gpsiq.quantize_blocks(d, 2.6e6, 1000)
    d["gain"][1, 0] = float("nan")
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.quantize_blocks(d, 2.6e6, 1000)
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.reference_blocks(d, 2.6e6, 1000)


def test_seeding_a_shard_in_place_refuses_a_view_it_would_have_to_copy():
    from gpsiq.shard import seed_own_shard
    q, _ = gpsiq.quantize_blocks(synth_blocks(6, 4, seed=5), 2.6e6, 1000)
    with pytest.raises(ValueError):
        seed_own_shard(q[:, ::2], 1000, 0, 1, lambda b: [b])          # a strided view: not seeded in place
    own = q[2:5]                                                      # a contiguous slice of a larger timeline is fine
    before = own.copy()
    assert seed_own_shard(own, 1000, 0, 1, lambda b: [b]) is own and np.array_equal(own, before)


def test_quantiser_rejects_running_off_the_word_buffer():
    d = synth_blocks(1, 1, seed=3)[0]
    d["iword"], d["ibit"], d["icode"] = 59, 29, 19      # the block needs dwrd[60]
    with pytest.raises(gpsiq.GpsiqError):
        gpsiq.quantize(d, 2.6e6, 260000)
    with pytest.raises(gpsiq.GpsiqError):                # more than 32 nav bits in one block
        gpsiq.quantize(synth_blocks(1, 1, seed=4)[0], 2.6e6, 2600000)


# ---- fifo hand-off (gps.c:2839-2865) --------------------------------------------
class IqBuf(C.Structure):
    pass


IqBuf._fields_ = [("data8", C.c_void_p), ("data16", C.c_void_p), ("totalLength", C.c_uint),
                  ("validLength", C.c_uint), ("next", C.POINTER(IqBuf))]
ACQ = C.CFUNCTYPE(C.c_void_p, C.c_void_p)
ENQ = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(IqBuf))


class Chunker(C.Structure):
    _fields_ = [("acquire", ACQ), ("enqueue", ENQ), ("user", C.c_void_p), ("cur", C.POINTER(IqBuf)),
                ("sink_kind", C.c_int), ("sample_size", C.c_int)]


def run_chunker(sink, ss, blocks, buf_len, in_place=False):
    lib = C.CDLL(gpsiq.LIB_PATH)
    dt = np.int8 if ss == SC08 else np.int16
    pool, enq = [], []

    def acquire(_):
        arr = np.zeros(buf_len, dtype=dt)
        b = IqBuf()
        if ss == SC08:
            b.data8 = arr.ctypes.data
        else:
            b.data16 = arr.ctypes.data
        b.totalLength, b.validLength = buf_len, 0
        pool.append((b, arr))
        return C.addressof(b)

    def enqueue(_, p):
        for b, arr in pool:
            if C.addressof(b) == C.addressof(p.contents):
                enq.append(arr[: b.validLength].copy())
                return
        raise AssertionError("unknown buffer")

    ck = Chunker()
    a, e = ACQ(acquire), ENQ(enqueue)
    lib.gpsiq_chunker_init.argtypes = [C.c_void_p, C.c_int, C.c_int, ACQ, ENQ, C.c_void_p]
    lib.gpsiq_chunker_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert lib.gpsiq_chunker_init(C.byref(ck), sink, ss, a, e, None) == 0
    total = 0
    lib.gpsiq_chunker_reserve.argtypes = [C.c_void_p, C.c_size_t]
    lib.gpsiq_chunker_reserve.restype = C.c_void_p
    lib.gpsiq_chunker_commit.argtypes = [C.c_void_p, C.c_size_t]
    for blk in blocks:
        where = lib.gpsiq_chunker_reserve(C.byref(ck), blk.size) if in_place else None
        if where:                      # the producer writes into the fifo buffer itself
            C.memmove(where, blk.ctypes.data, blk.nbytes)
            n = lib.gpsiq_chunker_commit(C.byref(ck), blk.size)
        else:
            n = lib.gpsiq_chunker_push(C.byref(ck), blk.ctypes.data, blk.size)
        assert n >= 0, gpsiq._last_error()
        total += n
    assert total == len(enq)
    return enq


@pytest.mark.parametrize("sink", [SINK_IQFILE, SINK_HACKRF, SINK_PLUTOSDR])
@pytest.mark.parametrize("ss", [SC08, SC16])
def test_chunker_follows_reference_rules(oracle, sink, ss):
    rng = np.random.default_rng(1)
    nelem, nb = 600000, 3
    dt = np.int8 if ss == SC08 else np.int16
    blocks = [rng.integers(-100, 100, size=nelem).astype(dt) for _ in range(nb)]
    buf_len = 262144 if sink == SINK_HACKRF else nelem          # sdr_hackrf.c:215 / sdr_iqfile.c:59
    enq = run_chunker(sink, ss, blocks, buf_len)
    plan = oracle.chunk_plan(sink, nelem, nb)
    assert [len(x) for x in enq] == list(plan)
    flat = np.concatenate(blocks)
    assert np.array_equal(np.concatenate(enq), flat[: plan.sum()])


@pytest.mark.parametrize("sink", [SINK_IQFILE, SINK_HACKRF, SINK_PLUTOSDR])
def test_chunker_in_place_handoff_equals_push(sink):
    """reserve/commit (block written straight into the fifo buffer) enqueues exactly what push
    does; HackRF chunks and too-small buffers decline the reservation instead."""
    rng = np.random.default_rng(3)
    nelem, nb = 520000, 4
    blocks = [rng.integers(-100, 100, size=nelem).astype(np.int16) for _ in range(nb)]
    buf_len = 262144 if sink == SINK_HACKRF else nelem
    a = run_chunker(sink, SC16, blocks, buf_len, in_place=True)
    b = run_chunker(sink, SC16, blocks, buf_len, in_place=False)
    assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    if sink != SINK_HACKRF:
        assert [len(x) for x in a] == [nelem] * nb
        # a buffer smaller than the block: no reservation, and push reports the misfit
        with pytest.raises(AssertionError):
            run_chunker(sink, SC16, blocks, nelem - 2, in_place=True)


def test_chunker_matches_reference_capture():
    z = np.load(os.path.join(GOLD, "hackrf_chunks.npz"))
    rng = np.random.default_rng(2)
    nb, nelem = z["desc"].shape[0], 2 * int(z["nsamp"])
    blocks = [rng.integers(-100, 100, size=nelem).astype(np.int8) for _ in range(nb)]
    enq = run_chunker(SINK_HACKRF, SC08, blocks, 262144)
    assert [len(x) for x in enq] == list(z["chunk_len"])


def test_a_rank_quantises_only_its_own_blocks():
    """gpsiq_shard_carry / gpsiq_shard_seed: every rank quantises its own rows, learns its carrier seed from
    32 bytes per channel and rank, and ends up with exactly the rows of the whole-timeline quantiser --
    with satellites setting, slots re-allocated, unused slots, empty ranks."""
    import gpsiq
    from gpsiq import quantize_blocks, shard_carry, shard_range, shard_seed
    from gpsiq.shard import quantize_own_shard
    rng = np.random.default_rng(7)
    for trial in range(60):
        nb, nc, world, ns = int(rng.integers(1, 40)), int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(1, 5000))
        d = synth_blocks(nb, nc, seed=trial)
        for _ in range(int(rng.integers(0, 6))):
            b, c = int(rng.integers(0, nb)), int(rng.integers(0, nc))
            d["prn"][b:, c] = int(rng.integers(0, 33))
            d["carr_phase"][b:, c] = rng.random()
        whole, _ = quantize_blocks(d, 2.6e6, ns)
        own, carry = [], []
        for r in range(world):
            b0, b1 = shard_range(nb, r, world)
            q = quantize_blocks(d[b0:b1], 2.6e6, ns)[0] if b1 > b0 else np.zeros((0, nc), dtype=whole.dtype)
            own.append(np.ascontiguousarray(q))
            carry.append(shard_carry(q, ns))
        for r in range(world):
            b0, b1 = shard_range(nb, r, world)
            if b1 > b0:
                shard_seed(own[r], ns, np.stack(carry), r)
                assert np.array_equal(own[r], whole[b0:b1]), (trial, r, world, nb)
    # the recipe as one call, with a stand-in for the exchange
    d = synth_blocks(9, 5, seed=3)
    whole, _ = quantize_blocks(d, 2.6e6, 777)
    recs = [shard_carry(quantize_blocks(d[slice(*shard_range(9, r, 3))], 2.6e6, 777)[0], 777).tobytes() for r in range(3)]
    for r in range(3):
        b0, b1 = shard_range(9, r, 3)
        got = quantize_own_shard(d[b0:b1], 2.6e6, 777, r, 3, lambda mine: recs)
        assert np.array_equal(got, whole[b0:b1])
