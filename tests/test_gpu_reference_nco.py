"""GPSIQ_NCO_REFERENCE on the GPU: the HIP path reproduces the reference's double-accumulator loop
element for element over whole runs -- compared DIRECTLY with the reference's own compiled lines
(oracle/_ref/libgpsref.so travels to the GPU box), with the oracle's restatement of them, and with
the committed captures, incl. the t1diff_* blocks on which the fixed-point model is known to differ."""
import hashlib

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import NCO_FIXED, NCO_REFERENCE, SC08, SC16, SINK_IQFILE
from gpsiq.scenario import synth_blocks
from test_golden import CASES, check_fixed_block_against_golden, load_case, start_state
from test_gpu_parity import VARIANTS, run_device

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU path in libgpsiq"
    c = gpsiq.Context(0)
    c.set_nco_mode(NCO_REFERENCE)
    yield c
    c.close()


def float_run(oracle, d, fs, ss):
    """The reference loop over consecutive blocks via the oracle's restatement (pinned to the reference
    in test_oracle_vs_ref.py): used where oracle/_ref is absent."""
    ns = int(fs) // 10
    out, carr, prev = [], None, None
    for b in range(len(d)):
        db = d[b].copy()
        if b:
            keep = (prev == db["prn"]) & (db["prn"] > 0)
            db["carr_phase"] = np.where(keep, carr, db["carr_phase"])
        o, carr = oracle.block_float(db, ns, fs, ss)
        out.append(o)
        prev = db["prn"].copy()
    return np.stack(out), carr


@pytest.mark.parametrize("name", CASES)
def test_golden_blocks_reference_nco(rctx, name):
    """Every committed capture as ONE batch from the captured descriptors alone: the library carries the
    carrier like the reference's accumulator, so all blocks hash to the reference's bytes -- also the
    t1diff_* blocks, where the fixed-point model differs in the recorded elements."""
    g = load_case(name)
    carr = np.zeros(g["desc"].shape[1])
    got = rctx.generate_batch(g["desc"], g["nsamp"], g["fs"], g["ss"], carr_out=carr)
    for b in range(len(g["sha"])):
        assert hashlib.sha256(got[b].tobytes()).hexdigest() == g["sha"][b], (name, b)
    act = g["desc"][-1]["prn"] > 0
    assert (carr[act] == g["carr"][-1][act]).all()


@pytest.mark.parametrize("name", [c for c in CASES if c.startswith("t1diff_")])
def test_known_mismatch_blocks_fixed_nco(rctx, name):
    """The same captures in the default model: GPU == reference except exactly at the recorded elements."""
    g = load_case(name)
    rctx.set_nco_mode(NCO_FIXED)
    for b in range(len(g["sha"])):
        q, _ = gpsiq.quantize(start_state(g, b), g["fs"], g["nsamp"])
        rctx.set_descriptors(q[None, :])
        check_fixed_block_against_golden(g, b, run_device(rctx, q[None, :], g["nsamp"], g["ss"], "auto")[0])


@pytest.mark.parametrize("fs,nchan,ss,nb,seed", [(2600000, 16, SC08, 299, 20250215),     # BASELINE config 1 length
                                                 (2600000, 12, SC16, 60, 7),
                                                 (3000000, 12, SC08, 40, 8),
                                                 (10000000, 16, SC16, 24, 9),
                                                 (25000000, 16, SC16, 12, 10),
                                                 (25000000, 16, SC08, 12, 11)])
def test_whole_run_equals_the_reference_itself(rctx, ref, fs, nchan, ss, nb, seed):
    """T2 = 0: a whole run through gpsiq_generate_batch against the reference's own loop run here on the
    host (carrier carried by the reference's double accumulator), every element of every block."""
    d = synth_blocks(nb, nchan, seed=seed)
    want, _, carr_ref = ref.run_blocks(d, fs, ss, SINK_IQFILE)
    carr = np.zeros(nchan)
    got = rctx.generate_batch(d, fs // 10, float(fs), ss, carr_out=carr)
    assert np.array_equal(got.reshape(-1), want)
    assert np.array_equal(carr, carr_ref[-1])


def test_block_at_a_time_equals_the_reference_loop(rctx, oracle):
    """The drop-in pattern: gpsiq_generate_block once per 0.1 s block, the carr_phase it hands out passed
    back in (what the patched gps thread does) == the float loop; slots going out of view and being
    re-allocated re-seed from the descriptor like allocateChannel (gps.c:2208-2214)."""
    fs, nb, nc = 2.6e6, 12, 10
    d = synth_blocks(nb, nc, seed=61)
    d["prn"][4:, 3] = 0
    d["prn"][7:, 5] = 29
    d["carr_phase"][7:, 5] = 0.3125
    for ss in (SC08, SC16):
        want, carr_want = float_run(oracle, d, fs, ss)
        carr, prev = None, None
        for b in range(nb):
            db = d[b].copy()
            if b:
                keep = (prev == db["prn"]) & (db["prn"] > 0)
                db["carr_phase"] = np.where(keep, carr, db["carr_phase"])
            out, carr = rctx.generate_block(db, int(fs) // 10, fs, ss)
            assert np.array_equal(out, want[b]), (ss, b)
            prev = db["prn"].copy()
        act = d[-1]["prn"] > 0
        assert np.array_equal(carr[act], carr_want[act])


@pytest.mark.parametrize("variant", VARIANTS)
def test_patches_on_the_resident_path_every_kernel(rctx, oracle, variant):
    """gpsiq_reference_batch -> gpsiq_set_descriptors + gpsiq_set_patches -> gpsiq_launch (any kernel
    variant, also a sub-range of the blocks) == the float loop."""
    fs, nb, nc, ss = 25000000, 2, 16, SC16
    d = synth_blocks(nb, nc, seed=3032)                  # 4 + 1 patches (the t1diff_25M_16ch_sc08 scenario)
    q, patches, _ = gpsiq.reference_blocks(d, fs, fs // 10)
    assert len(patches) >= 2
    want, _ = float_run(oracle, d, float(fs), ss)
    rctx.set_descriptors(q)
    rctx.set_patches(patches)
    got = run_device(rctx, q, fs // 10, ss, variant)
    assert np.array_equal(got, want)
    sub = run_device(rctx, q, fs // 10, ss, variant, block0=1, nblocks=1)
    assert np.array_equal(sub, want[1:2])
    rctx.set_patches(patches[:0])                                   # cleared: back to the plain closed form
    plain = run_device(rctx, q, fs // 10, ss, variant)
    assert np.array_equal(plain, np.stack([oracle.block_fixed(q[b], fs // 10, ss, seq=True) for b in range(nb)]))
    assert not np.array_equal(plain, want)


def test_reference_nco_edge_cases(rctx, oracle):
    """Zero / tiny / negative Doppler, phases on boundaries, unused slots, short and ragged blocks."""
    d = synth_blocks(3, 8, seed=5)
    d["f_carr"][:] = [0.0, -4999.7, 4999.7, 1e-9, -1e-9, 0.25, -1234.5, 3e-14]
    d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
    d["carr_phase"][:] = [0.0, 0.0, 0.999999999999, 0.5, 0.5, 0.0, 1e-300, 0.75]
    d["code_phase"][0, :3] = [0.0, 1022.9999999999, 511.99999999999994]
    d["prn"][:, 6] = 0
    for fs, ns in ((2.6e6, 70001), (25e6, 120000), (2.6e6, 1), (2.6e6, 65), (1.023e6, 30000)):
        want, carr, prev = [], None, None
        for b in range(3):
            db = d[b].copy()
            if b:
                db["carr_phase"] = np.where((prev == db["prn"]) & (db["prn"] > 0), carr, db["carr_phase"])
            o, carr = oracle.block_float(db, ns, fs, SC16)
            want.append(o)
            prev = db["prn"].copy()
        got = rctx.generate_batch(d, ns, fs, SC16)
        assert np.array_equal(got, np.stack(want)), (fs, ns)


def test_mode_switch_and_patch_validation(rctx):
    d = synth_blocks(2, 4, seed=9)
    q, patches, _ = gpsiq.reference_blocks(d, 2.6e6, 1000)
    rctx.set_descriptors(q)
    bad = np.zeros(1, dtype=patches.dtype)
    bad["block"] = 5
    with pytest.raises(gpsiq.GpsiqError):
        rctx.set_patches(bad)
    bad["block"], bad["lut"] = 0, 600
    with pytest.raises(gpsiq.GpsiqError):
        rctx.set_patches(bad)
    # a slot counts the block's ACTIVE channels: with one of four channels unused, slot 3 does not exist
    q2 = q.copy()
    q2["prn"][1, 2] = 0
    rctx.set_descriptors(q2)
    bad["block"], bad["lut"], bad["slot"] = 1, 7, 3
    with pytest.raises(gpsiq.GpsiqError):
        rctx.set_patches(bad)
    bad["slot"] = 2
    rctx.set_patches(bad)
    bad["block"] = 0
    bad["slot"] = 3
    rctx.set_patches(bad)
    rctx.set_patches(bad[:0])
    with pytest.raises(gpsiq.GpsiqError):
        rctx.set_nco_mode(7)
    rctx.set_nco_mode(NCO_FIXED)
    rctx.set_nco_mode(NCO_REFERENCE)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_reference_nco_fuzz(rctx, oracle, seed):
    """Random rates, block lengths, channel counts, Doppler ranges, dyadic steps (exact rounding ties), phases on powers
    of two and LUT boundaries, integer code phases, slots going out of view: gpsiq_generate_batch in
    GPSIQ_NCO_REFERENCE == the float loop, every element, and the phase handed out == the loop's."""
    rng = np.random.default_rng(seed)
    for _ in range(25):
        fs = float(rng.choice([1.2e6, 2.048e6, 2.6e6, 3e6, 4.092e6, 10e6, 16.368e6, 25e6]))
        ns, nc, nb, ss = int(rng.integers(1, 90000)), int(rng.integers(1, 17)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
        d = synth_blocks(nb, nc, seed=int(rng.integers(0, 1 << 30)), doppler_hz=float(rng.choice([5000.0, 10000.0, 50.0, 0.001])))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            k = rng.integers(8, 40, size=nc)
            d["f_carr"] = (fs * (rng.integers(1, 8, size=nc) * 2.0 - 7) / 2.0 ** k)[None, :]
        elif kind == 1:
            d["carr_phase"] = (rng.integers(0, 512, size=nc) / 512.0)[None, :]
        elif kind == 2:
            d["carr_phase"] = (2.0 ** -rng.integers(1, 60, size=nc).astype(float))[None, :]
        elif kind == 3:
            d["code_phase"] = np.floor(d["code_phase"])
        d["f_code"] = 1.023e6 + d["f_carr"] / 1540.0
        if rng.random() < 0.3:
            d["prn"][nb // 2:, 0] = 0
        want, carr_want = float_run_ns(oracle, d, fs, ns, ss)
        carr = np.zeros(nc)
        got = rctx.generate_batch(d, ns, fs, ss, carr_out=carr)
        assert np.array_equal(got, want), (fs, ns, nc, nb, ss, kind)
        act = d[-1]["prn"] > 0
        assert np.array_equal(carr[act], carr_want[act]), (fs, ns, kind)


def float_run_ns(oracle, d, fs, ns, ss):
    out, carr, prev = [], None, None
    for b in range(len(d)):
        db = d[b].copy()
        if b:
            db["carr_phase"] = np.where((prev == db["prn"]) & (db["prn"] > 0), carr, db["carr_phase"])
        o, carr = oracle.block_float(db, ns, fs, ss)
        out.append(o)
        prev = db["prn"].copy()
    return np.stack(out), carr


def test_asynchronous_block_calls_reference_nco(rctx, oracle):
    """gpsiq_generate_block_async in GPSIQ_NCO_REFERENCE: the host walks the block's carrier before anything
    is queued, so the phase handed out is final at once and the block's patches ride along on the stream.  Blocks queued
    back to back, more than the ring of four holds, a synchronous call in between: every buffer == the float loop, the
    phase after every call == the loop's.  25 Msps: this scenario has patched samples."""
    import torch
    fs, nb, nc = 25e6, 7, 16
    ns = int(fs) // 10
    d = synth_blocks(nb, nc, seed=3032)
    d["prn"][3:, 5] = 0
    for ss in (SC16, SC08):
        want, carr_want = float_run(oracle, d, fs, ss)
        bufs = [torch.zeros(2 * ns * ss, dtype=torch.uint8).pin_memory() for _ in range(nb)]
        carr, prev = None, None
        for b in range(nb):
            db = d[b].copy()
            if b:
                db["carr_phase"] = np.where((prev == db["prn"]) & (db["prn"] > 0), carr, db["carr_phase"])
            if b == 4:
                out, carr = rctx.generate_block(db, ns, fs, ss)
                bufs[b].numpy().view(out.dtype)[:] = out
            else:
                carr = rctx.generate_block_async(db, ns, fs, ss, bufs[b].data_ptr())
            prev = db["prn"].copy()
        rctx.wait()
        for b in range(nb):
            got = bufs[b].numpy().view(np.int8 if ss == SC08 else np.int16)
            assert np.array_equal(got, want[b]), (ss, b)
        act = d[-1]["prn"] > 0
        assert np.array_equal(carr[act], carr_want[act])


@pytest.mark.parametrize("pieces", ["1", "3", "0"])
def test_batch_walked_and_rendered_in_pieces(rctx, oracle, monkeypatch, pieces):
    """gpsiq_generate_batch in GPSIQ_NCO_REFERENCE walks the timeline in pieces and renders piece k under the walk of
    piece k+1: whatever the piece length (1 block, 3 blocks, the whole batch), host or device destination, the result is
    the float loop, every element, and the phase handed out is the loop's."""
    import torch
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", pieces)
    fs, nb, nc = 10e6, 8, 16
    ns = int(fs) // 10
    d = synth_blocks(nb, nc, seed=77)
    d["prn"][3:, 1] = 0
    d["prn"][5:, 1] = 21
    d["carr_phase"][5:, 1] = 0.3125
    for ss in (SC16, SC08):
        want, carr_want = float_run(oracle, d, fs, ss)
        carr = np.zeros(nc)
        got = rctx.generate_batch(d, ns, fs, ss, carr_out=carr)
        assert np.array_equal(got, want)
        act = d[-1]["prn"] > 0
        assert np.array_equal(carr[act], carr_want[act])
        buf = torch.zeros(nb * 2 * ns * ss, dtype=torch.uint8, device="cuda")
        rctx.generate_batch(d, ns, fs, ss, device_ptr=buf.data_ptr())
        torch.cuda.synchronize()
        dev = buf.cpu().numpy().view(np.int8 if ss == SC08 else np.int16).reshape(nb, 2 * ns)
        assert np.array_equal(dev, got)


def test_multi_device_reference_nco_in_pieces(oracle, monkeypatch):
    """gpsiq_generate_batch_multi in GPSIQ_NCO_REFERENCE: one thread walks, every device renders its range piece by piece
    as the walk reaches it (three contexts on the one GPU here) == the single-context batch == the float loop."""
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", "2")
    fs, nb, nc = 2.6e6, 13, 12
    ns = int(fs) // 10
    d = synth_blocks(nb, nc, seed=12)
    d["prn"][6:, 3] = 0
    ctxs = [gpsiq.Context(0) for _ in range(3)]
    try:
        ctxs[0].set_nco_mode(NCO_REFERENCE)
        want, carr_want = float_run(oracle, d, fs, SC16)
        carr = np.zeros(nc)
        got = gpsiq.generate_batch_multi(ctxs, d, ns, fs, SC16, carr_out=carr)
        assert np.array_equal(got, want)
        act = d[-1]["prn"] > 0
        assert np.array_equal(carr[act], carr_want[act])
    finally:
        for c in ctxs:
            c.close()


def _batch_digest_and_time(ctx, d, ns, fs, ss, buf, reps=3):
    import hashlib
    import time
    import torch
    best = float("inf")
    for _ in range(reps):
        t = time.perf_counter()
        ctx.generate_batch(d, ns, fs, ss, device_ptr=buf.data_ptr())
        best = min(best, time.perf_counter() - t)
    torch.cuda.synchronize()
    return hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest(), best


def test_two_contexts_walk_at_the_same_time_on_two_host_threads():
    """Two contexts of one process, both in GPSIQ_NCO_REFERENCE, called from two host threads at once: their walks share the
    library's worker pool (neither falls back to one thread, together they use no more threads than GPSIQ_THREADS allows), both
    results are bit-identical to the calls made one after the other, and running them together takes less than twice the
    longer one alone would suggest for a serial fallback (< 2 x the sequential pair)."""
    import threading
    import time
    import torch
    fs, nb, nc, ss = 2.6e6, 600, 16, SC08
    ns = int(fs) // 10
    pat = synth_blocks(64, nc, seed=20250215)
    d = [pat[np.arange(nb) % 64], pat[(np.arange(nb) + 17) % 64]]
    ctxs = [gpsiq.Context(0), gpsiq.Context(0)]
    try:
        bufs = [torch.zeros(nb * 2 * ns * ss, dtype=torch.uint8, device="cuda") for _ in range(2)]
        for c in ctxs:
            c.set_nco_mode(NCO_REFERENCE)
        alone = [_batch_digest_and_time(ctxs[k], d[k], ns, fs, ss, bufs[k]) for k in range(2)]
        import hashlib

        def run(k):
            ctxs[k].generate_batch(d[k], ns, fs, ss, device_ptr=bufs[k].data_ptr())
        best = float("inf")
        for _ in range(3):
            for b in bufs:
                b.zero_()
            torch.cuda.synchronize()
            th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
            t = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            best = min(best, time.perf_counter() - t)
            torch.cuda.synchronize()
            assert [hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest() for b in bufs] == [a[0] for a in alone]
        seq = alone[0][1] + alone[1][1]
        print("two reference-NCO batches: alone %.2f + %.2f ms, together %.2f ms" % (alone[0][1] * 1e3, alone[1][1] * 1e3, best * 1e3))
        assert best < 2.0 * seq
    finally:
        for c in ctxs:
            c.close()


def test_fewer_host_threads_than_channels_is_piece_major_and_exact(tmp_path):
    """GPSIQ_THREADS=2 with 16 channels (the share a rank gets when eight of them divide a 16-CPU host).  Host evaluation
    (GPSIQ_EVAL=host, rounds 4-5's path): the library never starts more than its share, works the timeline piece-major (all
    channels through piece k before piece k+1, so rendering still overlaps the host side) and says so under GPSIQ_TRACE; the
    result is bit-identical to the uncapped run and the call costs no more than the thread ratio allows (16 / 2 = 8 x the
    host-bound time, with slack).  Device evaluation (the default): the same bytes, and two threads cost next to nothing."""
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(%r, "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE, SC08
from gpsiq.scenario import synth_blocks
fs, nb = 2.6e6, 600
ns = int(fs) // 10
d = synth_blocks(64, 16, seed=20250215)[np.arange(nb) %% 64]
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
buf = torch.zeros(nb * 2 * ns, dtype=torch.uint8, device="cuda")
best = 1e9
for _ in range(3):
    t = time.perf_counter(); ctx.generate_batch(d, ns, fs, SC08, device_ptr=buf.data_ptr()); best = min(best, time.perf_counter() - t)
torch.cuda.synchronize()
import threading
print("RESULT", hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest(), best, threading.active_count(), len(os.listdir("/proc/self/task")))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for how in ("host", "device"):
        for threads in ("2", None):
            env = dict(os.environ, GPSIQ_TRACE="1", GPSIQ_EVAL=how)
            env.pop("GPSIQ_THREADS", None)
            if threads:
                env["GPSIQ_THREADS"] = threads
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0].split()
            out[how, threads] = (line[1], float(line[2]), r.stderr)
    assert len({v[0] for v in out.values()}) == 1                        # the same bytes, whoever evaluates and with however many threads
    assert "piece-major" in out["host", "2"][2] and "piece-major" not in out["host", None][2]
    assert "device evaluation" in out["device", "2"][2]
    print("reference-NCO batch, 600 blocks: host evaluation %.2f ms with all host threads, %.2f ms with GPSIQ_THREADS=2; device evaluation %.2f / %.2f ms"
          % (out["host", None][1] * 1e3, out["host", "2"][1] * 1e3, out["device", None][1] * 1e3, out["device", "2"][1] * 1e3))
    assert out["host", "2"][1] < 12.0 * out["host", None][1]
    assert out["device", "2"][1] < 1.5 * out["device", None][1] + 0.2e-3


@pytest.mark.parametrize("how", ["host", "device"])
def test_many_small_pieces_on_two_streams_over_four_descriptor_sets(rctx, oracle, monkeypatch, how):
    """The piece scheduling (two alternating streams, four descriptor sets taken in turn) under a piece size of 2 blocks: 21 blocks,
    both NCO models, host and device evaluation, every element the oracle's / the float loop's."""
    import torch
    monkeypatch.setenv("GPSIQ_EVAL", how)
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", "2")
    fs, nb, nc = 2.6e6, 21, 7
    ns = 26000
    d = synth_blocks(nb, nc, seed=131)
    want, carr_want = float_run_ns(oracle, d, fs, ns, SC08)
    buf = torch.zeros(nb * 2 * ns, dtype=torch.uint8, device="cuda")
    carr = np.zeros(nc)
    rctx.generate_batch(d, ns, fs, SC08, device_ptr=buf.data_ptr(), carr_out=carr)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy().view(np.int8).reshape(nb, 2 * ns), want)
    assert np.array_equal(carr, carr_want)
    rctx.set_nco_mode(0)
    try:
        qo = oracle.quantize_blocks(d, fs, ns)
        buf.zero_()
        rctx.generate_batch(d, ns, fs, SC08, device_ptr=buf.data_ptr())
        torch.cuda.synchronize()
        got = buf.cpu().numpy().view(np.int8).reshape(nb, 2 * ns)
        for b in range(nb):
            assert np.array_equal(got[b], oracle.block_fixed(qo[b], ns, SC08)), b
    finally:
        rctx.set_nco_mode(NCO_REFERENCE)
