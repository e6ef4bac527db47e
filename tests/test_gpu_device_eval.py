"""The batch calls with quantiser, carrier chain and evaluation ON THE DEVICE (csrc/gpsiq_evaldev.cpp, gpsiq_eval_kernels.hip,
the lane code gpsiq_eval.h -- whose CPU twin is tests/eval_twin.cpp): the rendered bytes and the carried phases are those of the
host path (GPSIQ_EVAL=host: rounds 4-5, itself T2 = 0 against the reference's own loop) and of the reference itself, for both NCO
models, all three kinds of descriptor memory (pageable, page-locked, device-resident), timelines with slow blocks, Doppler
through zero, re-seeded and unused slots (the host walker repairs those slots' chains), and descriptors the quantiser refuses."""
import numpy as np
import pytest

import gpsiq
from gpsiq.abi import CHAN_DTYPE, NCO_FIXED, NCO_REFERENCE, SC08, SC16, SINK_IQFILE
from gpsiq.scenario import synth_blocks

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU path in libgpsiq"
    c = gpsiq.Context(0)
    yield c
    c.close()


def rough_timeline(nb, nc, seed):
    """synth_blocks with what breaks a certified map: Doppler through zero (slow blocks, a sign change), a slot that gets another
    satellite, a slot unused for a while, an exact-tie addend."""
    d = synth_blocks(nb, nc, seed=seed)
    rng = np.random.default_rng(seed)
    b = np.arange(nb)
    d["f_carr"][:, 1] = (b - nb * 0.4) * 1.5 + rng.uniform(-0.02, 0.02, nb)        # through zero at 40 % of the timeline
    d["f_code"][:, 1] = 1.023e6 + d["f_carr"][:, 1] / 1540.0
    cut = nb // 3
    d["prn"][cut:, 2] = 1 + (int(d["prn"][0, 2]) % 32)
    d["prn"][nb // 2: nb // 2 + 9, 3] = 0
    if nc > 5:
        d["f_carr"][:, 5] = 2600000.0 / 1024.0                                    # c = 2^-10 exactly: every binade an exact tie
        d["f_code"][:, 5] = 1.023e6 + d["f_carr"][:, 5] / 1540.0
    return d


def render(ctx, d, ns, fs, ss, mode, how, monkeypatch, kind="pageable"):
    import torch
    monkeypatch.setenv("GPSIQ_EVAL", how)
    ctx.set_nco_mode(mode)
    nb, nc = d.shape
    carr = np.zeros(nc)
    out = torch.zeros(nb * 2 * ns * ss, dtype=torch.uint8, device="cuda")
    keep = None
    if kind == "pageable":
        src = d
    elif kind == "pinned":
        keep = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).pin_memory()
        src = (keep.data_ptr(), nb, nc)
    else:
        keep = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).cuda()
        src = (keep.data_ptr(), nb, nc)
    ctx.generate_batch(src, ns, fs, ss, device_ptr=out.data_ptr(), carr_out=carr)
    torch.cuda.synchronize()
    del keep
    return out.cpu().numpy(), carr


@pytest.mark.parametrize("mode", [NCO_REFERENCE, NCO_FIXED])
@pytest.mark.parametrize("fs,ns,nb,nc,ss", [(2.6e6, 260000, 150, 16, SC08), (2.6e6, 26000, 700, 12, SC16), (10e6, 1000000, 70, 16, SC16),
                                            (25e6, 2500000, 60, 16, SC08),
                                            (2.6e6, 26000, 2500, 16, SC08), (2.6e6, 13000, 3100, 7, SC16)])    # (several rounds of the 1 024-block scans)
def test_device_evaluation_equals_the_host_path(ctx, monkeypatch, mode, fs, ns, nb, nc, ss):
    d = synth_blocks(nb, nc, seed=int(fs) % 1000 + nb)
    before = gpsiq.device_eval_stats()
    got, carr = render(ctx, d, ns, fs, ss, mode, "device", monkeypatch)
    after = gpsiq.device_eval_stats()
    assert after[0] == before[0] + 1 and after[5] == before[5], "the device path did not take the call"
    want, carr_want = render(ctx, d, ns, fs, ss, mode, "host", monkeypatch)
    assert gpsiq.device_eval_stats()[0] == after[0]
    assert np.array_equal(got, want)
    assert carr.tobytes() == carr_want.tobytes()


@pytest.mark.parametrize("mode", [NCO_REFERENCE, NCO_FIXED])
@pytest.mark.parametrize("kind", ["pinned", "device"])
def test_descriptors_in_page_locked_and_device_memory(ctx, monkeypatch, mode, kind):
    fs, ns, nb, nc, ss = 2.6e6, 52000, 400, 16, SC08
    d = rough_timeline(nb, nc, 17)
    got, carr = render(ctx, d, ns, fs, ss, mode, "device", monkeypatch, kind)
    want, carr_want = render(ctx, d, ns, fs, ss, mode, "host", monkeypatch)
    assert np.array_equal(got, want)
    assert carr.tobytes() == carr_want.tobytes()
    assert ctx.device_eval_host_ms() >= 0.0


@pytest.mark.parametrize("fs,ns,nb,nc,ss,seed", [(2.6e6, 260000, 120, 16, SC08, 5), (2.6e6, 33333, 900, 9, SC16, 6), (25e6, 2500000, 50, 16, SC16, 7)])
def test_slots_the_host_walker_repairs(ctx, monkeypatch, fs, ns, nb, nc, ss, seed):
    """Slow blocks, a sign change, an exact-tie addend: blocks without a usable map.  The device hands those slots' chains to the
    host walker (statistics), everything else stays on the device, and the bytes are the host path's."""
    d = rough_timeline(nb, nc, seed)
    before = gpsiq.device_eval_stats()
    got, carr = render(ctx, d, ns, fs, ss, NCO_REFERENCE, "device", monkeypatch)
    after = gpsiq.device_eval_stats()
    assert after[3] > before[3], "no slot was repaired: the timeline did not exercise the repair"
    assert after[5] == before[5]
    want, carr_want = render(ctx, d, ns, fs, ss, NCO_REFERENCE, "host", monkeypatch)
    assert np.array_equal(got, want)
    assert carr.tobytes() == carr_want.tobytes()


@pytest.mark.parametrize("fs,nchan,ss,nb,seed", [(2600000, 16, SC08, 299, 20250215), (10000000, 16, SC16, 60, 9), (25000000, 16, SC16, 48, 10)])
def test_device_evaluation_equals_the_reference_itself(ctx, ref, monkeypatch, fs, nchan, ss, nb, seed):
    """T2 = 0 with nothing of the exact mode left on the host: against the reference's own loop (oracle/_ref) run here."""
    monkeypatch.setenv("GPSIQ_EVAL", "device")
    ctx.set_nco_mode(NCO_REFERENCE)
    d = synth_blocks(nb, nchan, seed=seed)
    want, _, carr_ref = ref.run_blocks(d, fs, ss, SINK_IQFILE)
    carr = np.zeros(nchan)
    before = gpsiq.device_eval_stats()
    got = ctx.generate_batch(d, fs // 10, float(fs), ss, carr_out=carr)
    assert gpsiq.device_eval_stats()[0] == before[0] + 1
    assert np.array_equal(got.reshape(-1), want)
    assert np.array_equal(carr, carr_ref[-1])


def test_seeded_render_on_the_device(ctx, monkeypatch):
    """gpsiq_generate_seeded (the evaluation half of a time-sharded run) through the device evaluation == the whole-timeline call."""
    fs, ns, nb, nc, ss = 2.6e6, 26000, 300, 10, SC16
    d = rough_timeline(nb, nc, 23)
    starts, _, _ = gpsiq.reference_chain(gpsiq.chain_inputs(d), fs, ns)
    want, _ = render(ctx, d, ns, fs, ss, NCO_REFERENCE, "host", monkeypatch)
    monkeypatch.setenv("GPSIQ_EVAL", "device")
    ctx.set_nco_mode(NCO_REFERENCE)
    before = gpsiq.device_eval_stats()
    lo, hi = 70, 251
    got = ctx.generate_seeded(d[lo:hi], ns, fs, ss, starts[lo:hi])
    assert gpsiq.device_eval_stats()[0] == before[0] + 1
    assert np.array_equal(got.view(np.uint8).reshape(-1), want.reshape(nb, -1)[lo:hi].reshape(-1))


@pytest.mark.parametrize("mode", [NCO_REFERENCE, NCO_FIXED])
def test_a_refused_descriptor_in_a_late_piece(ctx, monkeypatch, mode):
    """The quantiser on the device refuses a descriptor: the call fails with the host quantiser's words, everything queued is
    drained, and the context goes on working."""
    monkeypatch.setenv("GPSIQ_PIECE_BLOCKS", "40")
    fs, ns, nb, nc = 2.6e6, 26000, 200, 6
    d = synth_blocks(nb, nc, seed=91)
    bad = d.copy()
    bad["f_code"][150, 2] = 0.0
    bad["icode"][170, 1] = 25
    monkeypatch.setenv("GPSIQ_EVAL", "device")
    ctx.set_nco_mode(mode)
    with pytest.raises(gpsiq.GpsiqError, match="block 150.*f_code"):
        ctx.generate_batch(bad, ns, fs, SC08)
    got, carr = render(ctx, d, ns, fs, SC08, mode, "device", monkeypatch)
    want, carr_want = render(ctx, d, ns, fs, SC08, mode, "host", monkeypatch)
    assert np.array_equal(got, want) and carr.tobytes() == carr_want.tobytes()


def test_a_continued_fixed_point_batch(ctx, monkeypatch):
    """Two device-evaluated calls in the fixed-point model, the second handed the phases the first gave out, == one call."""
    fs, ns, nb, nc, ss = 2.6e6, 26000, 260, 8, SC08
    d = synth_blocks(nb, nc, seed=77)
    whole, _ = render(ctx, d, ns, fs, ss, NCO_FIXED, "device", monkeypatch)
    monkeypatch.setenv("GPSIQ_EVAL", "device")
    ctx.set_nco_mode(NCO_FIXED)
    carr = np.zeros(nc)
    a = ctx.generate_batch(d[:130], ns, fs, ss, carr_out=carr)
    d2 = d[130:].copy()
    d2["carr_phase"][0] = carr
    b = ctx.generate_batch(d2, ns, fs, ss)
    assert np.array_equal(np.concatenate([a, b]).view(np.uint8).reshape(-1), whole)
