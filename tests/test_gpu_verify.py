"""GPSIQ_CHAIN_VERIFY=N, the run-time self-check of the certified maps of GPSIQ_NCO_REFERENCE: every N-th block that went through
its map is also walked serially from the same start state (the reference's own additions, gps.c:2821-2826) and must end on the
same double; a mismatch fails the call with GPSIQ_E_VERIFY (-6) and names block and slot.  The mechanism's failure mode would be
silent wrong bytes, so the test needs a map that IS wrong: a library built with -DGPSIQ_TEST_HOOKS (fault injection, never in the
shipped build) shifts one block's end offset on the device (device evaluation) or in the walkers' link (host path).  The same build
shortens the device evaluation's lists on request, for the one path of it no honest input reaches: lists that overflow -> the host path."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import gpsiq
from gpsiq.abi import NCO_REFERENCE, SC08
from gpsiq.scenario import synth_blocks
assert os.path.samefile(gpsiq.LIB_PATH, os.environ["GPSIQ_LIB"])
fs, ns, nb, nc = 2.6e6, 26000, 300, 8
d = synth_blocks(nb, nc, seed=44)
ctx = gpsiq.Context(0)
ctx.set_nco_mode(NCO_REFERENCE)
for how in ("device", "host"):
    os.environ["GPSIQ_EVAL"] = how
    os.environ["GPSIQ_CHAIN"] = "device"
    for k in ("GPSIQ_CHAIN_VERIFY", "GPSIQ_TEST_CORRUPT_MAP", "GPSIQ_TEST_CORRUPT_MAP_AT"):
        os.environ.pop(k, None)
    carr0 = np.zeros(nc)
    good = ctx.generate_batch(d, ns, fs, SC08, carr_out=carr0)
    # the self-check on, nothing wrong: same bytes, no error
    os.environ["GPSIQ_CHAIN_VERIFY"] = "1"
    carr1 = np.zeros(nc)
    again = ctx.generate_batch(d, ns, fs, SC08, carr_out=carr1)
    assert np.array_equal(good, again) and carr0.tobytes() == carr1.tobytes(), how
    # one map made wrong (block 123, slot 5): without the check the call succeeds and the carrier it hands out is another one
    del os.environ["GPSIQ_CHAIN_VERIFY"]
    os.environ["GPSIQ_TEST_CORRUPT_MAP"] = "123,5"
    os.environ["GPSIQ_TEST_CORRUPT_MAP_AT"] = str(123 * nc + 5)
    carr2 = np.zeros(nc)
    ctx.generate_batch(d, ns, fs, SC08, carr_out=carr2)
    assert carr2[5] != carr0[5] and np.array_equal(np.delete(carr2, 5), np.delete(carr0, 5)), (how, carr2, carr0)
    # with it, the call fails and says where; also when only every 7th block is looked at, if that block is one of them
    for every in ("1", str(7)):
        os.environ["GPSIQ_CHAIN_VERIFY"] = every
        try:
            ctx.generate_batch(d, ns, fs, SC08)
            caught = None
        except gpsiq.GpsiqError as e:
            caught = e
        hit = (123 + 5) % int(every) == 0
        if hit:
            assert caught is not None and caught.code == -6 and "block 123" in str(caught) and "slot 5" in str(caught), (how, every, caught)
        else:
            assert caught is None, (how, every, caught)
    print(how, "verify ok")
# the lists of the device evaluation too short for the call's patches (GPSIQ_TEST_LIST_CAP, a hook of this build): the call notices,
# renders again on the host path, and the bytes and the carrier are the ones it always gives
for k in ("GPSIQ_CHAIN_VERIFY", "GPSIQ_TEST_CORRUPT_MAP", "GPSIQ_TEST_CORRUPT_MAP_AT", "GPSIQ_CHAIN"):
    os.environ.pop(k, None)
fs2, ns2, nb2 = 25e6, 2500000, 50
d2 = synth_blocks(nb2, nc, seed=45)
os.environ["GPSIQ_EVAL"] = "host"
carr_h = np.zeros(nc)
want = ctx.generate_batch(d2, ns2, fs2, SC08, carr_out=carr_h)
os.environ["GPSIQ_EVAL"] = "device"
s0 = gpsiq.device_eval_stats()
carr_d = np.zeros(nc)
got = ctx.generate_batch(d2, ns2, fs2, SC08, carr_out=carr_d)
s1 = gpsiq.device_eval_stats()
assert s1[4] - s0[4] > 8 and s1[5] == s0[5], (s0, s1)                      # patches there are, and the lists held them
assert np.array_equal(got, want) and carr_d.tobytes() == carr_h.tobytes()
os.environ["GPSIQ_TEST_LIST_CAP"] = "3"
carr_f = np.zeros(nc)
got = ctx.generate_batch(d2, ns2, fs2, SC08, carr_out=carr_f)
s2 = gpsiq.device_eval_stats()
assert s2[5] == s1[5] + 1, (s1, s2)                                         # fell back
assert np.array_equal(got, want) and carr_f.tobytes() == carr_h.tobytes()
del os.environ["GPSIQ_TEST_LIST_CAP"]
got = ctx.generate_batch(d2, ns2, fs2, SC08)                                # and the context is as good as before
assert np.array_equal(got, want) and gpsiq.device_eval_stats()[5] == s2[5]
print("fall-back ok")
ctx.close()
print("all ok")
'''


@pytest.mark.gpu
def test_a_wrong_map_is_caught_by_the_sampling_verify_mode(tmp_path):
    out = str(tmp_path / "libgpsiq_hooks.so")
    b = subprocess.run(["make", "-B", "-s", "-C", os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc"), "OUT=" + out, "EXTRA=-DGPSIQ_TEST_HOOKS"],
                       capture_output=True, text=True, timeout=900)
    assert b.returncode == 0 and os.path.getsize(out) > 100000, b.stderr[-3000:]
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=dict(os.environ, GPSIQ_LIB=out), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
def test_the_verify_mode_costs_little(tmp_path):
    """GPSIQ_CHAIN_VERIFY=64 at the headline workload: the sampled walks run on host threads under the synthesis kernel."""
    import time
    import numpy as np
    import torch
    import gpsiq
    from gpsiq.abi import NCO_REFERENCE
    from gpsiq.scenario import synth_blocks
    fs, ns, nb = 2.6e6, 260000, 2000
    d = synth_blocks(64, 16)[np.arange(nb) % 64]
    ring = torch.empty(nb * 2 * ns, dtype=torch.uint8, device="cuda")
    ctx = gpsiq.Context(0)
    ctx.set_nco_mode(NCO_REFERENCE)
    os.environ["GPSIQ_EVAL"] = "device"
    try:
        t = {}
        for every in ("0", "64", "0", "64"):
            os.environ["GPSIQ_CHAIN_VERIFY"] = every
            ctx.generate_batch(d, ns, fs, 1, device_ptr=ring.data_ptr())
            ts = []
            for _ in range(8):
                t0 = time.perf_counter()
                ctx.generate_batch(d, ns, fs, 1, device_ptr=ring.data_ptr())
                ts.append(time.perf_counter() - t0)
            t[every] = min(t.get(every, 1e9), sorted(ts)[len(ts) // 2])
        print(f"median call {t['0'] * 1e3:.3f} ms, with GPSIQ_CHAIN_VERIFY=64 {t['64'] * 1e3:.3f} ms")
        assert t["64"] < 1.05 * t["0"] + 40e-6, t
    finally:
        os.environ.pop("GPSIQ_CHAIN_VERIFY", None)
        os.environ.pop("GPSIQ_EVAL", None)
        ctx.close()
