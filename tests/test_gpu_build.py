"""The library compiled ON the GPU box: `make -B` of csrc/ with hipcc for gfx950 into a scratch directory, loaded instead of the
shipped libgpsiq.so (GPSIQ_LIB) in a fresh interpreter, which checks that it carries the shipped library's kernel id (= the same
device-code source) and runs the headline parity case (2.6 Msps int8 16 ch, every element against the oracle) and one
GPSIQ_NCO_REFERENCE batch with the carrier chain on the device through it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpsiq, _oracle
from gpsiq.abi import NCO_REFERENCE, SC08
from gpsiq.scenario import synth_blocks
assert os.path.samefile(gpsiq.LIB_PATH, os.environ["GPSIQ_LIB"])
print("kernels_id", gpsiq.kernels_id())
fs, ns, nchan = 2.6e6, 260000, 16
d = synth_blocks(2, nchan, seed=2616)
q, _ = gpsiq.quantize_blocks(d, fs, ns)
orc = _oracle.load_oracle()
ctx = gpsiq.Context(0)
out = ctx.generate_batch(d, ns, fs, SC08)
for b in range(2):
    assert np.array_equal(out[b], orc.block_fixed(q[b], ns, SC08, seq=True)), b
ctx.set_nco_mode(NCO_REFERENCE)
d = synth_blocks(64, nchan, seed=7)
os.environ["GPSIQ_CHAIN"] = "device"
dev = ctx.generate_batch(d, ns, fs, SC08)
os.environ["GPSIQ_CHAIN"] = "host"
host = ctx.generate_batch(d, ns, fs, SC08)
assert np.array_equal(dev, host)
ctx.close()
print("parity ok")
'''


@pytest.mark.gpu
def test_the_library_builds_on_the_gpu_box_and_the_fresh_build_passes_parity(tmp_path):
    import gpsiq
    out = str(tmp_path / "libgpsiq.so")
    b = subprocess.run(["make", "-B", "-s", "-C", os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc"), "OUT=" + out],
                       capture_output=True, text=True, timeout=900)
    assert b.returncode == 0 and os.path.getsize(out) > 100000, b.stderr[-3000:]
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=dict(os.environ, GPSIQ_LIB=out), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "parity ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert f"kernels_id {gpsiq.kernels_id()}" in r.stdout          # the same device-code source as the shipped library
