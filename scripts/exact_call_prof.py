"""One GPSIQ_NCO_REFERENCE gpsiq_generate_batch call after another into device memory, for rocprofv3 passes (kernel trace; PMC):
2.6 Msps int8 16 ch, 2 000 and 4 130 blocks (the headline's blocks per launch), descriptors in pageable memory (the device evaluation packs them on the pool).
No torch (a profiled interpreter with torch loaded has been seen to hang in teardown: run it under `timeout`); the output ring
comes from hipMalloc through ctypes.
   python scripts/exact_call_prof.py [calls per size]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import gpsiq  # noqa: E402
from gpsiq.abi import NCO_REFERENCE  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = gpsiq.Context(0)
hip = C.CDLL("libamdhip64.so")            # the runtime the binding has loaded
ring = C.c_void_p()
assert hip.hipMalloc(C.byref(ring), C.c_size_t(4130 * 520000)) == 0      # bench.py's ring: 4 130 blocks of 2.6 Msps int8
pat = synth_blocks(64, 16)
ctx.set_nco_mode(NCO_REFERENCE)
for fs, ss, nb in ((2.6e6, 1, 2000), (2.6e6, 1, 4130)) + (((25e6, 2, 200),) if os.environ.get("EXACT_PROF_25M") else ()):
    ns = int(round(fs / 10))
    d = pat[np.arange(nb) % 64]
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.value)
        ts.append(time.perf_counter() - t0)
    print(f"{fs / 1e6:g} Msps int{8 * ss}, {nb} blocks: calls of {', '.join(f'{t * 1e3:.3f}' for t in ts)} ms", flush=True)
print("device evaluation statistics:", gpsiq.device_eval_stats(), flush=True)
sys.stdout.flush()
ctx.close()
