"""A/B of the knobs of the device carrier chain inside gpsiq_generate_batch (GPSIQ_NCO_REFERENCE): GPSIQ_CHAIN_HEAD,
GPSIQ_CHAIN_STRETCHES, GPSIQ_REF_CHUNK_BLOCKS, GPSIQ_CHAIN, GPSIQ_PIECE_STREAMS (also for the fixed-point batch), and one GPSIQ_TRACE=2 timeline of the default.  Run on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import gpsiq  # noqa: E402
from gpsiq.abi import NCO_REFERENCE  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402
import torch  # noqa: E402


def best(fn, n=8):
    t = float("inf")
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t = min(t, time.perf_counter() - t0)
    return t


def main():
    ring = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
    pat = synth_blocks(64, 16)
    ctx = gpsiq.Context(0)
    ctx.set_nco_mode(NCO_REFERENCE)
    legs = (("2M6_int8", 2.6e6, 1, 2000), ("10M_int16", 10e6, 2, 536), ("25M_int16", 25e6, 2, 200))
    for label, fs, ss, nb in (() if "fixed" in sys.argv[1:] else legs):
        ns = int(round(fs / 10))
        d = pat[np.arange(nb) % 64]
        call = lambda: ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())  # noqa: E731
        call()
        os.environ["GPSIQ_TRACE"] = "2"
        call()
        del os.environ["GPSIQ_TRACE"]
        print(f"{label}: default {best(call) * 1e3:.3f} ms", flush=True)
        knobs = (("GPSIQ_PIECE_STREAMS", ("1", "2")), ("GPSIQ_DESC_SETS", ("2", "3", "4", "2", "4"))) if "streams" in sys.argv[1:] else None
        for knob, values in knobs or (("GPSIQ_CHAIN_HEAD", ("0", "128", "256", "400", "600", "900")), ("GPSIQ_CHAIN_STRETCHES", ("16", "32")),
                             ("GPSIQ_REF_CHUNK_BLOCKS", ("128", "256", "512")), ("GPSIQ_CHAIN", ("host", "device")),
                             ("GPSIQ_PIECE_STREAMS", ("1", "2", "1", "2"))):
            for v in values:
                os.environ[knob] = v
                call()
                print(f"  {knob}={v}: {best(call) * 1e3:.3f} ms", flush=True)
            del os.environ[knob]
        print(f"{label}: default again {best(call) * 1e3:.3f} ms", flush=True)
    ctx.set_nco_mode(0)
    nb = min(4130, ring.numel() // (2 * 260000))             # never a byte beyond the ring
    d = pat[np.arange(nb) % 64]
    call = lambda: ctx.generate_batch(d, 260000, 2.6e6, 1, device_ptr=ring.data_ptr())  # noqa: E731
    call()
    for knob, values in (("GPSIQ_PIECE_STREAMS", ("1", "2", "1", "2")), ("GPSIQ_DESC_SETS", ("2", "4", "2", "4"))):
        for v in values:
            os.environ[knob] = v
            call()
            t = best(call)
            print(f"fixed-point batch, {nb} blocks at 2.6 Msps int8, {knob}={v}: {t * 1e3:.3f} ms = {nb * 260000 / t / 1e9:.1f} G samples/s", flush=True)
        del os.environ[knob]
    ctx.close()


if __name__ == "__main__":
    main()
