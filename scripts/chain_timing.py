"""GPSIQ_NCO_REFERENCE with the carrier chain on the device against the chain on host threads, at bench.py's reference-NCO
workloads: the chain kernels alone for 4 / 8 / 16 stretches per block, level 2 on the host, and the whole
gpsiq_generate_batch call either way (best of 8).  Prints one line per measurement; run on the GPU box."""
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


def best(fn, n):
    t = float("inf")
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t = min(t, time.perf_counter() - t0)
    return t


def main():
    ctx = gpsiq.Context(0)
    ring = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
    pat = synth_blocks(64, 16)
    for label, fs, ss, nb in (("2M6_int8", 2.6e6, 1, 2000), ("10M_int16", 10e6, 2, 536), ("25M_int16", 25e6, 2, 200), ("2M6_int8_long", 2.6e6, 1, 4130)):
        ns = int(round(fs / 10))
        nb = min(nb, ring.numel() // (2 * ns * ss))
        d = pat[np.arange(nb) % 64]
        cin = gpsiq.chain_inputs(d)
        want = gpsiq.reference_chain(cin, fs, ns)
        t_serial = best(lambda: gpsiq.reference_chain(cin, fs, ns), 3)
        print(f"{label}: {nb} blocks x 16 ch; serial chain on host threads {t_serial * 1e3:.3f} ms", flush=True)
        for seg in (8, 16, 32):
            ms = min(gpsiq.chain_maps(cin, fs, ns, max_stretches=seg, ctx=ctx)[2] for _ in range(5))
            t_call = best(lambda: gpsiq.chain_maps(cin, fs, ns, max_stretches=seg, ctx=ctx), 5)
            maps = gpsiq.chain_maps(cin, fs, ns, max_stretches=seg, ctx=ctx)[0]
            s0 = gpsiq.chain_stats()
            got = gpsiq.chain_link(cin, maps, fs, ns)
            s1 = gpsiq.chain_stats()
            t_link = best(lambda: gpsiq.chain_link(cin, maps, fs, ns), 5)
            same = all(g.tobytes() == w.tobytes() for g, w in zip(got, want))
            print(f"  {seg:2d} stretches: kernels {ms:.3f} ms, maps call (copies + staging) {t_call * 1e3:.3f} ms, level 2 {t_link * 1e3:.3f} ms, "
                  f"linked {s1[0] - s0[0]}, walked {s1[1] - s0[1]}, equal to the serial chain: {same}", flush=True)
        ctx.set_nco_mode(NCO_REFERENCE)
        for where in ("host", "device"):
            os.environ["GPSIQ_CHAIN"] = where
            ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())
            t = best(lambda: ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr()), 8)
            print(f"  gpsiq_generate_batch, chain on the {where}: {t * 1e3:.3f} ms = {nb * ns / t / 1e9:.1f} G samples/s", flush=True)
            os.environ["GPSIQ_TRACE"] = "1"
            ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())
            del os.environ["GPSIQ_TRACE"]
        del os.environ["GPSIQ_CHAIN"]
        ctx.set_nco_mode(0)
    ctx.close()


if __name__ == "__main__":
    main()
