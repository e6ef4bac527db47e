"""The batch calls with the evaluation on the device (csrc/gpsiq_evaldev.cpp) against the host path of rounds 4-5, at bench.py's
workloads: whole gpsiq_generate_batch call into device memory, both NCO models, descriptors in pageable / page-locked / device
memory; the host time the call spends on descriptors (pack, repair, the host walker's share); the kernel + patches alone for
the ratio.  Prints one line per measurement and the library's own trace of one call each; run on the GPU box.
   python scripts/eval_timing.py [threads]      (threads: GPSIQ_THREADS for the whole process, default: what the box grants)"""
import os
import sys
import time

if len(sys.argv) > 1:
    os.environ["GPSIQ_THREADS"] = sys.argv[1]

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import gpsiq  # noqa: E402
from gpsiq.abi import NCO_FIXED, NCO_REFERENCE  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402
import torch  # noqa: E402


def timed(fn, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def main():
    ctx = gpsiq.Context(0)
    ring = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
    pat = synth_blocks(64, 16)
    print(f"GPSIQ_THREADS={os.environ.get('GPSIQ_THREADS', '(all)')}", flush=True)
    for label, fs, ss, nb in (("2M6_int8", 2.6e6, 1, 2000), ("2M6_int8_4130", 2.6e6, 1, 4130), ("10M_int16", 10e6, 2, 536), ("25M_int16", 25e6, 2, 200)):
        ns = int(round(fs / 10))
        nb = min(nb, ring.numel() // (2 * ns * ss))
        d = pat[np.arange(nb) % 64]
        raw = torch.from_numpy(d.view(np.uint8).reshape(-1).copy())
        pinned, resident = raw.pin_memory(), raw.cuda()
        srcs = {"pageable": d, "page-locked": (pinned.data_ptr(), nb, 16), "device": (resident.data_ptr(), nb, 16)}
        # kernel + patches alone (resident descriptors): what the call is held against
        ctx.set_nco_mode(NCO_FIXED)
        q_r, patches, _ = gpsiq.reference_blocks(d, fs, ns)
        ctx.set_descriptors(q_r)
        ctx.set_patches(patches)
        blk = 2 * ns * ss
        ctx.time_launches(0, nb, ns, ss, ring.data_ptr(), blk, 3)
        km = min(ctx.time_launches(0, nb, ns, ss, ring.data_ptr(), blk, 5) for _ in range(2))
        print(f"{label}: {nb} blocks x 16 ch, {len(patches)} patches; synthesis + patches alone {km:.3f} ms = {nb * ns / km / 1e6:.1f} G samples/s", flush=True)
        for mode, mname in ((NCO_REFERENCE, "reference"), (NCO_FIXED, "fixed")):
            ctx.set_nco_mode(mode)
            os.environ["GPSIQ_EVAL"] = "host"
            ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())
            best, med = timed(lambda: ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr()), 10)
            print(f"  {mname:9s} host path   (pageable)   : best {best * 1e3:.3f} ms, median {med * 1e3:.3f} ms = {nb * ns / med / 1e9:.1f} G samples/s, call / kernel {med * 1e3 / km:.2f}", flush=True)
            os.environ["GPSIQ_EVAL"] = "device"
            for kind, src in srcs.items():
                ctx.generate_batch(src, ns, fs, ss, device_ptr=ring.data_ptr())
                best, med = timed(lambda: ctx.generate_batch(src, ns, fs, ss, device_ptr=ring.data_ptr()), 10)
                host_ms = ctx.device_eval_host_ms()
                print(f"  {mname:9s} device path ({kind:11s}): best {best * 1e3:.3f} ms, median {med * 1e3:.3f} ms = {nb * ns / med / 1e9:.1f} G samples/s, call / kernel {med * 1e3 / km:.2f}, "
                      f"host stages {host_ms:.3f} ms -> GPUs this host feeds {km / host_ms if host_ms > 0 else float('inf'):.1f}", flush=True)
            os.environ["GPSIQ_TRACE"] = "1"
            ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())
            ctx.generate_batch(srcs["device"], ns, fs, ss, device_ptr=ring.data_ptr())
            del os.environ["GPSIQ_TRACE"]
        del os.environ["GPSIQ_EVAL"]
        ctx.set_nco_mode(NCO_FIXED)
    print("device evaluation statistics (calls, pairs, to the host walker, slots repaired, patches, fell back):", gpsiq.device_eval_stats(), flush=True)


if __name__ == "__main__":
    main()
