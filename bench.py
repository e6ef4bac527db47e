#!/usr/bin/env python3
"""bench.py — throughput of the libgpsiq hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one gpsiq_launch through the C-ABI) over one
batch of --blocks 0.1 s blocks of synthetic channel descriptors, with the
quantised descriptors already resident in HBM; the IQ output goes to a device ring of
blocks*block_bytes (>= 2 GiB by default, far beyond the 256 MiB Infinity Cache) so
that the writes really reach HBM.  Workload = BASELINE.json metric: 2.6 Msps, int8,
16 channels.  With N > 1 (launched by torch.distributed.run, one rank per GPU) the time
axis is sharded: rank r owns blocks [r*B, (r+1)*B) of one N*B-block timeline, its
carrier phases seeded by the exact closed-form prefix; no collective touches the data
path (weak scaling: per-GPU work is fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — algorithmic IQ bytes per launch / mean launch duration (HIP events on
                 the launch stream) against the 8 TB/s HBM peak,
  cpu_baseline — the reference's own loop (oracle/_ref, kind "reference") or our port
                 of it (oracle/, kind "port") timed on this host, 1 core, bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
_LIB = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "gpsiq", "libgpsiq.so")
if not os.path.exists(_LIB):
    # fresh checkout: build the HIP extension first (there is no other path to run); local rank 0
    # builds, the other ranks of the node wait for the file
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc")], check=True)
    else:
        for _ in range(600):
            if os.path.exists(_LIB):
                break
            time.sleep(0.5)
        time.sleep(1.0)

PREHEAT_LAUNCHES = 12   # untimed, before the warm-up steps: clock ramp (see main)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fs", type=float, default=2.6e6)
    ap.add_argument("--nchan", type=int, default=16)
    ap.add_argument("--sample-size", type=int, default=1, choices=(1, 2), help="1 = int8 IQ, 2 = int16 IQ")
    ap.add_argument("--blocks", type=int, default=0, help="0.1 s blocks per step per GPU (0: enough for a 2 GiB ring)")
    ap.add_argument("--variant", type=str, default="auto")
    ap.add_argument("--seed", type=int, default=20250215)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=299, help="blocks of the CPU baseline sample (299 = 30 s, config 1)")
    ap.add_argument("--sweep", action="store_true", help="also time every kernel variant (extra stderr lines)")
    return ap.parse_args()


def effective_cpus():
    """CPUs this process may actually use: hardware threads, affinity mask and cgroup-v2 quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(desc, fs, nsamp, sample_size, nblocks, passes=3):
    """Time the reference CPU path on this host, one core (the reference has exactly one
    generation thread, gps-sim.c:314).  Checker code only: nothing here is on the GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    ref = _oracle.load_ref()
    d = np.ascontiguousarray(desc[:nblocks])
    t0 = time.perf_counter()
    for _ in range(passes):
        if ref is not None:
            kind = "reference"
            ref.run_blocks(d, int(fs), sample_size, 1)
        else:
            kind = "port"
            orc = _oracle.load_oracle()
            for b in range(nblocks):
                orc.block_float(d[b], nsamp, fs, sample_size)
    dt = time.perf_counter() - t0
    how = ("oracle/_ref: reference gps.c:2767-2865 compiled in place with the reference's own flags (-std=c11 -Og)"
           if kind == "reference" else "oracle_block_float, gcc -O2")
    out = {"value": round(passes * nblocks * nsamp / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": kind,
           "sample": f"{passes} passes over the first {nblocks} blocks ({nblocks * 0.1:.1f} s of signal) of the same workload, "
                     f"{how}, {dt:.1f} s wall"}
    # informational: the same loop on every core this process may use (the reference itself is
    # single-threaded; blocks are independent given their descriptors, so this is the best a CPU port could do)
    try:
        import multiprocessing as mp
        ncpu = effective_cpus()
        per = max(4, min(nblocks, int(6.0 / (dt / (passes * nblocks)))))    # ~6 s of work per process
        with mp.get_context("fork").Pool(ncpu) as pool:
            pool.map(_cpu_worker, [(d[:2].copy(), fs, nsamp, sample_size)] * ncpu)      # start the workers, load the library
            t1 = time.perf_counter()
            pool.map(_cpu_worker, [(d[:per].copy(), fs, nsamp, sample_size)] * ncpu, chunksize=1)
            dt_all = time.perf_counter() - t1
        out["all_cores"] = {"value": round(ncpu * per * nsamp / dt_all / 1e6, 1), "unit": "Msamples/s", "cores": ncpu,
                            "sample": f"{ncpu} processes (cgroup/affinity limit; {os.cpu_count()} hardware threads) x {per} blocks, "
                                      f"{dt_all:.1f} s wall"}
    except Exception as e:                                           # never fail the bench for the extra figure
        out["all_cores"] = {"error": str(e)[:100]}
    return out


def _cpu_worker(a):
    d, fs, nsamp, sample_size = a
    import _oracle
    ref = _oracle.load_ref()
    if ref is not None:
        ref.run_blocks(d, int(fs), sample_size, 1)
    else:
        orc = _oracle.load_oracle()
        for b in range(len(d)):
            orc.block_float(d[b], nsamp, fs, sample_size)
    return len(d)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # CPU baseline first (rank 0, N = 1 only): it forks worker processes for the all-cores
    # figure, which must happen before this process holds a GPU context
    cpu_base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        from gpsiq.scenario import synth_blocks as _sb
        nsamp_c = int(round(args.fs / 10))
        pat = _sb(64, args.nchan, seed=args.seed)
        d_cpu = np.concatenate([pat] * (-(-args.cpu_blocks // 64)))[: args.cpu_blocks]
        cpu_base = cpu_baseline(d_cpu, args.fs, nsamp_c, args.sample_size, args.cpu_blocks)

    import torch
    import gpsiq
    from gpsiq.scenario import synth_blocks
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (libgpsiq has no CPU path)"
    # one rank per GPU; the modulo only matters when the multi-rank path is exercised on a box
    # with fewer GPUs than ranks (GPSIQ_BENCH_BACKEND=gloo, see scripts/gpu_validate.sh)
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist = None
    backend = os.environ.get("GPSIQ_BENCH_BACKEND", "nccl")    # nccl == RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    fs, nchan, ss = args.fs, args.nchan, args.sample_size
    nsamp = int(round(fs / 10))                      # NUM_IQ_SAMPLES, reference sdr.h:26
    blk_bytes = 2 * nsamp * ss
    stride = (blk_bytes + 15) & ~15
    nblocks = args.blocks or -(-(2 << 30) // stride)  # ring >= 2 GiB
    variant = gpsiq.variants()[args.variant]

    # one global timeline of world*nblocks blocks; this rank's shard is [rank*nblocks, ...)
    # Descriptors: distinct code/Doppler state per block for a short pattern, tiled over
    # the timeline (host generation cost only), then quantised with the exact carrier prefix.
    from gpsiq.shard import max_over_ranks, shard_descriptors
    pattern = synth_blocks(min(64, nblocks), nchan, seed=args.seed)
    reps = -(-nblocks * world // len(pattern))
    desc_all = np.concatenate([pattern] * reps)[: nblocks * world]
    q, (b0, b1) = shard_descriptors(desc_all, fs, nsamp, rank, world)
    assert b1 - b0 == nblocks

    ctx = gpsiq.Context(dev)
    ctx.set_descriptors(q)
    ring = torch.empty(nblocks * stride, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.launch(0, nblocks, nsamp, ss, ring.data_ptr(), stride, stream=stream, variant=variant)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # The GPU needs ~25 ms of load to reach its sustained clock (per-dispatch durations in
    # profiles/r01_final_kernel_trace_stats.txt fall from 4.35 ms to 3.54 ms over the first seven
    # launches), so the device is brought to that state before the W warm-up steps; untimed.
    for _ in range(PREHEAT_LAUNCHES):
        step()
    for _ in range(args.warmup):
        step()
    sync_all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t_max = max_over_ranks(t_local, dist, device="cuda" if backend == "nccl" else "cpu")
    launch_ms = e0.elapsed_time(e1) / args.steps       # HIP events on the launch stream

    if args.sweep and rank == 0:
        for name, v in gpsiq.variants().items():
            if name == "auto":
                continue
            try:
                ms = min(ctx.time_launches(0, nblocks, nsamp, ss, ring.data_ptr(), stride, 5, stream=stream, variant=v)
                         for _ in range(3))
                print(f"[sweep] variant {name}: {ms:.3f} ms/launch, {nblocks * nsamp / ms / 1e6:.1f} Gsamples/s",
                      file=sys.stderr)
            except gpsiq.GpsiqError as e:
                print(f"[sweep] variant {name}: {e}", file=sys.stderr)

    if args.sweep and rank == 0:
        # the drop-in path with HOST destination buffers (gpsiq_generate_batch: quantise,
        # H2D descriptors, kernel, D2H samples): PCIe-bound, reported separately, never `value`
        nb_h = min(256, nblocks)
        pinned = torch.empty(nb_h * blk_bytes, dtype=torch.uint8).pin_memory()
        ctx.generate_batch(desc_all[:nb_h], nsamp, fs, ss, host_ptr=pinned.data_ptr())
        t1 = time.perf_counter()
        ctx.generate_batch(desc_all[:nb_h], nsamp, fs, ss, host_ptr=pinned.data_ptr())
        dt = time.perf_counter() - t1
        print(f"[host-dst] gpsiq_generate_batch -> pinned host memory: {nb_h} blocks in {dt * 1e3:.1f} ms = "
              f"{nb_h * nsamp / dt / 1e6:.0f} Msamples/s, {nb_h * blk_bytes / dt / 1e9:.1f} GB/s over PCIe", file=sys.stderr)
        ctx.set_descriptors(q)

    if args.sweep and rank == 0:
        # the real-time drop-in call: ONE block per call into a page-locked host buffer, as the
        # patched gps thread would issue it every 0.1 s (quantise, H2D, kernel, D2H, sync)
        lat = []
        for k in range(60):
            ch1 = desc_all[k % len(desc_all)]
            t1 = time.perf_counter()
            ctx.generate_block(ch1, nsamp, fs, ss, host_ptr=pinned.data_ptr())
            lat.append(time.perf_counter() - t1)
        lat = sorted(lat[10:])
        print(f"[block] gpsiq_generate_block -> pinned host memory: median {lat[len(lat) // 2] * 1e6:.0f} us, "
              f"max {lat[-1] * 1e6:.0f} us per 0.1 s block ({0.1 / lat[len(lat) // 2]:.0f}x real time, one call at a time)",
              file=sys.stderr)
        ctx.set_descriptors(q)

    if args.sweep and rank == 0:
        # the whole drop-in batch call with a DEVICE destination: host quantiser + descriptor
        # upload + kernel (no D2H) -- shows what the host side of gpsiq_generate_batch costs
        nb_d = min(nblocks, len(desc_all))
        ctx.generate_batch(desc_all[:nb_d], nsamp, fs, ss, device_ptr=ring.data_ptr())
        dt = float("inf")
        for _ in range(5):
            t1 = time.perf_counter()
            ctx.generate_batch(desc_all[:nb_d], nsamp, fs, ss, device_ptr=ring.data_ptr())
            dt = min(dt, time.perf_counter() - t1)
        print(f"[device-dst] gpsiq_generate_batch -> device memory: {nb_d} blocks in {dt * 1e3:.1f} ms = "
              f"{nb_d / dt / 1e3:.0f} kblocks/s = {nb_d * 0.1 / dt:.0f}x real time (kernel alone {launch_ms:.1f} ms)", file=sys.stderr)
        ctx.set_descriptors(q)

    if args.sweep and rank == 0:
        # the whole run-ahead chain of INTEGRATION.md section 3 on a BASELINE config-4 shaped scenario:
        # RINEX file -> subframes -> nav words -> per-block refresh with the 30 s nav refreshes
        # (host, gpsiq/pipeline.py) -> IQ in device memory (one gpsiq_generate_batch)
        import tempfile
        from gpsiq.pipeline import RunAhead
        from gpsiq.scenario import circle_track, llh_to_ecef, synth_rinex_records, write_rinex_nav
        pos = llh_to_ecef(35.681298, 139.766247, 10.0)
        utc = dict(alpha=[0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06], beta=[0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07],
                   A0=-0.931322574615e-09, A1=-0.355271367880e-14, tot=233472, wnt=2190, dtls=18)
        with tempfile.TemporaryDirectory() as td:
            path = write_rinex_nav(os.path.join(td, "cfg4.21n"), synth_rinex_records(12, pos, 2190, 270000.0, seed=35, sets=2), utc, 2)
            nb4 = min(5999, (ring.numel() // (4 * nsamp)))          # 600 s when the ring holds it (int16 output)
            xyz = circle_track(pos, nb4)
            t1 = time.perf_counter()
            eph4, utc4, nset = gpsiq.rinex_read(path, 2)
            ieph = gpsiq.rinex_select(eph4, nset, 2190, 270000.0)
            svs = [sv for sv in range(32) if eph4[ieph, sv]["vflg"]]
            ra = RunAhead(eph4[ieph], utc4, svs, 2190, 270000.0, xyz[0])
            d4 = ra.descriptors(xyz[1:])
            t2 = time.perf_counter()
            ctx.generate_batch(d4, nsamp, fs, 2, device_ptr=ring.data_ptr())
            t3 = time.perf_counter()
        print(f"[pipeline] RINEX -> {nb4} blocks x {len(svs)} ch, circle track, int16 @ {fs / 1e6:g} Msps: host chain "
              f"{(t2 - t1) * 1e3:.1f} ms + gpsiq_generate_batch {(t3 - t2) * 1e3:.1f} ms = {nb4 * 0.1 / (t3 - t1):.0f}x real time "
              f"end to end ({nb4 * 0.1:.0f} s of signal)", file=sys.stderr)
        ctx.set_descriptors(q)

    if args.sweep and rank == 0:
        # the host refresh that feeds the kernel (gpsiq_refresh_batch, reference gps.c:2731-2765):
        # blocks per second on this host, 1 thread and all threads
        from gpsiq.scenario import circle_track, llh_to_ecef, synth_constellation, synth_iono, synth_tracks
        pos = llh_to_ecef(35.681298, 139.766247, 10.0)
        eph = synth_constellation(nchan, pos, 270000.0, seed=3)
        xyz = circle_track(pos, 200000)
        for nt in (1, 0):
            trk = synth_tracks(nchan, 2190, 270000.0)
            gpsiq.track_init(eph, synth_iono(), 2190, 270000.0, xyz[0], trk)
            t1 = time.perf_counter()
            gpsiq.refresh_batch(eph, synth_iono(), 2190, 270000.0, xyz[1:], trk, nthreads=nt)
            dt = time.perf_counter() - t1
            print(f"[refresh] gpsiq_refresh_batch {nchan} ch, {len(xyz) - 1} blocks, threads={'all' if nt == 0 else nt}: "
                  f"{(len(xyz) - 1) / dt / 1e3:.1f} kblocks/s = {(len(xyz) - 1) * 0.1 / dt:.0f}x real time", file=sys.stderr)

    if rank == 0:
        samples_step = nblocks * nsamp * world
        value = samples_step * args.steps / t_max / 1e6
        alg_bytes = nblocks * blk_bytes                 # algorithmic: 2 or 4 B per complex sample, reads ~0
        # which tile kernel ran: plain adds for int8 always, for int16 when no block's sum of (int)(250*|gain|) exceeds 32767
        forced_packed = os.environ.get("GPSIQ_NO_FAST", "0") not in ("", "0")
        plain_add = (ss == 1 or int(np.floor(250.0 * np.abs(q["gain"])).sum(axis=1).max()) <= 32767) and not forced_packed
        core_cycles = 23.85 if plain_add else 26.7
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(f"{int(fs)}_{nchan}_{ss}_{nblocks}")
            except Exception:
                traffic = None
        out = {
            "metric": "IQ Msamples/s @16ch int8" if (nchan == 16 and ss == 1) else f"IQ Msamples/s @{nchan}ch int{8 * ss}",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{fs / 1e6:g} Msps int{8 * ss} IQ, {nchan} channels, {nblocks} x 0.1 s blocks per GPU per step "
                                   f"({nblocks * 0.1:.1f} s of signal, {nblocks * stride / 2**30:.2f} GiB ring), time-sharded x{world}",
                       "fs_hz": fs, "channels": nchan, "sample_bytes": ss, "blocks_per_gpu": nblocks,
                       "samples_per_block": nsamp, "variant": args.variant, "preheat_launches": PREHEAT_LAUNCHES,
                       "x_realtime": round(value * 1e6 / fs, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel_ms": round(launch_ms, 4), "algorithmic_bytes_per_launch": alg_bytes},
            # informational: the limiter actually hit.  Per (channel, 64-sample row)
            # and SIMD the row-kernel core costs, with the single-instruction rates measured on this
            # chip (profiles/r01_ubench_valu_encodings.txt: 4.3 nominal cycles for SGPR-operand / VOP3 /
            # SDWA / packed forms, 4.4 for v_lshl_add_u64, 2.5 for plain VOP2),
            #   plain-add kernels (all sums inside int16): 3 x 4.3 + 4.3 / 2 + 2 x 4.4 = 23.85 cycles
            #   packed kernels (larger gains):             3 x 4.3 + 2 x 2.5 + 2 x 4.4 = 26.7 cycles
            # peak = every SIMD of 256 CUs issuing only that core at the 2.4 GHz maximum clock.
            "issue_roofline": {"bound": "valu-issue", "unit": "Gchannel-samples/s", "core_cycles": core_cycles,
                               "achieved": round(nblocks * nsamp * nchan / (launch_ms * 1e-3) / 1e9, 1),
                               "peak": round(256 * 4 * 64 * 2.4e9 / core_cycles / 1e9, 1),
                               "frac": round(nblocks * nsamp * nchan / (launch_ms * 1e-3) / (256 * 4 * 64 * 2.4e9 / core_cycles), 4)},
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
