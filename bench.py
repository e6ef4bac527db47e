#!/usr/bin/env python3
"""bench.py — throughput of the libgpsiq hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Launch: with N > 1 and no WORLD_SIZE in the environment this script starts its own N ranks
(torch.distributed.run, one per GPU, rendezvous on 127.0.0.1); under torchrun it checks
WORLD_SIZE == N.  Fewer visible GPUs than ranks, or a WORLD_SIZE that disagrees with --gpus, is an
error (non-zero exit), never a silent 1-GPU run.

A "step" is one pass of the hot path over one batch of synthetic input: --launches
gpsiq_launch calls (C-ABI) over --blocks 0.1 s blocks each, i.e. launches*blocks DISTINCT blocks of
one timeline whose quantised descriptors are resident in HBM; the IQ output goes to a device ring of
blocks*block_bytes (>= 2 GiB by default, far beyond the 256 MiB Infinity Cache) that every launch
of the step rewrites, so the writes really reach HBM.  Workload = BASELINE.json metric: 2.6 Msps,
int8, 16 channels.  With N > 1 the time axis is sharded: rank r owns blocks [r*B, (r+1)*B) of one
N*B-block timeline; it builds and quantises ONLY those blocks and learns its carrier seed from 32
bytes per channel that every rank publishes about its own range (gpsiq_shard_carry/_seed: one
all-gather at set-up).  No collective touches the data path (weak scaling: per-GPU work is fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — algorithmic IQ bytes per launch / mean launch duration (HIP events on the launch
                 stream) against the 8 TB/s HBM peak,
  counters     — instruction-mix-independent fractions of the dominant kernel from committed rocprofv3 PMC passes of this
                 command (profiles/pmc_counters.json): VALU issue fraction, LDS busy, LDS bank-conflict ratio
                 (+ own_stream_efficiency: the kernel against the cost of its own instruction stream; not a roofline),
  exact_mode_value / _roofline / _bound — the model whose output IS the reference's (GPSIQ_NCO_REFERENCE), whole
                 gpsiq_generate_batch call at the headline workload, beside `value`,
  reference_nco — that mode in detail: the call at the headline workload and at 25 Msps, the same with the carrier chain on host
                 threads, the chain's parts (level 1 kernels on the device, level 2 link on the host, blocks linked / walked,
                 equality with the serial chain), evaluation alone, kernel alone, which side bounds it; at N > 1 the mode
                 time-sharded over the ranks (chain by time),
  placement    — (N > 1) where every rank runs: host, device ordinal, PCI bus id, CPUs granted, GPSIQ_THREADS,
  cpu_baseline — the reference's own loop (oracle/_ref, kind "reference") or our port of it
                 (oracle/, kind "port") timed on this host, 1 core, bounded sample (N = 1 only),
  end_to_end   — the same per-GPU block count from scratch on every rank: host refresh of its own
                 blocks (gpsiq_refresh_epochs) -> quantise -> carrier seed exchange -> upload -> kernel,
                 once as one serial batch and once as a stream of rounds with the host side of the next
                 round overlapping the kernel of this one (end_to_end.streamed, with every rank's host and kernel
                 time per round and which of the two it is bound by),
  extra        — short legs for the other BASELINE configs (each with its own roofline), the
                 host-destination and single-block drop-in calls, GPSIQ_NCO_REFERENCE, first-launch times.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
_LIB = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "gpsiq", "libgpsiq.so")

PREHEAT_LAUNCHES = 12   # untimed, before the warm-up steps: clock ramp (see main)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TOKYO_LLH = (35.681298, 139.766247, 10.0)   # BASELINE configs 1/2: static position
WEEK, SEC0 = 2190, 270000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fs", type=float, default=2.6e6)
    ap.add_argument("--nchan", type=int, default=16)
    ap.add_argument("--sample-size", type=int, default=1, choices=(1, 2), help="1 = int8 IQ, 2 = int16 IQ")
    ap.add_argument("--blocks", type=int, default=0, help="0.1 s blocks per launch per GPU (0: enough for a 2 GiB ring)")
    ap.add_argument("--launches", type=int, default=18, help="launches (of --blocks distinct blocks each) per step")
    ap.add_argument("--variant", type=str, default="auto")
    ap.add_argument("--seed", type=int, default=20250215)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs (other configs, drop-in calls)")
    ap.add_argument("--cpu-blocks", type=int, default=299, help="blocks of the CPU baseline sample (299 = 30 s, config 1)")
    ap.add_argument("--rounds", type=int, default=16, help="rounds of the streamed end-to-end leg")
    ap.add_argument("--sweep", action="store_true", help="also time every kernel variant (extra stderr lines)")
    ap.add_argument("--exact-leg-only", action="store_true", help="only the GPSIQ_NCO_REFERENCE batch call at the headline size, one JSON line "
                                                                   "(the default run starts this with GPSIQ_THREADS=2 for reference_nco.two_host_threads)")
    ap.add_argument("--dry-run", action="store_true",
                    help="host side only (launch logic, rendezvous, sharded refresh/quantise, seed exchange); no device, value null")
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


def self_launch(args):
    """python bench.py --gpus N without a launcher: become the launcher."""
    if not args.dry_run and os.environ.get("GPSIQ_BENCH_SHARE_GPU", "0") in ("", "0"):
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
            return 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def ensure_built():
    if os.path.exists(_LIB):
        return
    # fresh checkout: build the HIP extension first (there is no other path to run); local rank 0
    # builds, the other ranks of the node wait for the file
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc")], check=True)
    else:
        for _ in range(600):
            if os.path.exists(_LIB):
                break
            time.sleep(0.5)
        time.sleep(1.0)


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
    # informational: the same lines compiled at -O3 (the reference ships -Og), one core, one pass
    try:
        o3_path = os.path.join(ROOT, "oracle", "_ref", "libgpsref_O3.so")
        if kind == "reference" and os.path.exists(o3_path):
            ref3 = _oracle.Ref(o3_path)
            ref3.run_blocks(d[:4], int(fs), sample_size, 1)
            t3 = time.perf_counter()
            ref3.run_blocks(d, int(fs), sample_size, 1)
            dt3 = time.perf_counter() - t3
            out["at_O3"] = {"value": round(nblocks * nsamp / dt3 / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                            "sample": f"1 pass over the same {nblocks} blocks, the same lines at -std=c11 -O3 (no -march=native: the object is built on another host), {dt3:.1f} s wall"}
    except Exception as e:
        out["at_O3"] = {"error": str(e)[:100]}
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


def roofline_obj(alg_bytes, launch_ms, traffic=None):
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "kernel_ms": round(launch_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes)}


def counters_obj(key, launch_ms):
    """Counter-based fractions of the dominant kernel: rocprofv3 --pmc passes of this very command, committed under
    profiles/ (scripts/gpu_prof.sh -> scripts/prof_summary.py -> profiles/pmc_counters.json), per launch, against the
    launch duration measured live in this run.  Independent of the kernel's own instruction mix (unlike issue_roofline).
    The counts are replayed, not measured here: they are only emitted when the profile was taken from a library built
    from the same device code as the one loaded now (gpsiq_kernels_id), else {"stale_profile": true} and nothing else."""
    import gpsiq
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        c = json.load(open(path)).get(key)
    except Exception:
        c = None
    if not c:
        return None
    if c.get("kernels_id") != gpsiq.kernels_id():
        return {"stale_profile": True, "profile_kernels_id": c.get("kernels_id"), "loaded_kernels_id": gpsiq.kernels_id(),
                "source": c.get("source")}
    valu_peak = 256 * 4 * 2.4e9 / 2.0                     # wave64 VALU instructions/s: 1024 SIMD-32s, 2 cycles each, 2.4 GHz
    cu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0                 # cycles the launch was resident, per XCD = per CU
    out = {"source": c.get("source"), "kernels_id": c.get("kernels_id"), "stale_profile": False,
           "collected_with": "rocprofv3 --pmc (SQ passes alone, no tracing), averages per launch of the dominant kernel",
           "valu_wave_insts_per_launch": int(c["SQ_INSTS_VALU"]),
           "valu_issue_frac": round(c["SQ_INSTS_VALU"] / (launch_ms * 1e-3) / valu_peak, 4),
           "valu_issue_peak": "256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 /s",
           "lds_busy": round(c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cu_cycles), 4),
           "lds_conflict_ratio": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4),
           "lds_wave_insts_per_launch": int(c["SQ_INSTS_LDS"]),
           "clock_ghz_under_load": round(cu_cycles / c["_avg_duration_ns_pmc_sq1"], 3),
           "profiled_launch_ms": round(c["_avg_duration_ns_pmc_sq1"] * 1e-6, 4)}
    return out


def replayed_traffic(key):
    """HBM bytes per launch from the committed WRITE_SIZE / FETCH_SIZE passes (profiles/pmc_traffic.json), or None when
    there is no profile of this workload or it was taken from another kernel than the loaded library's."""
    import gpsiq
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None, False
    if key not in d:
        return None, False
    if (d.get(key + "_detail") or {}).get("kernels_id") != gpsiq.kernels_id():
        return None, True
    return d[key], False


class Scenario:
    """BASELINE config 1/2 geometry for the end-to-end leg: static receiver, synthetic RINEX v2 file with
    nchan satellites in view -> ephemeris -> the host chain of gpsiq/pipeline.py."""

    def __init__(self, nchan, tmpdir):
        import gpsiq
        from gpsiq.scenario import llh_to_ecef, synth_rinex_records, write_rinex_nav
        self.pos = llh_to_ecef(*TOKYO_LLH)
        utc = dict(alpha=[0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06], beta=[0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07],
                   A0=-0.931322574615e-09, A1=-0.355271367880e-14, tot=233472, wnt=WEEK, dtls=18)
        path = write_rinex_nav(os.path.join(tmpdir, f"bench_{os.getpid()}.21n"),
                               synth_rinex_records(nchan, self.pos, WEEK, SEC0, seed=35, sets=2), utc, 2)
        self.eph, self.utc, nset = gpsiq.rinex_read(path, 2)
        self.ieph = gpsiq.rinex_select(self.eph, nset, WEEK, SEC0)
        self.svs = [sv for sv in range(32) if self.eph[self.ieph, sv]["vflg"]][:nchan]

    def runahead(self):
        from gpsiq.pipeline import RunAhead
        return RunAhead(self.eph[self.ieph], self.utc, self.svs, WEEK, SEC0, self.pos)

    def quantised(self, b0, b1, fs, nsamp, ra=None):
        """gpsiq_qchan_t rows [b0, b1) of the run (self-seeded), refresh and quantiser fused in C
        (gpsiq_refresh_epochs_quantized); ra as in descriptors()."""
        from gpsiq.abi import QCHAN_DTYPE
        ra = ra or self.runahead()
        ra.seek(b0, self.pos)
        xyz = np.repeat(self.pos[None, :], b1 - b0, axis=0)
        if getattr(self, "_qbuf", None) is None or self._qbuf.shape != (b1 - b0, len(self.svs)):
            self._qbuf = np.empty((b1 - b0, len(self.svs)), dtype=QCHAN_DTYPE)
        return ra.descriptors_quantized(xyz, fs, nsamp, out=self._qbuf)

    def descriptors(self, b0, b1, nthreads=0, ra=None, out=None):
        """gpsiq_chan_t rows [b0, b1) of the run, computing only those (RunAhead.seek); ra: a RunAhead that
        has got as far as some block <= b0 (the rounds of the streamed leg), default a fresh one.  out: where to (default: one
        buffer reused by later calls -- a caller that refreshes ahead of the consumer brings its own two)."""
        from gpsiq.abi import CHAN_DTYPE
        ra = ra or self.runahead()
        ra.seek(b0, self.pos)
        xyz = np.repeat(self.pos[None, :], b1 - b0, axis=0)
        if out is None:
            if getattr(self, "_buf", None) is None or self._buf.shape != (b1 - b0, len(self.svs)):
                self._buf = np.empty((b1 - b0, len(self.svs)), dtype=CHAN_DTYPE)      # reused by later calls
            out = self._buf
        return ra.descriptors(xyz, nthreads=nthreads, out=out)


def main():
    args = parse()
    if args.exact_leg_only:
        ensure_built()
        return exact_leg_only(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args))
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world) if env_world is not None else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run a different job than asked for", file=sys.stderr)
        sys.exit(2)
    ensure_built()
    dry = args.dry_run
    # the ranks of one node share its host cores: each one's library thread pool gets its share
    if world > 1 and "GPSIQ_THREADS" not in os.environ:
        os.environ["GPSIQ_THREADS"] = str(max(1, effective_cpus() // int(os.environ.get("LOCAL_WORLD_SIZE", world))))

    fs, nchan, ss = args.fs, args.nchan, args.sample_size
    nsamp = int(round(fs / 10))                      # NUM_IQ_SAMPLES, reference sdr.h:26
    blk_bytes = 2 * nsamp * ss
    stride = (blk_bytes + 15) & ~15
    nblocks = args.blocks or -(-(2 << 30) // stride)  # per launch; ring >= 2 GiB
    L = max(1, args.launches)
    B = nblocks * L                                   # blocks per GPU per step

    # CPU baseline first (rank 0, N = 1 only): it forks worker processes for the all-cores
    # figure, which must happen before this process holds a GPU context
    cpu_base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not dry:
        from gpsiq.scenario import synth_blocks as _sb
        pat = _sb(64, nchan, seed=args.seed)
        d_cpu = np.concatenate([pat] * (-(-args.cpu_blocks // 64)))[: args.cpu_blocks]
        cpu_base = cpu_baseline(d_cpu, fs, nsamp, ss, args.cpu_blocks)

    import torch
    import gpsiq
    from gpsiq.scenario import synth_blocks
    from gpsiq.shard import max_over_ranks, quantize_own_shard, seed_own_shard, shard_range, torch_all_gather_bytes
    dev = 0
    if not dry:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (libgpsiq has no CPU path)")
        ndev = torch.cuda.device_count()
        # One visible device and a local rank beyond it: a launcher that gives every rank its own device through
        # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES -- ordinal 0 is then this rank's GPU, and the PCI bus ids gathered below
        # tell isolation (all different) from a box with too few GPUs (two ranks on one bus id: exit 5).
        if local_rank >= ndev and ndev > 1 and os.environ.get("GPSIQ_BENCH_SHARE_GPU", "0") in ("", "0"):
            print(f"bench.py: rank {rank} (local {local_rank}) has no GPU of its own: {ndev} visible", file=sys.stderr)
            sys.exit(3)
        dev = local_rank % ndev                       # the modulo: per-rank visibility (above), or GPSIQ_BENCH_SHARE_GPU=1
        torch.cuda.set_device(dev)
    dist = None
    backend = "gloo" if dry else os.environ.get("GPSIQ_BENCH_BACKEND", "nccl")    # nccl == RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    xdev = "cuda" if (backend == "nccl" and not dry) else "cpu"
    gather = torch_all_gather_bytes(dist, xdev)
    variant = gpsiq.variants()[args.variant]
    placement = None
    if world > 1:
        # Where every rank runs, before anything of libgpsiq touches a device: the first collective of the job is this 512-byte
        # all-gather, so a broken RCCL / gloo shows up HERE, named, and not as a hang in the middle of the bench; two ranks that
        # resolved to one device (a launcher that did not set LOCAL_RANK, a box with fewer GPUs than ranks) fail the run.
        me = {"rank": rank, "local_rank": local_rank, "device": None if dry else dev,
              "pci_bus_id": None if dry else pci_bus_id(torch, dev), "host": os.uname().nodename[-64:],
              "cpus_granted": effective_cpus(), "GPSIQ_THREADS": os.environ.get("GPSIQ_THREADS"),
              "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")}
        if me["visible"]:
            me["visible"] = me["visible"][:64]           # every rank sends the same 512 bytes whatever the names are
        try:
            placement = [json.loads(b.decode()) for b in gather(json.dumps(me).ljust(512).encode()[:512])]
        except Exception as ex:
            print(f"bench.py: rank {rank}: the first collective ({backend}) failed before libgpsiq was used: {type(ex).__name__}: {ex}", file=sys.stderr)
            sys.exit(4)
        seen = {}
        for pl in placement:
            key = (pl["host"], pl["pci_bus_id"] if pl["pci_bus_id"] else (pl["device"], pl.get("visible")))
            if not dry and key in seen and os.environ.get("GPSIQ_BENCH_SHARE_GPU", "0") in ("", "0"):
                if rank == 0:
                    print(f"bench.py: ranks {seen[key]} and {pl['rank']} resolve to the same device {key}: refusing to run", file=sys.stderr)
                sys.exit(5)
            seen[key] = pl["rank"]
        if rank == 0:
            for pl in placement:
                print("[bench] rank {rank}: host {host}, local rank {local_rank}, device {device} (PCI {pci_bus_id}), {cpus_granted} CPUs granted, "
                      "GPSIQ_THREADS={GPSIQ_THREADS}".format(**pl), file=sys.stderr)

    # one global timeline of world*B blocks; this rank builds, quantises and seeds ONLY its own rows.
    # Descriptors: distinct code/Doppler state per block for a 64-block pattern tiled over the timeline.
    b0, b1 = shard_range(B * world, rank, world)
    assert b1 - b0 == B
    pattern = synth_blocks(64, nchan, seed=args.seed)
    t_q = time.perf_counter()
    desc_own = pattern[(b0 + np.arange(B)) % len(pattern)]
    q = quantize_own_shard(desc_own, fs, nsamp, rank, world, gather)
    t_q = time.perf_counter() - t_q

    ctx = ring = stream = None
    first = {}
    if not dry:
        ctx = gpsiq.Context(dev)
        ctx.set_descriptors(q)
        ring = torch.empty(nblocks * stride, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        if rank == 0:
            # cold figures, before anything warms the device up: the very first launch (module load, clock
            # ramp) and the mean of the next five -- what a caller that launches once in a while sees
            torch.cuda.synchronize()
            first["first_launch_ms"] = round(ctx.time_launches(0, nblocks, nsamp, ss, ring.data_ptr(), stride, 1, stream=stream, variant=variant), 3)
            first["next5_launch_ms"] = round(ctx.time_launches(nblocks % B, nblocks, nsamp, ss, ring.data_ptr(), stride, 5, stream=stream, variant=variant), 3)

    def step():
        for k in range(L):
            ctx.launch(k * nblocks, nblocks, nsamp, ss, ring.data_ptr(), stride, stream=stream, variant=variant)

    def sync_all():
        if not dry:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not dry:
                torch.cuda.synchronize()

    launch_ms = float("nan")
    if not dry:
        # The GPU needs ~25 ms of load to reach its sustained clock (per-dispatch durations in
        # profiles/r01_final_kernel_trace_stats.txt fall from 4.35 ms to 3.54 ms over the first seven
        # launches), so the device is brought to that state before the W warm-up steps; untimed.
        for k in range(PREHEAT_LAUNCHES):
            ctx.launch((k % L) * nblocks, nblocks, nsamp, ss, ring.data_ptr(), stride, stream=stream, variant=variant)
        for _ in range(args.warmup):
            step()
    import gc
    gc.collect()                                           # no collector pause inside a timed region (as timeit does)
    gc.disable()
    sync_all()
    t0 = time.perf_counter()
    if not dry:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    gc.enable()
    sync_all()
    t_max = max_over_ranks(t_local, dist, device=xdev)
    if not dry:
        launch_ms = e0.elapsed_time(e1) / (args.steps * L)       # HIP events on the launch stream

    # ---- end to end: the same per-GPU block count from nothing, host side sharded as well ---------------
    import tempfile
    e2e = None
    with tempfile.TemporaryDirectory() as td:
        scen = Scenario(nchan, td)
        nb_e = nblocks
        g0, g1 = shard_range(nb_e * world, rank, world)
        best = None
        for _ in range(2):                                # the second pass has warm thread pools and buffers
            gc.collect()
            gc.disable()
            sync_all()
            ta = time.perf_counter()
            q_e = scen.quantised(g0, g1, fs, nsamp)
            tb = time.perf_counter()
            q_e = seed_own_shard(q_e, nsamp, rank, world, gather)
            tc = time.perf_counter()
            if not dry:
                ctx.set_descriptors(q_e)
                td_ = time.perf_counter()
                ctx.launch(0, nb_e, nsamp, ss, ring.data_ptr(), stride, stream=stream, variant=variant)
                torch.cuda.synchronize()
            else:
                td_ = time.perf_counter()
            te = time.perf_counter()
            gc.enable()
            parts = (tb - ta, tc - tb, td_ - tc, te - td_)
            tot = max_over_ranks(te - ta, dist, device=xdev)
            if best is None or tot < best[0]:
                best = (tot, [max_over_ranks(p, dist, device=xdev) for p in parts])
        tot, parts = best
        e2e = {"value": None if dry else round(nb_e * world * nsamp / tot / 1e6, 1), "unit": "Msamples/s",
               "blocks_per_gpu": nb_e, "channels": len(scen.svs), "seconds": round(tot, 5),
               "x_realtime": None if dry else round(nb_e * world * 0.1 / tot, 1),
               "host_refresh_and_quantise_ms": round(parts[0] * 1e3, 2), "seed_exchange_ms": round(parts[1] * 1e3, 2),
               "validate_upload_ms": round(parts[2] * 1e3, 2), "kernel_ms": round(parts[3] * 1e3, 2),
               "host_cpus": effective_cpus(),
               "what": "static receiver (BASELINE config 1/2 geometry), RINEX-derived ephemeris; per rank: RunAhead.seek to its first block, "
                       "nav words rolled over its 30 s epochs (gpsiq_nav_roll), gpsiq_refresh_epochs_quantized of its own blocks only "
                       "(refresh and quantiser in one pass), 32 B/channel carrier-seed all-gather, gpsiq_set_descriptors, one gpsiq_launch; slowest rank, best of 2 passes"}
        # ---- the same chain as a stream of rounds: round m gives this rank the blocks after rank-1's of round m
        # and after all of round m-1; launches are asynchronous, so the host side of round m+1 (refresh, quantise,
        # seed exchange on a gloo group, validate + upload into the other descriptor buffer) runs while the GPU is
        # busy with round m.  What a long scenario costs per block once it is running.
        R = args.rounds
        side = None
        cpu_gather = gather
        exchange = backend if dist is not None else None
        if dist is not None:                              # always a side group (also under gloo): one code path, tested on CPU
            try:                                          # 32 B per channel of host data: a CPU collective, so that it
                side = dist.new_group(backend="gloo")     # does not queue behind the kernels on the device
                cpu_gather = torch_all_gather_bytes(dist, "cpu", group=side)
                exchange = "gloo"
            except Exception as ex:                       # no gloo here: the RCCL group still gives the right answer
                print(f"[bench] no gloo side group ({ex}); seed exchange of the streamed leg over RCCL", file=sys.stderr)
        best_s = None
        passes_s = []
        for _ in range(4):
            ra, hist = scen.runahead(), []
            gc.collect()                                   # as timeit does: a full collection takes 35 ms in this process and,
            gc.disable()                                   # left to itself, lands in the middle of the second pass
            sync_all()
            ta = time.perf_counter()
            host_busy = 0.0
            for m in range(R):
                th = time.perf_counter()
                g0 = (m * world + rank) * nb_e
                q_e = seed_own_shard(scen.quantised(g0, g0 + nb_e, fs, nsamp, ra=ra), nsamp, rank, world, cpu_gather, history=hist)
                host_busy += time.perf_counter() - th
                if not dry:
                    ctx.set_descriptors(q_e)
                    ctx.launch(0, nb_e, nsamp, ss, ring.data_ptr(), stride, stream=stream, variant=variant)
            if not dry:
                torch.cuda.synchronize()
            tot_s = max_over_ranks(time.perf_counter() - ta, dist, device=xdev)
            gc.enable()
            passes_s.append(round(tot_s, 5))
            if best_s is None or tot_s < best_s[0]:
                best_s = (tot_s, max_over_ranks(host_busy, dist, device=xdev), host_busy)
        # which side every rank is bound by: its own host time per round (refresh + quantise + seed exchange, with the
        # threads it was given) against its kernel time per round
        mine = {"rank": rank, "host_ms_per_round": round(best_s[2] / R * 1e3, 3),
                "kernel_ms_per_round": None if dry else round(launch_ms * nb_e / nblocks, 3),
                "threads": int(os.environ.get("GPSIQ_THREADS", "0")) or effective_cpus()}
        mine["bound"] = None if dry else ("host" if mine["host_ms_per_round"] > mine["kernel_ms_per_round"] else "kernel")
        per_rank = [json.loads(b.decode()) for b in gather(json.dumps(mine).ljust(200).encode())]
        e2e["streamed"] = {"value": None if dry else round(R * nb_e * world * nsamp / best_s[0] / 1e6, 1), "unit": "Msamples/s",
                           "per_rank": per_rank,
                           "bound": None if dry else ("host" if any(r["bound"] == "host" for r in per_rank) else "kernel"),
                           "rounds": R, "blocks_per_gpu_per_round": nb_e, "seconds": round(best_s[0], 5), "seconds_each_pass": passes_s,
                           "x_realtime": None if dry else round(R * nb_e * world * 0.1 / best_s[0], 1),
                           "host_refresh_and_quantise_ms_per_round": round(best_s[1] / R * 1e3, 2), "seed_exchange": exchange,
                           "what": "the chain above over one continuous timeline in rounds, launches asynchronous: the host side of round "
                                   "m+1 overlaps the kernel of round m (double-buffered descriptor sets); slowest rank, best of 4 passes"}
        # the refresh needs ~6 host threads per GPU to stay under the kernel: what this host grants per rank
        thr = int(os.environ.get("GPSIQ_THREADS", "0")) or effective_cpus()
        e2e["streamed"]["expected_bound_at_this_host"] = {"threads_per_rank": thr, "threads_needed_per_gpu": 6,
                                                           "expected": "kernel" if thr >= 6 else "host", "host_share_of_kernel_rate": round(min(1.0, thr / 6.0), 2)}
        # what the rounds of a long run are bound by on the slowest rank (the serial batch above is host + kernel by construction)
        e2e["bound"] = e2e["streamed"]["bound"]
        ref_sharded = None
        if world > 1 and not args.no_extra:
            ref_sharded = reference_sharded_leg(ctx, ring, stream, args, rank, world, dist, cpu_gather, dry, xdev)
        if not dry:
            ctx.set_descriptors(q)

    extra = {}
    if not dry and rank == 0 and world == 1 and not args.no_extra:
        extra = extra_legs(ctx, ring, stream, args, first)
    if not dry and world == 1 and os.environ.get("GPSIQ_BENCH_NO_RCCL_SELFTEST", "0") in ("", "0"):
        # also under --no-extra: the driver's N = 1 run must have executed the collective stack of its N > 1 runs
        extra["rccl_selftest"] = rccl_selftest(dev)
        # and the library still works next to an initialised-and-destroyed RCCL: one more launch, timed
        ctx.set_descriptors(q)                            # the extra legs left their own sets resident
        extra["rccl_selftest"]["launch_after_ms"] = round(ctx.time_launches(0, nblocks, nsamp, ss, ring.data_ptr(), stride, 1, stream=stream, variant=variant), 3)
    if not dry and args.sweep and rank == 0:
        sweep_legs(ctx, ring, stream, args, q, nblocks, nsamp, ss, stride)

    if rank == 0:
        samples_step = B * nsamp * world
        value = None if dry else samples_step * args.steps / t_max / 1e6
        alg_bytes = nblocks * blk_bytes                 # algorithmic: 2 or 4 B per complex sample, reads ~0
        # which tile kernel ran: plain adds for int8 always, for int16 when no block's sum of (int)(250*|gain|) exceeds 32767
        forced_packed = os.environ.get("GPSIQ_NO_FAST", "0") not in ("", "0")
        plain_add = (ss == 1 or int(np.floor(250.0 * np.abs(q["gain"])).sum(axis=1).max()) <= 32767) and not forced_packed
        core_cycles = 23.85 if plain_add else 26.7
        traffic, stale_traffic = replayed_traffic(f"{int(fs)}_{nchan}_{ss}_{nblocks}")
        out = {
            "metric": "IQ Msamples/s @16ch int8" if (nchan == 16 and ss == 1) else f"IQ Msamples/s @{nchan}ch int{8 * ss}",
            "value": None if dry else round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{fs / 1e6:g} Msps int{8 * ss} IQ, {nchan} channels, {L} launches x {nblocks} distinct 0.1 s blocks per GPU per step "
                                   f"({B * 0.1:.0f} s of signal per GPU and step, {nblocks * stride / 2**30:.2f} GiB ring), time-sharded x{world}",
                       "fs_hz": fs, "channels": nchan, "sample_bytes": ss, "blocks_per_launch": nblocks, "launches_per_step": L,
                       "blocks_per_gpu_per_step": B, "samples_per_block": nsamp, "variant": args.variant,
                       "nco_mode": "fixed (GPSIQ_NCO_FIXED: 59/56-bit closed form, bit-exact vs the oracle; the reference-identical "
                                   "model is measured in `reference_nco`)",
                       "host_threads_per_rank": int(os.environ.get("GPSIQ_THREADS", "0")) or effective_cpus(),
                       "preheat_launches": PREHEAT_LAUNCHES, "x_realtime": None if dry else round(value * 1e6 / fs, 1),
                       "host_quantise_own_shard_ms": round(t_q * 1e3, 1)},
            "end_to_end": e2e,
        }
        if dry:
            out["dry_run"] = True
        else:
            out["roofline"] = roofline_obj(alg_bytes, launch_ms, traffic)
            if stale_traffic:
                out["roofline"]["stale_profile"] = True     # profiles/pmc_traffic.json is of another kernel than the loaded library's
            # informational: the limiter actually hit.  Per (channel, 64-sample row)
            # and SIMD the row-kernel core costs, with the single-instruction rates measured on this
            # chip (profiles/r01_ubench_valu_encodings.txt: 4.3 nominal cycles for SGPR-operand / VOP3 /
            # SDWA / packed forms, 4.4 for v_lshl_add_u64, 2.5 for plain VOP2),
            #   plain-add kernels (all sums inside int16): 3 x 4.3 + 4.3 / 2 + 2 x 4.4 = 23.85 cycles
            #   packed kernels (larger gains):             3 x 4.3 + 2 x 2.5 + 2 x 4.4 = 26.7 cycles
            # peak = every SIMD of 256 CUs issuing only that core at the 2.4 GHz maximum clock.
            cnt = counters_obj(f"{int(fs)}_{nchan}_{ss}_{nblocks}", launch_ms) or {}
            # not a roofline: the kernel against the cost of its OWN instruction stream (profiles/r06_synth_tile_row_loop.txt)
            cnt["own_stream_efficiency"] = {"note": "launch time vs the issue cost of the kernel's own seven-instruction core at the measured "
                                            "single-instruction rates; says how close the kernel runs to its stream, not that the stream is minimal",
                                            "core_cycles": core_cycles,
                                            "frac": round(nblocks * nsamp * nchan / (launch_ms * 1e-3) / (256 * 4 * 64 * 2.4e9 / core_cycles), 4)}
            out["counters"] = cnt
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        if extra:
            if "reference_nco" in extra:
                out["reference_nco"] = extra.pop("reference_nco")
                # the mode whose output IS the reference's, beside the headline: whole gpsiq_generate_batch call at the same workload
                out["exact_mode_value"] = out["reference_nco"]["value"]
                out["exact_mode_unit"] = "Msamples/s"
                out["exact_mode_roofline"] = out["reference_nco"]["roofline"]
                hl = out["reference_nco"]["legs"]["2M6_int8_16ch"]
                out["exact_mode_bound"] = hl["bound"]
                out["exact_mode_call_over_kernel"] = hl["call_over_kernel"]
                out["exact_mode_value_best_call"] = round(hl["value"] * hl["call_ms_median_of_12"] / hl["call_ms_best"], 1)
                out["exact_mode_what"] = "median of 12 gpsiq_generate_batch calls in GPSIQ_NCO_REFERENCE at the headline's blocks per launch, descriptors in pageable memory"
            out["extra"] = extra
        if ref_sharded is not None:
            out["reference_nco"] = dict(ref_sharded, nco_mode="reference (GPSIQ_NCO_REFERENCE), time-sharded: see `what`",
                                        value=ref_sharded["legs"]["2M6_int8_16ch"]["value"], unit="Msamples/s")
            out["exact_mode_value"], out["exact_mode_unit"] = out["reference_nco"]["value"], "Msamples/s"
            out["exact_mode_bound"] = ref_sharded["legs"]["2M6_int8_16ch"]["bound"]
        if placement is not None:
            out["placement"] = placement
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def pci_bus_id(torch, dev):
    """PCI bus id of a device, as a string; None where this torch does not say."""
    try:
        p = torch.cuda.get_device_properties(dev)
        if hasattr(p, "pci_bus_id"):
            return "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    except Exception:
        pass
    return None


def rccl_selftest(dev, backend="nccl"):
    """Pre-flight of the N > 1 branch on the one GPU of the default run: everything `bench.py --gpus N` asks of
    torch.distributed -- init_process_group("nccl", device_id=...) (RCCL), a float64 all_reduce(MAX) and a uint8 all_gather on
    device tensors (gpsiq/shard.py), barrier, a gloo side group under the RCCL default group and an all_gather on it, a
    device-tensor send/recv to itself being impossible at world size 1 left out, destroy_process_group -- as a world of ONE
    rank, in this process, AFTER gpsiq.Context has been created and used: libgpsiq and torch (with RCCL inside) must share one
    HIP runtime.  Never fatal: the outcome goes into the JSON."""
    import socket
    import torch
    import torch.distributed as dist
    from gpsiq.shard import max_over_ranks, torch_all_gather_bytes
    xd = "cuda" if backend == "nccl" else "cpu"            # backend "gloo": the CPU tests' walk through the same statements
    out = {"ok": False, "backend": "nccl (RCCL)" if backend == "nccl" else backend, "world_size": 1, "after_gpsiq_context": backend == "nccl"}
    if dist.is_initialized():
        out["error"] = "torch.distributed already initialised"
        return out
    t0 = time.perf_counter()
    step = "free port"
    try:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        step = "init_process_group(nccl, device_id)"
        if backend == "nccl":
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        t1 = time.perf_counter()
        step = "all_reduce(MAX) float64 on the device"
        tt = torch.tensor([1.25], dtype=torch.float64, device=xd)     # max_over_ranks' statement (it short-cuts a world of one)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        assert float(tt.item()) == 1.25 and max_over_ranks(1.25, dist, device=xd) == 1.25
        step = "all_gather uint8 on the device"
        t = torch.arange(512, dtype=torch.int32).to(torch.uint8).to(xd) ^ 0x5a
        outs = [torch.empty_like(t)]
        dist.all_gather(outs, t)
        assert torch.equal(outs[0], t)
        step = "barrier"
        dist.barrier()
        if xd == "cuda":
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        step = "new_group(gloo) beside the RCCL group"
        side = dist.new_group(backend="gloo")
        step = "all_gather on the gloo side group"
        c = torch.arange(512, dtype=torch.int32).to(torch.uint8)
        outs = [torch.empty_like(c)]
        dist.all_gather(outs, c, group=side)
        assert torch.equal(outs[0], c)
        assert torch_all_gather_bytes(dist, "cpu", group=side)(b"seed") == [b"seed"]
        t3 = time.perf_counter()
        step = "destroy_process_group"
        dist.destroy_process_group()
        out.update(ok=True, ms=round((time.perf_counter() - t0) * 1e3, 1), init_ms=round((t1 - t0) * 1e3, 1),
                   device_collectives_ms=round((t2 - t1) * 1e3, 1), gloo_side_group_ms=round((t3 - t2) * 1e3, 1),
                   rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None)
    except Exception as ex:                                # the headline must survive a broken collective stack
        out["error"] = f"{step}: {type(ex).__name__}: {ex}"[:400]
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
    return out


def reference_sharded_leg(ctx, ring, stream, args, rank, world, dist, cpu_gather, dry, xdev):
    """GPSIQ_NCO_REFERENCE (the model whose output IS the reference's) time-sharded over the ranks, per rank: its own blocks'
    descriptors -> gpsiq/shard.py reference_own_shard: the carrier chain sharded by TIME (reference_chain_by_time: two all-gathers
    of 56 B per slot for the estimate of where the range starts, the certified map of every own block on this rank's GPU, the
    true states relayed rank to rank, 12 B per slot), gpsiq_reference_seeded over its own blocks -> gpsiq_set_descriptors +
    gpsiq_set_patches -> one gpsiq_launch.  Reported per rank with what bounds it; `value` = all ranks' samples / slowest rank's
    time, best of 2 passes."""
    import gpsiq
    from gpsiq.abi import NCO_REFERENCE  # noqa: F401
    from gpsiq.scenario import synth_blocks
    from gpsiq.shard import max_over_ranks, reference_chain_by_time, reference_own_shard, shard_range
    out = {"legs": {}}
    pat = synth_blocks(64, args.nchan, seed=args.seed)
    for label, fs_r, ss_r, nb_r in (("2M6_int8_16ch", args.fs, args.sample_size, 2000), ("25M_int16_16ch", 25e6, 2, 200)):
        ns_r = int(round(fs_r / 10))
        blk_r = 2 * ns_r * ss_r
        if ring is not None:
            nb_r = min(nb_r, ring.numel() // blk_r)
        b0, b1 = shard_range(nb_r * world, rank, world)            # nb_r blocks per rank: weak scaling, like the headline
        d_own = pat[np.arange(b0, b1) % 64]
        best = None
        for _ in range(2):
            if dist is not None:
                dist.barrier()
            parts = {}
            t0 = time.perf_counter()

            def timed_gather(b, _p=parts):
                t = time.perf_counter()
                r = cpu_gather(b)
                _p["exchange"] = _p.get("exchange", 0.0) + time.perf_counter() - t
                return r
            host_stage = 0.0
            if dry:
                q_r, patches, _, _ = reference_own_shard(d_own, fs_r, ns_r, rank, world, timed_gather, by_time=True, ctx=None)
                t1 = t2 = time.perf_counter()
                npatch_ = len(patches)
            else:
                # the chain by time (level 1 on this rank's GPU, the true states relayed rank to rank), then the rank's own blocks
                # from their start states: quantiser, evaluation, synthesis and patches on the device (gpsiq_generate_seeded)
                start_own, _, _ = reference_chain_by_time(gpsiq.chain_inputs(d_own), fs_r, ns_r, rank, world, timed_gather, ctx=ctx)
                t1 = time.perf_counter()
                ctx.generate_seeded(d_own, ns_r, fs_r, ss_r, start_own, device_ptr=ring.data_ptr())
                t2 = time.perf_counter()
                host_stage = ctx.device_eval_host_ms() * 1e-3
                npatch_ = 0
            t3 = time.perf_counter()
            tot = max_over_ranks(t3 - t0, dist, device=xdev)
            if best is None or tot < best[0]:
                best = (tot, t1 - t0, parts.get("exchange", 0.0), host_stage, t3 - t1, npatch_)
        tot, host, exch, host_stage, render, npatch = best
        mine = {"rank": rank, "chain_by_time_ms": round((host - exch) * 1e3, 3), "exchange_ms": round(exch * 1e3, 3),
                "render_call_ms": None if dry else round(render * 1e3, 3), "render_host_stage_ms": round(host_stage * 1e3, 3),
                "threads": int(os.environ.get("GPSIQ_THREADS", "0")) or effective_cpus()}
        mine["bound"] = None if dry else ("host" if (host + host_stage) > (render - host_stage) else "kernel")
        per_rank = [json.loads(b.decode()) for b in cpu_gather(json.dumps(mine).ljust(320).encode())]
        out["legs"][label] = {"workload": f"{fs_r / 1e6:g} Msps int{8 * ss_r}, {args.nchan} ch, {nb_r} blocks per GPU, one serial pass "
                                          "(chain by time: summaries + maps on the GPU + link relayed; then gpsiq_generate_seeded: quantiser, evaluation, synthesis, patches on the device)",
                              "value": None if dry else round(nb_r * world * ns_r / tot / 1e6, 1), "unit": "Msamples/s",
                              "x_realtime": None if dry else round(nb_r * world * 0.1 / tot, 1), "seconds": round(tot, 5), "per_rank": per_rank,
                              "bound": None if dry else ("host" if any(r["bound"] == "host" for r in per_rank) else "kernel")}
    out["what"] = ("GPSIQ_NCO_REFERENCE time-sharded over the ranks: every rank walks the carrier chain of ITS OWN blocks (level 1, the certified "
                   "map of every block, on its GPU; level 2 an addition per block on the host, relayed rank to rank), then renders them from their "
                   "start states with quantiser, evaluation and patches on the device; what is left on the host per rank is the relayed link and the pack of its descriptors")
    return out


def exact_leg_shapes(args, ring_bytes):
    """(label, fs, sample bytes, blocks): the headline workload's size first (what `value` renders per launch), then round 5's sizes."""
    ns = int(round(args.fs / 10))
    nb_head = min(4130, ring_bytes // (2 * ns * args.sample_size))
    return (("2M6_int8_16ch", args.fs, args.sample_size, nb_head), ("2M6_int8_16ch_2000_blocks", args.fs, args.sample_size, 2000),
            ("25M_int16_16ch", 25e6, 2, 200))


def exact_leg(ctx, ring, stream, pat, fs_r, ss_r, nb_r, nchan, calls=12, parts=True):
    """One GPSIQ_NCO_REFERENCE workload: the whole batch call (median and best of `calls`) for descriptors in pageable, page-locked and
    device memory, the host stages of each, kernel + patches alone, and (parts) the host-evaluation path and the chain's parts."""
    import torch
    import gpsiq
    ns_r = int(round(fs_r / 10))
    blk_r = 2 * ns_r * ss_r
    nb_r = min(nb_r, ring.numel() // blk_r)
    d_r = pat[np.arange(nb_r) % 64]
    raw = torch.from_numpy(d_r.view(np.uint8).reshape(-1).copy())
    pinned, resident = raw.pin_memory(), raw.cuda()
    srcs = (("pageable", d_r), ("page_locked", (pinned.data_ptr(), nb_r, nchan)), ("device", (resident.data_ptr(), nb_r, nchan)))

    def timed(src, n):
        ctx.generate_batch(src, ns_r, fs_r, ss_r, device_ptr=ring.data_ptr())
        dts = []
        for _ in range(n):
            t1 = time.perf_counter()
            ctx.generate_batch(src, ns_r, fs_r, ss_r, device_ptr=ring.data_ptr())
            dts.append(time.perf_counter() - t1)
        dts.sort()
        return dts[len(dts) // 2], dts[0]

    keep = {k: os.environ.get(k) for k in ("GPSIQ_EVAL", "GPSIQ_CHAIN")}
    os.environ.pop("GPSIQ_EVAL", None)
    os.environ.pop("GPSIQ_CHAIN", None)
    s0 = gpsiq.device_eval_stats()
    by_mem = {}
    for kind, src in srcs:
        med, best = timed(src, calls)
        by_mem[kind] = {"call_ms_median": round(med * 1e3, 3), "call_ms_best": round(best * 1e3, 3), "value": round(nb_r * ns_r / med / 1e6, 1),
                        "host_stage_ms": round(ctx.device_eval_host_ms(), 3)}
    s1 = gpsiq.device_eval_stats()
    q_r, patches, _ = gpsiq.reference_blocks(d_r, fs_r, ns_r)
    ctx.set_descriptors(q_r)
    ctx.set_patches(patches)
    ctx.time_launches(0, nb_r, ns_r, ss_r, ring.data_ptr(), blk_r, 3, stream=stream)
    km = min(ctx.time_launches(0, nb_r, ns_r, ss_r, ring.data_ptr(), blk_r, 5, stream=stream) for _ in range(2))
    for v in by_mem.values():
        v["call_over_kernel"] = round(v["call_ms_median"] / km, 3)
        v["gpus_the_host_can_feed"] = round(km / v["host_stage_ms"], 1) if v["host_stage_ms"] > 0 else None
    med = by_mem["pageable"]["call_ms_median"] * 1e-3
    calls_n = max(1, s1[0] - s0[0])
    traffic, stale = replayed_traffic(f"exact_{int(fs_r)}_{nchan}_{ss_r}_{nb_r}")
    leg = {"workload": f"{fs_r / 1e6:g} Msps int{8 * ss_r}, {nchan} ch, {nb_r} blocks, gpsiq_generate_batch -> device memory",
           "value": round(nb_r * ns_r / med / 1e6, 1), "unit": "Msamples/s", "x_realtime": round(nb_r * 0.1 / med, 1),
           "call_ms": by_mem["pageable"]["call_ms_median"], "call_ms_median_of_12": by_mem["pageable"]["call_ms_median"], "call_ms_best": by_mem["pageable"]["call_ms_best"],
           "call_by_descriptor_memory": by_mem,
           "kernel_and_patches_ms": round(km, 3), "patched_samples": int(len(patches)),
           "call_over_kernel": by_mem["pageable"]["call_over_kernel"],
           "host_stage_ms": by_mem["pageable"]["host_stage_ms"], "gpus_the_host_can_feed": by_mem["pageable"]["gpus_the_host_can_feed"],
           "bound": "host" if by_mem["pageable"]["host_stage_ms"] > km else "kernel",
           "evaluated_on": "device" if s1[1] > s0[1] else "host threads",
           "per_call": {"block_channel_pairs": (s1[1] - s0[1]) // calls_n, "pairs_to_the_host_walker": round((s1[2] - s0[2]) / calls_n, 2),
                        "slots_repaired_by_the_host": round((s1[3] - s0[3]) / calls_n, 2), "calls_that_fell_back_to_the_host_path": s1[5] - s0[5]},
           "roofline": roofline_obj(nb_r * blk_r, med * 1e3, traffic),
           "roofline_kernel_only": roofline_obj(nb_r * blk_r, km)}
    if stale:
        leg["roofline"]["stale_profile"] = True
    if parts:
        os.environ["GPSIQ_EVAL"] = "host"                       # rounds 4-5's path in the same process: chain link + evaluation on host threads
        med_h, best_h = timed(d_r, 6)
        os.environ.pop("GPSIQ_EVAL", None)
        cin = gpsiq.chain_inputs(d_r)
        starts, _, _ = gpsiq.reference_chain(cin, fs_r, ns_r)
        chain_ms = min(gpsiq.chain_maps(cin, fs_r, ns_r, ctx=ctx)[2] for _ in range(5))
        maps = gpsiq.chain_maps(cin, fs_r, ns_r, ctx=ctx)[0]
        c0 = gpsiq.chain_stats()
        linked_starts = gpsiq.chain_link(cin, maps, fs_r, ns_r)[0]
        c1 = gpsiq.chain_stats()
        leg["call_ms_host_evaluation"] = {"median": round(med_h * 1e3, 3), "best": round(best_h * 1e3, 3)}
        leg["chain"] = {"level1_device_kernels_ms": round(chain_ms, 3), "blocks_linked_through_their_map": int(c1[0] - c0[0]),
                        "blocks_walked_from_their_true_start": int(c1[1] - c0[1]), "equal_to_the_serial_chain": bool(linked_starts.tobytes() == starts.tobytes())}
    for k, v in keep.items():                                    # (a caller's own switches survive the leg)
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    del pinned, resident
    return leg


def exact_leg_only(args):
    """bench.py --exact-leg-only: the GPSIQ_NCO_REFERENCE call at the headline size under the environment given (the parent runs
    this with GPSIQ_THREADS=2), one JSON line."""
    import torch
    import gpsiq
    from gpsiq.abi import NCO_REFERENCE
    from gpsiq.scenario import synth_blocks
    ctx = gpsiq.Context(0)
    ring = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
    ctx.set_nco_mode(NCO_REFERENCE)
    pat = synth_blocks(64, args.nchan, seed=args.seed)
    shapes = exact_leg_shapes(args, ring.numel())
    label, fs_r, ss_r, nb_r = shapes[0]
    leg = exact_leg(ctx, ring, torch.cuda.current_stream().cuda_stream, pat, fs_r, ss_r, nb_r, args.nchan, calls=8, parts=True)
    out = {"host_threads": int(os.environ.get("GPSIQ_THREADS", "0")) or effective_cpus(), "workload": leg["workload"]}
    for k in ("value", "call_ms_median_of_12", "call_ms_best", "kernel_and_patches_ms", "call_over_kernel", "host_stage_ms", "gpus_the_host_can_feed", "bound",
              "call_by_descriptor_memory", "call_ms_host_evaluation"):
        out[k] = leg[k]
    if os.environ.get("GPSIQ_BENCH_ALL_EXACT_LEGS"):            # (development: every shape of the default run's reference_nco leg, on its own)
        for label, fs_r, ss_r, nb_r in shapes[1:]:
            l2 = exact_leg(ctx, ring, torch.cuda.current_stream().cuda_stream, pat, fs_r, ss_r, nb_r, args.nchan, calls=8, parts=False)
            out[label] = {k: l2[k] for k in ("call_by_descriptor_memory", "kernel_and_patches_ms")}
    print(json.dumps(out), flush=True)
    ctx.close()


def extra_legs(ctx, ring, stream, args, first):
    """Short legs outside the timed headline (rank 0, N = 1): every other rate this repository quotes,
    measured in the driver's own run."""
    import torch
    import gpsiq
    from gpsiq.abi import NCO_FIXED, NCO_REFERENCE
    from gpsiq.scenario import synth_blocks
    ex = dict(first)
    ring_bytes = ring.numel()

    def kernel_leg(fs, nchan, ss, note):
        nsamp = int(round(fs / 10))
        blk = 2 * nsamp * ss
        stride = (blk + 15) & ~15
        nb = ring_bytes // stride
        pat = synth_blocks(min(64, nb), nchan, seed=args.seed)
        d = pat[np.arange(nb) % len(pat)]
        qq, _ = gpsiq.quantize_blocks(d, fs, nsamp)
        ctx.set_descriptors(qq)
        ctx.time_launches(0, nb, nsamp, ss, ring.data_ptr(), stride, 8, stream=stream)       # warm: the headline is pre-heated too
        ms = min(ctx.time_launches(0, nb, nsamp, ss, ring.data_ptr(), stride, 8, stream=stream) for _ in range(3))
        return {"workload": f"{fs / 1e6:g} Msps int{8 * ss}, {nchan} ch, {nb} blocks per launch ({note})",
                "value": round(nb * nsamp / ms / 1e3, 1), "unit": "Msamples/s", "x_realtime": round(nb * 0.1 / (ms * 1e-3), 1),
                "roofline": roofline_obj(nb * blk, ms)}

    ex["reference_build_3M_int8_12ch"] = kernel_leg(3.0e6, 12, 1, "the reference as shipped: TX_SAMPLERATE 3 Msps, MAX_CHAN 12 (sdr.h:21, gps.h:36)")
    ex["cfg2_2M6_int8_12ch"] = kernel_leg(2.6e6, 12, 1, "BASELINE config 2")
    ex["cfg4_2M6_int16_16ch"] = kernel_leg(2.6e6, 16, 2, "BASELINE config 4 format")
    ex["cfg3_10M_int16_16ch"] = kernel_leg(10e6, 16, 2, "BASELINE config 3")
    ex["cfg5_25M_int16_16ch"] = kernel_leg(25e6, 16, 2, "BASELINE config 5, one GPU's share")

    # the drop-in calls (host destination): PCIe-bound, never `value`
    fs, nchan, ss = args.fs, args.nchan, args.sample_size
    nsamp = int(round(fs / 10))
    blk = 2 * nsamp * ss
    nb_h = 512
    pat = synth_blocks(64, nchan, seed=args.seed)
    d_h = pat[np.arange(nb_h) % 64]
    pinned = torch.empty(nb_h * blk, dtype=torch.uint8).pin_memory()
    for label, chunk in (("host_dst_batch", None), ("host_dst_batch_unchunked", "0")):
        if chunk is None:
            os.environ.pop("GPSIQ_PIECE_BLOCKS", None)
        else:
            os.environ["GPSIQ_PIECE_BLOCKS"] = chunk
        ctx.generate_batch(d_h, nsamp, fs, ss, host_ptr=pinned.data_ptr())
        dt = float("inf")
        for _ in range(3):
            t1 = time.perf_counter()
            ctx.generate_batch(d_h, nsamp, fs, ss, host_ptr=pinned.data_ptr())
            dt = min(dt, time.perf_counter() - t1)
        ex[label] = {"what": f"gpsiq_generate_batch -> page-locked host memory, {nb_h} blocks ({'kernel of piece k+1 overlaps the copy of piece k' if chunk is None else 'one kernel, then one copy'})",
                     "value": round(nb_h * nsamp / dt / 1e6, 1), "unit": "Msamples/s", "pcie_GBps": round(nb_h * blk / dt / 1e9, 2),
                     "x_realtime": round(nb_h * 0.1 / dt, 1)}
    os.environ.pop("GPSIQ_PIECE_BLOCKS", None)
    # what "PCIe-bound" means on this box: the same bytes as ONE plain device-to-host copy into the same page-locked buffer
    nbytes_h = nb_h * blk
    raw = float("inf")
    for _ in range(4):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pinned.copy_(ring[:nbytes_h], non_blocking=True)
        torch.cuda.synchronize()
        raw = min(raw, time.perf_counter() - t1)
    ex["pcie_d2h_raw"] = {"what": f"one device-to-host copy of the same {nbytes_h} bytes into the same page-locked buffer (no kernel)",
                          "GBps": round(nbytes_h / raw / 1e9, 2)}
    for label in ("host_dst_batch", "host_dst_batch_unchunked"):
        ex[label]["frac_of_raw_copy"] = round(ex[label]["pcie_GBps"] / ex["pcie_d2h_raw"]["GBps"], 3)

    pageable = np.empty(blk, dtype=np.uint8)             # what a malloc'd staging block is: the copy off the device goes through a bounce buffer
    for label, mode, dst_ptr in (("block_call", NCO_FIXED, pinned.data_ptr()), ("block_call_reference_nco", NCO_REFERENCE, pinned.data_ptr()),
                                 ("block_call_pageable", NCO_FIXED, pageable.ctypes.data)):
        ctx.set_nco_mode(mode)
        lat, carr = [], None
        for k in range(50):
            ch1 = d_h[k].copy()
            if carr is not None:
                ch1["carr_phase"] = carr
            t1 = time.perf_counter()
            _, carr = ctx.generate_block(ch1, nsamp, fs, ss, host_ptr=dst_ptr)
            lat.append(time.perf_counter() - t1)
        lat = sorted(lat[10:])
        ex[label] = {"what": "gpsiq_generate_block -> " + ("PAGEABLE host memory (malloc instead of gpsiq_host_alloc: what getting the staging block "
                             "of INTEGRATION.md wrong costs)" if label.endswith("pageable") else "page-locked host memory") +
                             ", one 0.1 s block per call, carr_phase handed back in (the patched gps thread's call)",
                     "median_us": round(lat[len(lat) // 2] * 1e6, 1),
                     "max_us": round(lat[-1] * 1e6, 1), "x_realtime": round(0.1 / lat[len(lat) // 2], 1)}
    # the same call without waiting: 50 blocks queued back to back, one wait at the end
    ctx.set_nco_mode(NCO_FIXED)
    many = [torch.empty(blk, dtype=torch.uint8).pin_memory() for _ in range(8)]
    carr = None
    for rep in range(2):
        t1 = time.perf_counter()
        for k in range(50):
            ch1 = d_h[k].copy()
            if carr is not None:
                ch1["carr_phase"] = carr
            carr = ctx.generate_block_async(ch1, nsamp, fs, ss, many[k % 8].data_ptr())
        ctx.wait()
        dt = time.perf_counter() - t1
    ex["block_call_async"] = {"what": "gpsiq_generate_block_async, 50 blocks queued back to back into page-locked buffers + gpsiq_wait",
                              "us_per_block": round(dt / 50 * 1e6, 1), "x_realtime": round(0.1 * 50 / dt, 1)}
    ctx.set_nco_mode(NCO_REFERENCE)
    carr = None
    for rep in range(2):
        t1 = time.perf_counter()
        for k in range(50):
            ch1 = d_h[k].copy()
            if carr is not None:
                ch1["carr_phase"] = carr
            carr = ctx.generate_block_async(ch1, nsamp, fs, ss, many[k % 8].data_ptr())
        ctx.wait()
        dt = time.perf_counter() - t1
    ex["block_call_async_reference_nco"] = {"what": "the same in GPSIQ_NCO_REFERENCE (the host walks block k+1's carrier while block k is rendered and copied)",
                                            "us_per_block": round(dt / 50 * 1e6, 1), "x_realtime": round(0.1 * 50 / dt, 1)}
    ctx.set_nco_mode(NCO_FIXED)
    # GPSIQ_NCO_REFERENCE, the model whose output IS the reference's (T2 = 0): the whole gpsiq_generate_batch call into device
    # memory, at the headline workload's size and two smaller ones.  Quantiser, carrier chain (level 1 lanes, level 2 as a scan)
    # and candidate evaluation run on the device (csrc/gpsiq_evaldev.cpp); what the host still does is reported as host_stage_ms.
    ctx.set_nco_mode(NCO_REFERENCE)
    ref = {"nco_mode": "reference (GPSIQ_NCO_REFERENCE: the reference's double accumulators reproduced exactly; whole runs equal "
                       "the reference program's file, tests/test_reference_program.py, tests/test_config4.py)",
           "host_cpus": effective_cpus(), "legs": {}}
    for label, fs_r, ss_r, nb_r in exact_leg_shapes(args, ring_bytes):
        ref["legs"][label] = exact_leg(ctx, ring, stream, pat, fs_r, ss_r, nb_r, args.nchan, calls=12, parts=True)
    # the same call on a thread-starved host (GPSIQ_THREADS=2: what a rank of an 8-rank job gets from a 16-CPU box), in a child
    # process (the pool is sized once per process)
    try:
        import subprocess
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--exact-leg-only", "--nchan", str(args.nchan), "--fs", str(args.fs),
                                "--sample-size", str(args.sample_size), "--seed", str(args.seed)],
                               env=dict(os.environ, GPSIQ_THREADS="2", GPSIQ_BENCH_NO_RCCL_SELFTEST="1"), capture_output=True, text=True, timeout=240)
        ref["two_host_threads"] = json.loads(child.stdout.strip().splitlines()[-1])
    except Exception as ex_:                               # the figure is informational: the bench line does not depend on it
        ref["two_host_threads"] = {"error": f"{type(ex_).__name__}: {ex_}"[:200]}
    # the same model from nothing: RINEX-derived ephemeris, static receiver -> per-block host refresh (gpsiq_refresh_epochs, the
    # double-precision descriptors the reference's host code would hand over) -> gpsiq_generate_batch in GPSIQ_NCO_REFERENCE,
    # in rounds of 1000 blocks chained through carr_phase as a run-ahead host does (host/gpsiq_runahead.c)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        scen = Scenario(args.nchan, td)
        ns_e = int(round(args.fs / 10))
        nb_e, rounds_e = 1000, 4
        carr = np.zeros(len(scen.svs))
        best = float("inf")
        for _ in range(3):
            ra = scen.runahead()
            t1 = time.perf_counter()
            t_host = 0.0
            for m in range(rounds_e):
                th0 = time.perf_counter()
                d_e = scen.descriptors(m * nb_e, (m + 1) * nb_e, ra=ra)
                if m:
                    d_e["carr_phase"][0] = carr                  # the accumulator the last round handed out (all slots keep their satellite)
                t_host += time.perf_counter() - th0
                ctx.generate_batch(d_e, ns_e, args.fs, args.sample_size, device_ptr=ring.data_ptr(), carr_out=carr)
            dt = time.perf_counter() - t1
            if dt < best:
                best, best_host = dt, t_host
        ref["end_to_end"] = {"what": f"RINEX-derived ephemeris, static receiver: {rounds_e} rounds of gpsiq_refresh_epochs ({nb_e} blocks, double-precision "
                                     "descriptors) + gpsiq_generate_batch in GPSIQ_NCO_REFERENCE, chained through carr_phase (serial: refresh, then the call)",
                             "value": round(rounds_e * nb_e * ns_e / best / 1e6, 1), "unit": "Msamples/s", "x_realtime": round(rounds_e * nb_e * 0.1 / best, 1),
                             "seconds": round(best, 5), "host_refresh_ms_per_round": round(best_host / rounds_e * 1e3, 3), "channels": len(scen.svs)}
        # ... and as a run-ahead host runs it: the refresh of round m+1 on a second host thread while round m is in
        # the library (both release the interpreter lock), block 0's carr_phase patched in when round m has handed its accumulators back
        import queue
        import threading
        rounds_s = 8
        best_s = float("inf")
        todo, ready = queue.SimpleQueue(), queue.SimpleQueue()
        two = [np.empty((nb_e, len(scen.svs)), dtype=gpsiq.abi.CHAN_DTYPE) for _ in range(2)]
        state = {}

        def refresher():                                   # one thread for the whole leg: a thread per round costs 0.1 ms of a 1 ms round
            while True:
                m = todo.get()
                if m is None:
                    return
                ready.put(scen.descriptors(m * nb_e, (m + 1) * nb_e, ra=state["ra"], out=two[m & 1]))

        worker = threading.Thread(target=refresher, daemon=True)
        worker.start()
        for _ in range(3):
            state["ra"] = scen.runahead()
            todo.put(0)
            d_e = ready.get()
            t1 = time.perf_counter()
            for m in range(rounds_s):
                if m + 1 < rounds_s:
                    todo.put(m + 1)
                if m:
                    d_e["carr_phase"][0] = carr
                ctx.generate_batch(d_e, ns_e, args.fs, args.sample_size, device_ptr=ring.data_ptr(), carr_out=carr)
                if m + 1 < rounds_s:
                    d_e = ready.get()
            best_s = min(best_s, time.perf_counter() - t1)
        todo.put(None)
        worker.join()
        ref["end_to_end"]["streamed"] = {"what": f"the same over {rounds_s} rounds with the refresh of round m+1 on a second host thread while round m is in gpsiq_generate_batch "
                                                 "(the first round's refresh outside the timed region: a run-ahead host is always one round ahead)",
                                         "value": round(rounds_s * nb_e * ns_e / best_s / 1e6, 1), "unit": "Msamples/s",
                                         "x_realtime": round(rounds_s * nb_e * 0.1 / best_s, 1), "seconds": round(best_s, 5)}
    head_leg = ref["legs"]["2M6_int8_16ch"]
    ref["value"] = head_leg["value"]
    ref["unit"] = "Msamples/s"
    ref["roofline"] = head_leg["roofline"]
    ref["what"] = ("whole gpsiq_generate_batch call in GPSIQ_NCO_REFERENCE into device memory, MEDIAN of 12 calls (the best beside it), descriptors in "
                   "pageable host memory as the Python binding hands them over (page-locked and device-resident descriptors: legs.*.call_by_descriptor_memory).  "
                   "The call: descriptors packed (pool) and uploaded; per piece chain_prepare -> quantize_est -> synthesis from the ESTIMATED start states; beside "
                   "it, on high-priority streams, chain_lanes -> chain_link_scan (the TRUE start states) -> eval_blocks (patches); apply_patches at the end.  "
                   "host_stage_ms: what is left on host threads (pack of pageable rows, repair of slots whose maps do not apply, the host walker's share of the "
                   "evaluation); gpus_the_host_can_feed = kernel_and_patches_ms / host_stage_ms; call_over_kernel = call_ms_median / kernel_and_patches_ms.  "
                   "call_ms_host_evaluation: the same call on rounds 4-5's path (GPSIQ_EVAL=host: chain link + evaluation on host threads)")
    ex["reference_nco"] = ref
    ctx.set_nco_mode(NCO_FIXED)
    # the batch call with a device destination from double-precision descriptors: host quantiser + upload + kernel
    stride = (blk + 15) & ~15
    nb_d = min(ring_bytes // stride, 4130)
    d_d = pat[np.arange(nb_d) % 64]
    ctx.generate_batch(d_d, nsamp, fs, ss, device_ptr=ring.data_ptr())
    dt = float("inf")
    for _ in range(8):                                   # best of 8, as the reference_nco legs
        t1 = time.perf_counter()
        ctx.generate_batch(d_d, nsamp, fs, ss, device_ptr=ring.data_ptr())
        dt = min(dt, time.perf_counter() - t1)
    ex["device_dst_batch"] = {"what": f"gpsiq_generate_batch -> device memory, {nb_d} blocks from gpsiq_chan_t in pageable host memory (pack on the pool + upload; "
                                      "quantiser and carrier prefix on the device; kernel), best of 8 calls",
                              "value": round(nb_d * nsamp / dt / 1e6, 1), "unit": "Msamples/s", "x_realtime": round(nb_d * 0.1 / dt, 1)}
    # the same with the gpsiq_chan_t rows already in device memory (a host that refreshes on the device, or uploaded them earlier):
    # nothing of the call is left on the host but its launches
    d_res = torch.from_numpy(d_d.view(np.uint8).reshape(-1).copy()).cuda()
    src = (d_res.data_ptr(), nb_d, d_d.shape[1])
    ctx.generate_batch(src, nsamp, fs, ss, device_ptr=ring.data_ptr())
    ts = []
    for _ in range(12):
        t1 = time.perf_counter()
        ctx.generate_batch(src, nsamp, fs, ss, device_ptr=ring.data_ptr())
        ts.append(time.perf_counter() - t1)
    ts.sort()
    ex["device_dst_batch"]["descriptors_on_device"] = {"value": round(nb_d * nsamp / ts[len(ts) // 2] / 1e6, 1), "unit": "Msamples/s", "call_ms_median_of_12": round(ts[len(ts) // 2] * 1e3, 3),
                                                       "call_ms_best": round(ts[0] * 1e3, 3)}
    del d_res
    return ex


def sweep_legs(ctx, ring, stream, args, q, nblocks, nsamp, ss, stride):
    import gpsiq
    ctx.set_descriptors(q)
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
    # the host refresh that feeds the kernel (gpsiq_refresh_batch, reference gps.c:2731-2765):
    # blocks per second on this host, 1 thread and all threads
    from gpsiq.scenario import circle_track, llh_to_ecef, synth_constellation, synth_iono, synth_tracks
    pos = llh_to_ecef(*TOKYO_LLH)
    eph = synth_constellation(args.nchan, pos, SEC0, seed=3)
    xyz = circle_track(pos, 200000)
    for nt, scalar in ((1, True), (1, False), (0, False)):
        # scalar: one channel at a time (the round-2 form, GPSIQ_REFRESH_SCALAR); else the stages of the orbit / range /
        # ionosphere chain looped over the block's channels, so that the libm calls of neighbouring channels overlap
        if scalar:
            os.environ["GPSIQ_REFRESH_SCALAR"] = "1"
        else:
            os.environ.pop("GPSIQ_REFRESH_SCALAR", None)
        dt = float("inf")
        for _ in range(2):
            trk = synth_tracks(args.nchan, WEEK, SEC0)
            gpsiq.track_init(eph, synth_iono(), WEEK, SEC0, xyz[0], trk)
            t1 = time.perf_counter()
            gpsiq.refresh_batch(eph, synth_iono(), WEEK, SEC0, xyz[1:], trk, nthreads=nt)
            dt = min(dt, time.perf_counter() - t1)
        print(f"[refresh] gpsiq_refresh_batch {args.nchan} ch, {len(xyz) - 1} blocks, threads={'all' if nt == 0 else nt}, "
              f"{'one channel at a time' if scalar else 'channels staged together'}: "
              f"{(len(xyz) - 1) / dt / 1e3:.1f} kblocks/s = {(len(xyz) - 1) * 0.1 / dt:.0f}x real time, "
              f"{dt / (len(xyz) - 1) / args.nchan * 1e6:.3f} us per channel-block", file=sys.stderr)


if __name__ == "__main__":
    main()
