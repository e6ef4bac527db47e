#!/usr/bin/env python3
"""The bench line of an N-rank run whose ranks SHARE one GPU (gpu_session.sh 8rank) -> a per-rank summary for profiles/."""
import json
import sys

j = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
n = j["n_gpus"]
print(f"{n} ranks on ONE MI355X over gloo (GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo python bench.py --gpus {n} ...): the N > 1 code path at current code,")
print("the host side as it is on such a box (every rank its share of the CPU quota), the kernels of the ranks taking turns on the one device --")
print("readiness evidence, NOT a scaling curve: `value` and every kernel time below are those of a shared device.")
print(f"value {j['value']} {j['unit']} aggregate, {j['ms_per_step']} ms per step; host threads per rank {j['config']['host_threads_per_rank']}")
st = j["end_to_end"]["streamed"]
print(f"end_to_end.streamed {st['value']} Msamples/s aggregate, bound {st['bound']}, seed exchange over {st['seed_exchange']}; per rank host / kernel ms per round:")
for r in st["per_rank"]:
    print(f"  rank {r['rank']}: host {r['host_ms_per_round']} ms, kernel {r['kernel_ms_per_round']} ms, threads {r['threads']}, bound {r['bound']}")
ref = j.get("reference_nco", {})
for label, leg in ref.get("legs", {}).items():
    print(f"reference_nco (time-sharded: chain by time, then gpsiq_generate_seeded with the evaluation on the device) {label}: {leg['value']} Msamples/s aggregate, {leg['seconds'] * 1e3:.1f} ms, bound {leg['bound']}")
    print("  per rank: chain by time (summaries, maps on the GPU, relayed link) / exchange / render call / host stages inside the render, ms:")
    for r in leg["per_rank"]:
        print(f"  rank {r['rank']}: {r['chain_by_time_ms']} / {r['exchange_ms']} / {r['render_call_ms']} / {r['render_host_stage_ms']}, threads {r['threads']}, bound {r['bound']}")
for p in j.get("placement", []):
    print(f"placement rank {p['rank']}: device {p['device']} (PCI {p['pci_bus_id']}), {p['cpus_granted']} CPUs granted, GPSIQ_THREADS={p['GPSIQ_THREADS']}")
