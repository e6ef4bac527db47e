# round 4: what ONE RANK OF AN 8-RANK RUN gets from the host -- (1) the default bench with two host threads, (2) eight ranks
# sharing the box's one GPU over gloo (the host side is the real thing: 8 processes on the 16-CPU quota; the kernels take turns)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( GPSIQ_THREADS=2 timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r4n_bench_2threads.json 2> gpurun_out/r4n.err; tail -1 gpurun_out/r4n.err
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 ) > gpurun_out/r4n_bench_8rank_shared_gpu.log 2>&1; tail -3 gpurun_out/r4n_bench_8rank_shared_gpu.log | cut -c1-600
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4n_bench_2threads.json").read().strip().splitlines()[-1])
print("2 threads: value", d["value"], "streamed", json.dumps(d["end_to_end"]["streamed"]))
for k, v in d["reference_nco"]["legs"].items():
    print(k, v["value"], "call", v["call_ms"], "host", v["host_walk_and_candidates_ms"], "chain", v["host_chain_only_ms"], "eval", v["host_evaluation_only_ms"], v["bound"])
for ln in open("gpurun_out/r4n_bench_8rank_shared_gpu.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("8 ranks on one GPU: value", d["value"], "ms_per_step", d["ms_per_step"])
        print("streamed", json.dumps(d["end_to_end"]["streamed"]))
        print("reference_nco", json.dumps(d["reference_nco"])[:6000])
PY
