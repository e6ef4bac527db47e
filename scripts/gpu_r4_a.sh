# round 4, first GPU session: the GPU suite at the start-of-round state (+ exact T2 lists, kernels id), the default bench line with
# the RCCL self-test (the N > 1 collective stack as a world of one, after gpsiq.Context), and the 2-rank run on the one GPU with the
# RCCL backend (outcome kept whatever it is), host CPU topology
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( nproc; lscpu | head -24; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$ ) > gpurun_out/r4a_cpuinfo.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -q --durations=10 2>&1 | tail -30 ) > gpurun_out/r4a_pytest_gpu.log 2>&1; tail -12 gpurun_out/r4a_pytest_gpu.log
( timeout 900 python bench.py ) > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; tail -3 gpurun_out/r4a_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4a_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "counters stale:", (d.get("counters") or {}).get("stale_profile"))
print("rccl_selftest", json.dumps(d["extra"].get("rccl_selftest")))
PY
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=nccl timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/r4a_bench_2rank_nccl.log 2>&1; echo "exit $?" >> gpurun_out/r4a_bench_2rank_nccl.log; tail -12 gpurun_out/r4a_bench_2rank_nccl.log | cut -c1-400
