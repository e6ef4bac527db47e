set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
( timeout 900 python -m pytest tests -m "not gpu" -x -q 2>&1 | tail -5 ) > gpurun_out/pytest_cpu_on_gpubox.log 2>&1
tail -3 gpurun_out/pytest_cpu_on_gpubox.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
( timeout 600 python bench.py --sweep --no-cpu-baseline --steps 10 ) > gpurun_out/bench1.log 2>&1; grep -E "refresh|host-dst|sweep" gpurun_out/bench1.log
# the multi-rank code path (self-launch, host-side sharding, seed exchange, barrier, max-over-ranks) on this 1-GPU box:
# 2 ranks, gloo, both on GPU 0 (GPSIQ_BENCH_SHARE_GPU=1 lifts the one-GPU-per-rank check for exactly this)
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/bench_2rank_gloo.log 2>&1; tail -1 gpurun_out/bench_2rank_gloo.log | cut -c1-400
