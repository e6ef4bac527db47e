# round 3, second GPU session: the whole GPU suite, smoke, the default bench line, the sweep (refresh A/B), a 2-rank run on the one GPU
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -60 ) > gpurun_out/r3b_pytest_gpu.log 2>&1
tail -30 gpurun_out/r3b_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r3b_smoke.log 2>&1; tail -1 gpurun_out/r3b_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; tail -c 9000 gpurun_out/r3b_bench.json; tail -5 gpurun_out/r3b_bench.err
( timeout 600 python bench.py --sweep --no-cpu-baseline --no-extra --steps 5 ) > gpurun_out/r3b_sweep.log 2>&1; grep -E "refresh|sweep" gpurun_out/r3b_sweep.log
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/r3b_bench_2rank.log 2>&1; tail -1 gpurun_out/r3b_bench_2rank.log | cut -c1-3000
