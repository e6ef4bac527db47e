set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --sweep --no-cpu-baseline --steps 10 ) > gpurun_out/bench1.log 2>&1; tail -9 gpurun_out/bench1.log
( timeout 300 python bench.py --sweep --sample-size 2 --fs 1e7 --no-cpu-baseline --steps 5 ) > gpurun_out/bench_10M16.log 2>&1; tail -9 gpurun_out/bench_10M16.log
( timeout 300 python bench.py --sweep --sample-size 2 --fs 2.5e7 --no-cpu-baseline --steps 5 ) > gpurun_out/bench_25M16.log 2>&1; tail -9 gpurun_out/bench_25M16.log
( timeout 300 python bench.py --sweep --nchan 12 --no-cpu-baseline --steps 5 ) > gpurun_out/bench_12ch.log 2>&1; tail -9 gpurun_out/bench_12ch.log
( timeout 300 python bench.py --sweep --nchan 8 --no-cpu-baseline --steps 5 ) > gpurun_out/bench_8ch.log 2>&1; tail -9 gpurun_out/bench_8ch.log
