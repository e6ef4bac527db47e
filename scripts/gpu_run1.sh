set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
( timeout 600 python bench.py --sweep ) > gpurun_out/bench1.log 2>&1; tail -8 gpurun_out/bench1.log
( timeout 300 python bench.py --sweep --sample-size 2 --fs 1e7 --no-cpu-baseline --steps 5 ) > gpurun_out/bench_10M16.log 2>&1; tail -6 gpurun_out/bench_10M16.log
( hipcc --offload-arch=gfx950 -O3 scripts/ubench.hip -o /tmp/ubench && timeout 300 /tmp/ubench ) > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log
