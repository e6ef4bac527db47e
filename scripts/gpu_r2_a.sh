set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log 2>&1
tail -6 gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
( timeout 600 python bench.py --steps 20 ) > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-600
nproc; cat /sys/fs/cgroup/cpu.max
