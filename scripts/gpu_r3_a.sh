# round 3, first GPU session: the GPU suite (incl. config 4 as written), smoke, the default bench line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -60 ) > gpurun_out/r3a_pytest_gpu.log 2>&1
tail -25 gpurun_out/r3a_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r3a_smoke.log 2>&1; tail -1 gpurun_out/r3a_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; tail -c 6000 gpurun_out/r3a_bench.json; tail -5 gpurun_out/r3a_bench.err
nproc; cat /sys/fs/cgroup/cpu.max
