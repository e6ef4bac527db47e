cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
