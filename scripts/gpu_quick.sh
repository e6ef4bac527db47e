cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/pytest_q.log 2>&1; tail -4 gpurun_out/pytest_q.log
( timeout 600 python bench.py --sweep --no-cpu-baseline --steps 5 ) > gpurun_out/bench_q.log 2>&1; grep -E "pipeline|device-dst|Traceback|Error" -A3 gpurun_out/bench_q.log | head -20
