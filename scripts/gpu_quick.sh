cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/pytest_q.log 2>&1; tail -4 gpurun_out/pytest_q.log
( timeout 900 python -m pytest tests -m "not gpu" -x -q 2>&1 | tail -5 ) > gpurun_out/pytest_c.log 2>&1; tail -3 gpurun_out/pytest_c.log
