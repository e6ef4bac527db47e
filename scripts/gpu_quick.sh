cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz" 2>&1 | tail -15 ) > gpurun_out/pytest_q.log 2>&1; tail -15 gpurun_out/pytest_q.log
