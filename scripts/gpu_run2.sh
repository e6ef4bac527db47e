set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
( hipcc --offload-arch=gfx950 -O3 scripts/ubench2.hip -o /tmp/ubench2 && timeout 300 /tmp/ubench2 ) > gpurun_out/ubench2.log 2>&1; cat gpurun_out/ubench2.log
