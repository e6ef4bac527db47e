# round 3: long soaks at the final HEAD (unused GPU minutes of the round)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1000 python tests/soak_reference.py 900 ) > gpurun_out/r3s2_soak_reference.txt 2>&1; tail -1 gpurun_out/r3s2_soak_reference.txt
( timeout 600 python tests/soak_fixed.py 480 ) > gpurun_out/r3s2_soak_fixed.txt 2>&1; tail -1 gpurun_out/r3s2_soak_fixed.txt
