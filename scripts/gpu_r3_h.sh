# round 3, eighth GPU session: soaks at HEAD (every kernel variant against the oracle; the reference NCO against the reference's own loop)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 400 python tests/soak_fixed.py 300 ) > gpurun_out/r3h_soak_fixed.txt 2>&1; tail -2 gpurun_out/r3h_soak_fixed.txt
( timeout 700 python tests/soak_reference.py 600 ) > gpurun_out/r3h_soak_reference.txt 2>&1; tail -2 gpurun_out/r3h_soak_reference.txt
