# round 4: third soak, at the HEAD that changed the host pool's wake-ups and the fixed-point batch's pieces
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python tests/soak_reference.py 180 ) > gpurun_out/r4s3_soak_reference.txt 2>&1; tail -1 gpurun_out/r4s3_soak_reference.txt
( timeout 150 python tests/soak_fixed.py 60 ) > gpurun_out/r4s3_soak_fixed.txt 2>&1; tail -1 gpurun_out/r4s3_soak_fixed.txt
