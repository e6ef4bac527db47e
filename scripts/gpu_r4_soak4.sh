# round 4: fourth soak at the final code: the default path and the path that walks every candidate (GPSIQ_NO_DRIFT=1)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 420 python tests/soak_reference.py 300 ) > gpurun_out/r4s4_soak_reference.txt 2>&1; tail -1 gpurun_out/r4s4_soak_reference.txt
( GPSIQ_NO_DRIFT=1 timeout 240 python tests/soak_reference.py 120 ) > gpurun_out/r4s4_soak_reference_nodrift.txt 2>&1; tail -1 gpurun_out/r4s4_soak_reference_nodrift.txt
