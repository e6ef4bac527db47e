# round 4: second soak at HEAD (other seeds), incl. GPSIQ_THREADS=2 (piece-major host side) and the scalar table build
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 400 python tests/soak_reference.py 300 ) > gpurun_out/r4s2_soak_reference.txt 2>&1; tail -1 gpurun_out/r4s2_soak_reference.txt
( GPSIQ_THREADS=2 timeout 200 python tests/soak_reference.py 90 ) > gpurun_out/r4s2_soak_reference_2threads.txt 2>&1; tail -1 gpurun_out/r4s2_soak_reference_2threads.txt
( GPSIQ_WALK_NOBATCH=1 timeout 200 python tests/soak_reference.py 60 ) > gpurun_out/r4s2_soak_reference_scalar.txt 2>&1; tail -1 gpurun_out/r4s2_soak_reference_scalar.txt
