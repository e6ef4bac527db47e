# round 3: the carrier walk's microbenchmark on the GPU box's host (EPYC), A/B of the number of binades walked by plain additions
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in 4 3 2 1; do
  g++ -O3 -std=c++17 -ffp-contract=off -DKLOW=$k -I multi-sdr-gps-sim_amd/csrc -o /tmp/uw$k scripts/ubench_walk.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread
  echo "== kLow = $k =="; /tmp/uw$k
done > gpurun_out/r3o_ubench_walk.txt 2>&1
cat gpurun_out/r3o_ubench_walk.txt
