set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default.log 2>&1; tail -4 gpurun_out/bench_default.log | cut -c1-6000
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/bench_2rank_gloo.log 2>&1; tail -1 gpurun_out/bench_2rank_gloo.log | cut -c1-1500
python bench.py --gpus 8 --steps 2 ; echo "rc=$?"
