cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python bench.py --sweep --no-cpu-baseline --steps 5 ) > gpurun_out/bench_q.log 2>&1; grep -E "device-dst|host-dst|block\]" gpurun_out/bench_q.log | tail; tail -3 gpurun_out/bench_q.log | cut -c1-300
