# One GPU-box session:   gpurun -- 'bash scripts/gpu_session.sh STEP [STEP ...]'      (TAG=r06 by default)
# Every step writes under gpurun_out/<TAG>_*; profiles are summarised into profiles/<TAG>_* by scripts/prof_summary.py.
#   tests    pytest -m gpu, the whole suite                      refnco   the reference-NCO GPU tests only
#   chain    device carrier chain: its tests, scripts/chain_timing.py   chainab  scripts/chain_ab.py (knobs + piece timelines)
#   chainpmc rocprofv3 counters of the chain kernels (scripts/chain_pmc.py)      soak     tests/soak_chain_device.py, 3 seeds
#   cpu      pytest -m "not gpu" on the box's host               smoke    __graft_entry__.smoke()
#   bench    the default bench line                              sweep    bench.py --sweep (all kernel variants, refresh sweep)
#   2rank    bench.py --gpus 2 over gloo on the box's one GPU (the N > 1 code path; GPSIQ_BENCH_SHARE_GPU=1)      8rank   the same with eight ranks
#   prof     rocprofv3 passes of the default bench (scripts/gpu_prof.sh)        profcfg  the same for configs 3 and 5
#   rates    throughput over the BASELINE / front-end sample rates
#   exactprof rocprofv3 kernel trace + HBM traffic of the GPSIQ_NCO_REFERENCE batch call (scripts/exact_call_prof.py)
#   soakeval tests/soak_device_eval.py, 3 seeds: device evaluation == host evaluation (== the reference's own loop) on random runs
#   eval     the device evaluation: tests/test_gpu_device_eval.py, scripts/eval_timing.py (all threads, then GPSIQ_THREADS=2)
TAG=${TAG:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for step in "$@"; do
  case $step in
    chain)
      ( timeout 900 python -m pytest tests/test_chain_parallel.py -m gpu -x -q -s 2>&1 | tail -25 ) > gpurun_out/${TAG}_chain_tests.log 2>&1; tail -5 gpurun_out/${TAG}_chain_tests.log
      ( timeout 600 python scripts/chain_timing.py ) > gpurun_out/${TAG}_chain_timing.log 2>&1; grep -v "trace\] descriptors" gpurun_out/${TAG}_chain_timing.log ;;
    chainab)
      ( timeout 500 python scripts/chain_ab.py ) > gpurun_out/${TAG}_chain_ab.log 2>&1; grep -v "trace\] descriptors" gpurun_out/${TAG}_chain_ab.log ;;
    chainpmc)
      ( cd /tmp && export TMPDIR=/tmp && O=$GRAFT_REPO_ROOT/gpurun_out/chain_pmc && mkdir -p $O &&
        timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
          -d $O/a -o pmc -- python $GRAFT_REPO_ROOT/scripts/chain_pmc.py > $O/a.log 2>&1; grep Msps $O/a.log ) ;;
    eval)
      ( timeout 1500 python -m pytest tests/test_gpu_device_eval.py -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/${TAG}_eval_tests.log 2>&1; tail -25 gpurun_out/${TAG}_eval_tests.log
      ( timeout 600 python scripts/eval_timing.py ) > gpurun_out/${TAG}_eval_timing.log 2>&1; grep -v "trace\] descriptors" gpurun_out/${TAG}_eval_timing.log
      ( timeout 600 python scripts/eval_timing.py 2 ) > gpurun_out/${TAG}_eval_timing_2threads.log 2>&1; grep -v "trace\] " gpurun_out/${TAG}_eval_timing_2threads.log ;;
    exactprof)
      ( cd /tmp && export TMPDIR=/tmp && O=$GRAFT_REPO_ROOT/gpurun_out/exact_prof && rm -rf $O && mkdir -p $O &&
        echo '{"calls": 6, "blocks": [2000, 4130], "nsamp": 260000, "ss": 1}' > $O/meta.json &&
        EXACT_PROF_25M=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/exact_call_prof.py 6 > $O/kt.log 2>&1; tail -3 $O/kt.log
        timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python $GRAFT_REPO_ROOT/scripts/exact_call_prof.py 6 > $O/pmc_write.log 2>&1
        timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/scripts/exact_call_prof.py 6 > $O/pmc_fetch.log 2>&1
        cd $GRAFT_REPO_ROOT && python scripts/exact_call_summary.py gpurun_out/exact_prof $TAG > $O/summary.log 2>&1; tail -60 $O/summary.log
        find $O -name "*.db" -size +8M -delete
        mkdir -p gpurun_out/profiles_out; cp profiles/${TAG}_exact_call_* profiles/pmc_traffic.json gpurun_out/profiles_out/ ) ;;
    refnco)
      ( timeout 1500 python -m pytest tests/test_gpu_reference_nco.py tests/test_config5_shares.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_refnco_tests.log 2>&1; tail -4 gpurun_out/${TAG}_refnco_tests.log ;;
    soak)
      for s in 1 2 3; do ( timeout $(( ${SOAK_SECONDS:-50} + 150 )) python tests/soak_chain_device.py ${SOAK_SECONDS:-50} $s ) 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${TAG}_soak_chain_device.txt ;;
    soakeval)
      for s in 1 2 3; do ( timeout $(( ${SOAK_SECONDS:-90} + 310 )) python tests/soak_device_eval.py ${SOAK_SECONDS:-90} $s ) 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${TAG}_soak_device_eval.txt ;;
    tests)
      ( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_gpu.log ;;
    cpu)
      ( timeout 900 python -m pytest tests -m "not gpu" -x -q 2>&1 | tail -5 ) > gpurun_out/${TAG}_pytest_cpu_on_gpubox.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_cpu_on_gpubox.log ;;
    smoke)
      ( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log ;;
    bench)
      ( timeout 900 python bench.py ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json ;;
    sweep)
      ( timeout 600 python bench.py --sweep --no-cpu-baseline --steps 10 ) > gpurun_out/${TAG}_bench_sweep.log 2>&1; grep -E "refresh|host-dst|sweep" gpurun_out/${TAG}_bench_sweep.log ;;
    2rank)
      ( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/${TAG}_bench_2rank_gloo.log 2>&1
      tail -1 gpurun_out/${TAG}_bench_2rank_gloo.log | cut -c1-400 ;;
    8rank)
      # eight ranks SHARING the box's one GPU over gloo: the N > 1 code path at current code with the host side as it is on such a box
      # (8 processes x 2 threads on the 16-CPU quota); the kernels take turns on the one device: readiness evidence, not a scaling curve
      ( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --blocks 1000 --launches 4 --rounds 4 ) > gpurun_out/${TAG}_bench_8rank_gloo.log 2> gpurun_out/${TAG}_bench_8rank_gloo.err
      tail -1 gpurun_out/${TAG}_bench_8rank_gloo.log | cut -c1-300
      python scripts/eight_rank_summary.py gpurun_out/${TAG}_bench_8rank_gloo.log > gpurun_out/${TAG}_8rank_shared_gpu_gloo.txt 2>&1; cat gpurun_out/${TAG}_8rank_shared_gpu_gloo.txt ;;
    prof)
      PROF_TAG=$TAG bash scripts/gpu_prof.sh ;;
    profcfg)
      PROF_TAG=$TAG bash scripts/gpu_prof_cfg.sh ;;
    rates)
      { echo "# python bench.py --no-cpu-baseline --no-extra --steps 30 --fs F --sample-size S --nchan C   (auto variant, one MI355X)"
        echo "# fs_hz sample_bytes channels blocks_per_launch Msamples/s x_realtime kernel_ms HBM_write_GB/s"
        for cfg in "2600000 1 16" "2600000 1 12" "2600000 2 16" "3000000 1 12" "2048000 1 16" "1100000 1 16" "4000000 1 16" "10000000 2 16" "25000000 2 16"; do
          set -- $cfg
          python bench.py --no-cpu-baseline --no-extra --steps 30 --fs $1 --sample-size $2 --nchan $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; r=j['roofline']
print($1, $2, $3, c['blocks_per_launch'], j['value'], c['x_realtime'], r['kernel_ms'], r['achieved'])"
        done; } > gpurun_out/${TAG}_rates.txt 2>&1; cat gpurun_out/${TAG}_rates.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
