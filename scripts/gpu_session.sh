# One GPU-box session: gpurun -- 'bash scripts/gpu_session.sh STEP [STEP ...]'.  Every step writes under gpurun_out/<tag>_*.
# Steps: chain (tests + timing of the device carrier chain), tests (pytest -m gpu), refnco (the reference-NCO GPU tests),
#        bench (the default bench line), prof (rocprofv3 passes of the default bench: scripts/gpu_prof.sh), smoke.
TAG=${TAG:-r05}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for step in "$@"; do
  case $step in
    chain)
      ( timeout 900 python -m pytest tests/test_chain_parallel.py -m gpu -x -q -s 2>&1 | tail -25 ) > gpurun_out/${TAG}_chain_tests.log 2>&1; tail -5 gpurun_out/${TAG}_chain_tests.log
      ( timeout 600 python scripts/chain_timing.py ) > gpurun_out/${TAG}_chain_timing.log 2>&1; cat gpurun_out/${TAG}_chain_timing.log ;;
    refnco)
      ( timeout 1500 python -m pytest tests/test_gpu_reference_nco.py tests/test_config5_shares.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_refnco_tests.log 2>&1; tail -4 gpurun_out/${TAG}_refnco_tests.log ;;
    tests)
      ( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_gpu.log ;;
    smoke)
      ( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log ;;
    bench)
      ( timeout 900 python bench.py ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json ;;
    prof)
      PROF_TAG=$TAG bash scripts/gpu_prof.sh ;;
    *) echo "unknown step $step" ;;
  esac
done
