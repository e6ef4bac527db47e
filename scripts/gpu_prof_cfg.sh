# rocprofv3 kernel-trace + HBM traffic passes for the int16 configurations (BASELINE configs 3 and 5 per GPU).
# usage: bash scripts/gpu_prof_cfg.sh ; results summarised into profiles/r02_cfg{3,5}_* and profiles/pmc_traffic.json
set -x
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "cfg3 10000000" "cfg5 25000000"; do
  set -- $cfg; name=$1; fs=$2
  OUT=$REPO/gpurun_out/prof_$name
  mkdir -p $OUT
  BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --launches 6 --steps 10 --fs $fs --sample-size 2"
  $BENCH > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-300
  rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
  nb=$(python -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['config']['blocks_per_launch'])")
  ( cd $REPO && PMC_KEY=${fs}_16_2_${nb} python scripts/prof_summary.py gpurun_out/prof_$name ${PROF_TAG:-r02}_$name > $OUT/summary.log 2>&1; tail -2 $OUT/summary.log )
  find $OUT -name "*.db" -size +20M -delete
done
cd $REPO; mkdir -p gpurun_out/profiles_out; cp profiles/${PROF_TAG:-r02}_cfg* profiles/pmc_traffic.json profiles/pmc_counters.json gpurun_out/profiles_out/
du -sh gpurun_out
