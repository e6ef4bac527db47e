# kernel trace of the streamed end-to-end leg: are the kernels of consecutive rounds back to back?
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_streamed
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $REPO/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --rounds 16 > $OUT/kt.log 2>&1
grep -c "^{" $OUT/kt.log
cd $REPO && python scripts/streamed_gaps.py gpurun_out/prof_streamed 16 r02 | tee gpurun_out/streamed_gaps.txt
find $OUT -name "*.db" -size +8M -delete
mkdir -p gpurun_out/profiles_out; cp profiles/r02_streamed_leg_gaps.txt gpurun_out/profiles_out/
