# round 3, final GPU session at HEAD: whole GPU suite, smoke, the default bench line, a 2-rank run on the one GPU, and a kernel
# trace of the reference-NCO batch call (synth kernel + apply_patches per piece)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14 ) > gpurun_out/r3final_pytest_gpu.log 2>&1; tail -10 gpurun_out/r3final_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r3final_smoke.log 2>&1; tail -1 gpurun_out/r3final_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r3final_bench.json 2> gpurun_out/r3final_bench.err; tail -2 gpurun_out/r3final_bench.err
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/r3final_bench_2rank.log 2>&1; tail -1 gpurun_out/r3final_bench_2rank.log | cut -c1-600
cat > /tmp/ref_trace.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((25e6, 2, 200), (2.6e6, 1, 2000)):
    d = pat[np.arange(nb) % 64]
    for _ in range(4):
        ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr())
PY
REPO=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_ref/kt -o kt -- python /tmp/ref_trace.py > $REPO/gpurun_out/prof_ref_kt.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
for db in glob.glob("gpurun_out/prof_ref/kt/*.db"):
    con = sqlite3.connect(db)
    lines = ["== rocprofv3 --kernel-trace --stats of 4 x gpsiq_generate_batch in GPSIQ_NCO_REFERENCE: 200 blocks at 25 Msps int16 (8 pieces of 26), then 2000 blocks at 2.6 Msps int8 (8 pieces of 256) =="]
    lines.append(f"{'kernel':<70} {'calls':>6} {'total_ns':>14} {'avg_ns':>14} {'pct':>7}")
    for name, calls, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{name[:70]:<70} {calls:>6} {tot:>14.0f} {avg:>14.1f} {pct:>7.2f}")
    lines.append("per dispatch of the last call of each workload: kernel, grid, duration_ns, gap to the previous dispatch's end (ns)")
    rows = list(con.execute("select name,grid_x,start,end from kernels order by start"))
    prev = None
    for i, (name, grid, st, en) in enumerate(rows):
        if i >= len(rows) - 40:
            lines.append(f"  {name[:56]:<56} {grid:>9} {en - st:>9} {'' if prev is None else st - prev:>9}")
        prev = en
    open("gpurun_out/r3final_reference_kernel_trace.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
PY
find gpurun_out/prof_ref -name "*.db" -size +8M -delete
