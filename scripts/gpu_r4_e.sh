# round 4, fifth GPU session: ramped pieces; reference-NCO piece timings; kernel trace of the reference batch (apply_patches);
# the int16 configurations' profiles with the current kernel file (PROF_TAG=r04)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reference_nco.py tests/test_config5_shares.py -m gpu -q -x --durations=4 2>&1 | tail -12 ) > gpurun_out/r4e_pytest_gpu.log 2>&1; tail -8 gpurun_out/r4e_pytest_gpu.log
cat > /tmp/ref_trace.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((25e6, 2, 200), (10e6, 2, 536), (2.6e6, 1, 2000)):
    d = pat[np.arange(nb) % 64]
    best = 1e9
    for _ in range(8):
        t = time.perf_counter(); ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
    print("fs %.1f: call %.3f ms = %.1f Gsamples/s" % (fs / 1e6, best * 1e3, nb * fs / 10 / best / 1e9), flush=True)
    if "--trace" in sys.argv:
        os.environ["GPSIQ_TRACE"] = "1"
        ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr())
        os.environ.pop("GPSIQ_TRACE")
PY
python /tmp/ref_trace.py --trace 2>&1 | grep -v "trace\] descriptors" | tee gpurun_out/r4e_ref_pieces.txt
GPSIQ_REF_CHUNK_RAMP=0 python /tmp/ref_trace.py 2>&1 | sed 's/^/no ramp: /' | tee -a gpurun_out/r4e_ref_pieces.txt
REPO=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_ref/kt -o kt -- python /tmp/ref_trace.py > $REPO/gpurun_out/prof_ref_kt.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
for db in glob.glob("gpurun_out/prof_ref/kt/*.db"):
    con = sqlite3.connect(db)
    lines = ["== rocprofv3 --kernel-trace --stats of 8 x gpsiq_generate_batch in GPSIQ_NCO_REFERENCE per workload: 200 blocks at 25 Msps int16, 536 blocks at 10 Msps int16, 2000 blocks at 2.6 Msps int8 (ramped pieces) =="]
    lines.append(f"{'kernel':<70} {'calls':>6} {'total_ns':>14} {'avg_ns':>14} {'pct':>7}")
    for name, calls, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{name[:70]:<70} {calls:>6} {tot:>14.0f} {avg:>14.1f} {pct:>7.2f}")
    lines.append("per dispatch, the last 60: kernel, grid, duration_ns, gap to the previous dispatch's end (ns)")
    rows = list(con.execute("select name,grid_x,start,end from kernels order by start"))
    prev = None
    for i, (name, grid, st, en) in enumerate(rows):
        if i >= len(rows) - 60:
            lines.append(f"  {name[:56]:<56} {grid:>9} {en - st:>9} {'' if prev is None else st - prev:>9}")
        prev = en
    open("gpurun_out/r4e_reference_kernel_trace.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))
PY
find gpurun_out/prof_ref -name "*.db" -size +8M -delete
PROF_TAG=r04 bash scripts/gpu_prof_cfg.sh > gpurun_out/r4e_prof_cfg.log 2>&1; tail -4 gpurun_out/r4e_prof_cfg.log
