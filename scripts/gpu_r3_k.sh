# round 3, eleventh GPU session: the pipelined fixed-point batch; whole suite; bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r3k_pytest_gpu.log 2>&1; tail -4 gpurun_out/r3k_pytest_gpu.log
cat > /tmp/batch_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")      # 4130 blocks of 520 000 B are 116 KB more than 2 GiB
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((2.6e6, 1, 4130), (25e6, 2, 214), (10e6, 2, 536)):
    ns = int(fs) // 10
    d = pat[np.arange(nb) % 64]
    for env in ({"GPSIQ_BATCH_PIECE_BLOCKS": "0"}, {}, {"GPSIQ_BATCH_PIECE_BLOCKS": "512"}, {"GPSIQ_BATCH_PIECE_BLOCKS": "2048"}):
        os.environ.pop("GPSIQ_BATCH_PIECE_BLOCKS", None); os.environ.update(env)
        ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
        print(f"fs {fs/1e6:g} M, {nb} blocks, {env or 'default pieces'}: {best*1e3:.2f} ms = {nb*ns/best/1e9:.1f} G samples/s", flush=True)
PY
( timeout 600 python /tmp/batch_ab.py ) > gpurun_out/r3k_batch_pieces.txt 2>&1; cat gpurun_out/r3k_batch_pieces.txt
( timeout 900 python bench.py ) > gpurun_out/r3k_bench.json 2> gpurun_out/r3k_bench.err; tail -2 gpurun_out/r3k_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3k_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["end_to_end"]["streamed"]["value"], d["extra"]["device_dst_batch"])
for k,v in d["reference_nco"]["legs"].items(): print(k, v["value"], v["call_ms"], v["host_walk_and_candidates_ms"], v["kernel_and_patches_ms"], v["bound"])
PY
