# round 4: the host-destination batch against one plain D2H copy of the same bytes (what "PCIe-bound" means on this box)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4r_bench.json 2> gpurun_out/r4r_bench.err; tail -2 gpurun_out/r4r_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4r_bench.json").read().strip().splitlines()[-1])
e = d["extra"]
for k in ("pcie_d2h_raw", "host_dst_batch", "host_dst_batch_unchunked", "block_call", "block_call_reference_nco", "block_call_async", "device_dst_batch"):
    print(k, json.dumps({a: b for a, b in e[k].items() if a != "what"}))
PY
cat > /tmp/hd.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0)
pat = synth_blocks(64, 16, seed=20250215)
for nb, fs, ss in ((512, 2.6e6, 1), (2048, 2.6e6, 1), (200, 25e6, 2)):
    ns = int(fs) // 10; blk = 2 * ns * ss
    d = np.ascontiguousarray(pat[np.arange(nb) % 64])
    pinned = torch.empty(nb * blk, dtype=torch.uint8).pin_memory()
    dev = torch.empty(nb * blk, dtype=torch.uint8, device="cuda")
    raw = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); pinned.copy_(dev, non_blocking=True); torch.cuda.synchronize(); raw = min(raw, time.perf_counter() - t)
    line = "%d blocks %.1f Msps ss%d (%.2f GB): raw copy %.2f GB/s;" % (nb, fs / 1e6, ss, nb * blk / 1e9, nb * blk / raw / 1e9)
    for chunk in (None, "16", "32", "64", "128", "256"):
        if chunk: os.environ["GPSIQ_D2H_CHUNK_BLOCKS"] = chunk
        else: os.environ.pop("GPSIQ_D2H_CHUNK_BLOCKS", None)
        ctx.generate_batch(d, ns, fs, ss, host_ptr=pinned.data_ptr())
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); ctx.generate_batch(d, ns, fs, ss, host_ptr=pinned.data_ptr()); best = min(best, time.perf_counter() - t)
        line += " chunk %s: %.2f GB/s;" % (chunk or "default", nb * blk / best / 1e9)
    print(line)
PY
python /tmp/hd.py 2>&1 | tail -4
