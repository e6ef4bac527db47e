# round 3, third GPU session: ubench3 (the sign-word pricing), the reference-NCO batch call taken apart (GPSIQ_TRACE, piece
# sizes, walk with and without the wrap-to-wrap table), this round's profiles (kernel trace + PMC passes, headline and int16 configs)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench3.hip -o /tmp/ubench3 && timeout 300 /tmp/ubench3 > gpurun_out/r3c_ubench3.txt 2>&1; grep "w/SIMD=4" gpurun_out/r3c_ubench3.txt
cat > /tmp/ref_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((2.6e6, 1, 2000), (25e6, 2, 200), (10e6, 2, 500)):
    ns = int(fs) // 10
    d = pat[np.arange(nb) % 64]
    for env in ({}, {"GPSIQ_REF_CHUNK_BLOCKS": "0"}, {"GPSIQ_REF_CHUNK_BLOCKS": "64"}, {"GPSIQ_REF_CHUNK_BLOCKS": "1000"}):
        for k in ("GPSIQ_REF_CHUNK_BLOCKS",):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())
        best = min((lambda t: (ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr()), time.perf_counter() - t)[1])(time.perf_counter()) for _ in range(4))
        print(f"fs {fs/1e6:g} M, {nb} blocks, {env or 'default pieces'}: {best*1e3:.2f} ms = {nb*ns/best/1e9:.1f} G samples/s", flush=True)
    os.environ.pop("GPSIQ_REF_CHUNK_BLOCKS", None)
    t = time.perf_counter(); gpsiq.reference_blocks(d, fs, ns); th = time.perf_counter() - t
    print(f"   host side alone (gpsiq_reference_batch): {th*1e3:.2f} ms", flush=True)
    os.environ["GPSIQ_TRACE"] = "1"
    ctx.generate_batch(d, ns, fs, ss, device_ptr=ring.data_ptr())
    os.environ.pop("GPSIQ_TRACE")
PY
( timeout 600 python /tmp/ref_ab.py ) > gpurun_out/r3c_reference_pieces.txt 2>&1; cat gpurun_out/r3c_reference_pieces.txt
( GPSIQ_WALK_NOMAP=1 timeout 600 python /tmp/ref_ab.py ) > gpurun_out/r3c_reference_pieces_nomap.txt 2>&1; grep -E "default|host side" gpurun_out/r3c_reference_pieces_nomap.txt
PROF_TAG=r03 bash scripts/gpu_prof.sh > gpurun_out/r3c_prof.log 2>&1; tail -5 gpurun_out/r3c_prof.log
PROF_TAG=r03 bash scripts/gpu_prof_cfg.sh > gpurun_out/r3c_prof_cfg.log 2>&1; tail -5 gpurun_out/r3c_prof_cfg.log
ls gpurun_out/profiles_out
