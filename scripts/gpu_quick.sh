cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/pytest_q.log 2>&1; tail -3 gpurun_out/pytest_q.log
# throughput of the C generator-thread / fifo / sink-thread program, 1000 blocks = 100 s of signal into tmpfs
python - <<'PY' > gpurun_out/play_rate.log 2>&1
import os, struct, subprocess, sys, time
sys.path.insert(0, "multi-sdr-gps-sim_amd")
import numpy as np
from gpsiq.abi import CHAN_DTYPE
from gpsiq.scenario import synth_blocks
fs, ns, nb, nc = 2.6e6, 260000, 1000, 16
d = synth_blocks(nb, nc, seed=5)
for ss, sink in ((1, "iqfile"), (1, "hackrf"), (2, "pluto")):
    with open("/dev/shm/desc.bin", "wb") as f:
        f.write(struct.pack("<8sIIIId", b"GPSIQD1\0", nb, nc, ss, ns, fs)); f.write(d.tobytes())
    t = time.perf_counter()
    r = subprocess.run(["multi-sdr-gps-sim_amd/host/gpsiq_play", "/dev/shm/desc.bin", "/dev/shm/out.bin", sink], capture_output=True, text=True)
    dt = time.perf_counter() - t
    print(f"[play] {sink} int{8*ss}: {r.stdout.strip()} rc={r.returncode} in {dt:.2f} s wall (incl. ~0.3 s start-up) = {nb/dt:.0f} blocks/s = {nb*0.1/dt:.0f}x real time")
    os.remove("/dev/shm/out.bin")
PY
cat gpurun_out/play_rate.log
