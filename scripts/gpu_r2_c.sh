set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_default.log") if l.startswith("{")][0])
print(d["value"], d["roofline"]["frac"], d["end_to_end"])
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench3.hip -o /tmp/ubench3 && timeout 300 /tmp/ubench3 > gpurun_out/ubench3.txt 2>&1; grep "w/SIMD=4" gpurun_out/ubench3.txt
PROF_TAG=r02 bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1; tail -5 gpurun_out/prof.log
