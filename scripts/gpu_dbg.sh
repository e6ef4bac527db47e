cd $GRAFT_REPO_ROOT
for m in gpsiq_then_torch gpsiq_ctx_then_torch; do python scripts/gpu_dbg.py $m 2>&1 | grep -v amdgpu.ids; done
readelf -d /usr/local/lib/python3.10/dist-packages/torch/lib/libamdhip64.so | grep -E "SONAME|RUNPATH|RPATH|NEEDED"
readelf -d /usr/local/lib/python3.10/dist-packages/torch/lib/libtorch_hip.so | grep -E "RUNPATH|RPATH|amdhip"
