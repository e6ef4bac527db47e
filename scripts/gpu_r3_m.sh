# round 3, thirteenth GPU session: piece sizes of the fixed-point device-destination batch (the A/B of gpu_r3_k.sh with a ring that is large enough)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
sed -n '/^cat > \/tmp\/batch_ab.py/,/^PY$/p' scripts/gpu_r3_k.sh | sed '1d;$d' > /tmp/batch_ab.py
( timeout 300 python /tmp/batch_ab.py ) > gpurun_out/r3m_batch_pieces.txt 2>&1; cat gpurun_out/r3m_batch_pieces.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py tests/test_host_c.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r3m_pytest.log 2>&1; tail -3 gpurun_out/r3m_pytest.log
