# round 3, twelfth GPU session: piece sizes of the fixed-point device-destination batch (the A/B of gpu_r3_k.sh with a ring that is large enough)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
sed -n '/^cat > \/tmp\/batch_ab.py/,/^PY$/p' scripts/gpu_r3_k.sh | sed '1d;$d' > /tmp/batch_ab.py
( timeout 300 python /tmp/batch_ab.py ) > gpurun_out/r3l_batch_pieces.txt 2>&1; cat gpurun_out/r3l_batch_pieces.txt
