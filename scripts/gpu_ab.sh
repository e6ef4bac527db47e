# A/B of two builds of the library in one session: gpsiq/libgpsiq_head.so vs gpsiq/libgpsiq_new.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; G=multi-sdr-gps-sim_amd/gpsiq
run() { python bench.py --no-cpu-baseline --steps 30 $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', 'value', j['value'], 'kernel_ms', j['roofline']['kernel_ms'])"; }
for rep in 1 2 3; do
  for w in head new; do
    cp $G/libgpsiq_$w.so $G/libgpsiq.so
    run $w ""
    run $w "--fs 10000000 --sample-size 2"
  done
done > gpurun_out/ab.log 2>&1
cp $G/libgpsiq_new.so $G/libgpsiq.so
cat gpurun_out/ab.log
