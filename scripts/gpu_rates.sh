# Throughput of the default (auto) kernel over the sample rates / formats / channel counts of the
# BASELINE configurations and the common front-end rates -> profiles/r01_rates.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "# python bench.py --no-cpu-baseline --steps 30 --fs F --sample-size S --nchan C   (auto variant, one MI355X)"
echo "# fs_hz sample_bytes channels blocks_per_launch Msamples/s x_realtime kernel_ms HBM_write_GB/s"
for cfg in "2600000 1 16" "2600000 1 12" "2600000 1 8" "2600000 2 16" "3000000 1 16" "3000000 1 12" "2048000 1 16" "2000000 1 16" "1100000 1 16" "1000000 1 16" "4000000 1 16" "10000000 2 16" "25000000 2 16"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --steps 30 --fs $1 --sample-size $2 --nchan $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; r=j['roofline']
print($1, $2, $3, c['blocks_per_gpu'], j['value'], c['x_realtime'], r['kernel_ms'], r['achieved'])"
done
} > gpurun_out/r01_rates.txt 2>&1
cat gpurun_out/r01_rates.txt
