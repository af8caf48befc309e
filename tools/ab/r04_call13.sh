#!/bin/bash
set -u
mkdir -p gpurun_out
cd /root/repo
for wl in config2_genome_search config4_hifi_uniform_sigs config4_hifi; do
for v in 1 2 4 8 1; do
  KMCPG_PRUNE_EVERY=$v timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras > gpurun_out/c13_${wl}_p$v.json 2> gpurun_out/c13_${wl}_p$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c13_${wl}_p$v.json'))
r=d['roofline']
print('$wl', 'every=$v', 'value %.4g'%d['value'], 'ms %.3f'%d['ms_per_step'], 'k2 %.3f'%r['kernel_ms'], 'traffic %.4g'%r['traffic'], 'ok', d.get('sanity_batch',{}).get('parity'))
PY
done; done
