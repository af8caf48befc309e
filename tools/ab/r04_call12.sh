#!/bin/bash
set -u
mkdir -p gpurun_out
cd /root/repo
export KMCPG_SPLIT_MIN=2048
for wl in config4_hifi_uniform_sigs config4_hifi; do
for v in 0 1 0 1; do
  KMCPG_PLANES12=$v timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras > gpurun_out/c12_${wl}_p$v.json 2> gpurun_out/c12_${wl}_p$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c12_${wl}_p$v.json'))
print('$wl', 'planes12=$v', 'value %.4g'%d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline'].get('kernel_ms', d['roofline'].get('achieved')))
PY
done; done
