#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
for al in 64 2048; do
  export KMCPG_STRIDE_ALIGN=$al
  timeout 400 python bench.py --steps 4 --warmup 1 --no-secondary --no-extras --no-cpu-baseline > gpurun_out/c21_gtdb_$al.json 2>> gpurun_out/c21.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c21_gtdb_$al.json')); rf=d['roofline']
print('gtdb align $al: k2 %.2f ms value %.4g rows %.5g recall %s'%(rf['kernel_ms'],d['value'],rf['row_bytes_per_launch'],d.get('planted_recall')))
PY
done
done
for al in 64 128 1024; do
  export KMCPG_STRIDE_ALIGN=$al
  timeout 400 python bench.py --workload config2_genome_search --steps 6 --warmup 2 --no-secondary --no-extras --cpu-sample-reads 32 > gpurun_out/c21_c2_$al.json 2>> gpurun_out/c21.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c21_c2_$al.json')); rf=d['roofline']
print('config2 align $al: k2 %.3f ms value %.4g rows %.5g parity %s'%(rf['kernel_ms'],d['value'],rf['row_bytes_per_launch'],(d.get('cpu_baseline') or {}).get('parity_on_sample')))
PY
done
