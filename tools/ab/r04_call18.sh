#!/bin/bash
cd /root/repo
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
for wl in config4_hifi_uniform_sigs config4_hifi config2_genome_search; do
for which in base pipe base pipe; do
  cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
  timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-extras 2>gpurun_out/c18_$which.err >gpurun_out/c18_$which.json
  python - <<PY
import json
d=json.load(open('gpurun_out/c18_$which.json'))
print('$wl', '$which', 'value %.4g'%d['value'], 'ms %.3f'%d['ms_per_step'], 'k2 %.3f'%d['roofline']['kernel_ms'], 'k1 %.3f'%d['roofline']['kmers_kernel_ms'])
PY
done; done
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
