#!/bin/bash
cd /root/repo
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
for which in base u4 base u4; do
  cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
  timeout 200 python bench.py --workload config2_genome_search --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras 2>gpurun_out/c17_$which.err >gpurun_out/c17_$which.json
  python - <<PY
import json
d=json.load(open('gpurun_out/c17_$which.json'))
print('$which', 'value %.4g'%d['value'], 'ms %.3f'%d['ms_per_step'], 'k2 %.3f'%d['roofline']['kernel_ms'], 'k1 %.3f'%d['roofline']['kmers_kernel_ms'])
PY
done
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
