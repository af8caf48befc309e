#!/bin/bash
cd /root/repo
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
for which in prev new prev new; do
  cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
  KMCP_BENCH_TRACE=1 timeout 200 python bench.py --workload config4_hifi_uniform_sigs --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras 2>gpurun_out/c15_$which.err >gpurun_out/c15_$which.json
  python - <<PY
import json,re
d=json.load(open('gpurun_out/c15_$which.json'))
ex=[float(m.group(1)) for m in re.finditer(r'exchange ([0-9.]+) ms', open('gpurun_out/c15_$which.err').read())]
print('$which', 'ms %.3f'%d['ms_per_step'], 'k2 %.3f'%d['roofline']['kernel_ms'], 'k1 %.3f'%d['roofline']['kmers_kernel_ms'], 'exchange waits:', ' '.join('%.1f'%x for x in ex[5:]))
PY
done
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
