#!/bin/bash
set -u
mkdir -p gpurun_out
cd /root/repo
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
timeout 560 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_golden.py tests/test_gpu_groups.py tests/test_gpu_fullsize.py tests/test_gpu_config0.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | head
for wl in config4_hifi_uniform_sigs config4_hifi config2_genome_search; do
for which in prev new prev new; do
  cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
  timeout 400 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-extras > gpurun_out/c14_${wl}_$which.json 2> gpurun_out/c14_${wl}_$which.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c14_${wl}_$which.json'))
r=d['roofline']
print('$wl', '$which', 'value %.4g'%d['value'], 'ms %.3f'%d['ms_per_step'], 'k2 %.3f'%r['kernel_ms'], 'traffic %.4g'%r['traffic'], 'frac %.3f'%r['frac'])
PY
  [ $wl = gtdb ] && [ $which = new ] && break
done; done
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
