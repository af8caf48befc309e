#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
run() { local name=$1; shift
  timeout 400 python bench.py "$@" --steps 4 --warmup 1 --no-secondary --no-extras > gpurun_out/c24_$name.json 2>> gpurun_out/c24.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c24_$name.json')); rf=d['roofline']
print('$name: k2 %.2f ms value %.4g rows %.5g parity %s'%(rf['kernel_ms'],d['value'],rf['row_bytes_per_launch'],(d.get('cpu_baseline') or {}).get('parity_on_sample')))
PY
}
for rep in 1 2; do
for which in base hybrid; do
  cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
  run pub_$which --workload gtdb_unchunked_k31 --no-cpu-baseline
  run gtdb_$which --no-cpu-baseline
done
done
cp scratch/ab/lib_hybrid.so kmcp_amd/libkmcpgpu.so
run pub_parity --workload gtdb_unchunked_k31 --cpu-sample-reads 4096
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
