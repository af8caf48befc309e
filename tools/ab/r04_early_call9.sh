#!/bin/bash
# same-box A/B: library before the pieces / eager-stream work (lib_c5) vs now (lib_new), alternating; host boundary numbers of config1 and the family db
cd /root/repo
mkdir -p gpurun_out
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
for rep in 1 2 3; do
  for which in c5 new; do
    cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
    timeout 600 python bench.py --workload config1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c9_config1_${which}_$rep.json 2> gpurun_out/c9_err.txt
    python - <<PY
import json
d=json.load(open('gpurun_out/c9_config1_${which}_$rep.json')); hb=d['host_boundary']
print('config1 ${which} rep $rep: value %.4g h2h %.4g single %.4g dev_only %.4g'%(d['value'], hb['value'], hb['single_batch_reads_per_s'], d['device_only']['value']))
PY
    timeout 600 python bench.py --workload config4_hifi_uniform_sigs --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c9_hifi_${which}_$rep.json 2>> gpurun_out/c9_err.txt
    python - <<PY
import json
d=json.load(open('gpurun_out/c9_hifi_${which}_$rep.json')); hb=d['host_boundary']
print('hifi_uniform ${which} rep $rep: value %.4g h2h %.4g single %.4g dev_only %.4g'%(d['value'], hb['value'], hb['single_batch_reads_per_s'], d['device_only']['value']))
PY
  done
done
for which in c5 new; do
  cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
  timeout 600 python tools/bench_real_families.py /tmp/famdb --cli-reads 1000 --modes 0 > gpurun_out/c9_fam_$which.json 2>> gpurun_out/c9_err.txt
  python - <<PY
import json
d=json.load(open('gpurun_out/c9_fam_$which.json'))
for k,v in d.items():
    if k.startswith('uniform'): print('family $which', k, 'single', [round(x,4) for x in v['search_batch_s_all']], 'pipelined %.4g'%v['pipelined_reads_per_s'])
PY
done
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
