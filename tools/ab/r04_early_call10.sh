#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
for rep in 1 2 3; do
  for which in c5 new newp0; do
    lib=$which; [ $which = newp0 ] && lib=new
    cp scratch/ab/lib_$lib.so kmcp_amd/libkmcpgpu.so
    if [ $which = newp0 ]; then export KMCPG_PIECES=0; else unset KMCPG_PIECES; fi
    timeout 600 python bench.py --workload config1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c10_config1_${which}_$rep.json 2> gpurun_out/c10_err.txt
    python - <<PY
import json
d=json.load(open('gpurun_out/c10_config1_${which}_$rep.json')); hb=d['host_boundary']
print('config1 ${which} rep $rep: value %.4g h2h %.4g single %.4g dev_only %.4g'%(d['value'], hb['value'], hb['single_batch_reads_per_s'], d['device_only']['value']))
PY
  done
done
unset KMCPG_PIECES
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
