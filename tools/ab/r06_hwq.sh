#!/bin/bash
# does GPU_MAX_HW_QUEUES (HIP streams -> hardware queues; default 4) matter once bench.py holds a second kernel stream beside the library's own four?
set -u
OUT=gpurun_out/r06_hwq.txt
: > $OUT
run() {  # label, workload, env...
  local label=$1 w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w [$label]: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k2', r.get('kernel_ms'), 'h2h', d.get('value_host_to_host'), d.get('value_host_to_host_packed'))" >> $OUT
}
for i in 1 2; do
run "default" config2_genome_search X=1
run "GPU_MAX_HW_QUEUES=8" config2_genome_search GPU_MAX_HW_QUEUES=8
done
run "default" config4_hifi_uniform_sigs X=1
run "GPU_MAX_HW_QUEUES=8" config4_hifi_uniform_sigs GPU_MAX_HW_QUEUES=8
run "default" config1 X=1
run "GPU_MAX_HW_QUEUES=8" config1 GPU_MAX_HW_QUEUES=8
cat $OUT
