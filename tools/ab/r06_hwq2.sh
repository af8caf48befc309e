#!/bin/bash
# GPU_MAX_HW_QUEUES=8 (bench.py's new default) against the runtime's 4 on the headline and on the published configuration, same box, alternating
set -u
OUT=gpurun_out/r06_hwq2.txt
: > $OUT
run() {  # label, args, env...
  local label=$1 w=$2; shift 2
  env "$@" timeout 900 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w [$label]: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k2', r.get('kernel_ms'), 'frac', r.get('frac'))" >> $OUT
}
for i in 1 2; do
run "GPU_MAX_HW_QUEUES=4" gtdb GPU_MAX_HW_QUEUES=4
run "GPU_MAX_HW_QUEUES=8" gtdb GPU_MAX_HW_QUEUES=8
run "GPU_MAX_HW_QUEUES=4" gtdb_unchunked_k31 GPU_MAX_HW_QUEUES=4
run "GPU_MAX_HW_QUEUES=8" gtdb_unchunked_k31 GPU_MAX_HW_QUEUES=8
done
cat $OUT
