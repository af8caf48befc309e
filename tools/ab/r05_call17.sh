#!/bin/bash
# round 5, call 17: the restart stall - waiting on a word of pinned memory (KMCP_BENCH_FLAG=1) instead of on an event, eight runs each
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-extras --workload config4_hifi_uniform_sigs --steps 12 --warmup 3"
for rep in 1 2 3 4 5 6 7 8; do
  for f in 0 1; do
    KMCP_BENCH_TRACE=1 HSA_ENABLE_INTERRUPT=$((1-f)) timeout 600 $B > $OUT/r5c17_i${f}_$rep.json 2> $OUT/r5c17_i${f}_$rep.err
    echo "hsa_interrupt=$((1-f)) rep $rep: $(python -c "import json;j=json.load(open('$OUT/r5c17_i${f}_$rep.json'));print('step %.3f ms' % j['ms_per_step'])")  waits: $(grep 'step' $OUT/r5c17_i${f}_$rep.err | tail -12 | head -4 | awk '{printf "%s ", $12}')"
  done
done
