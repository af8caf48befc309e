#!/bin/bash
# round 5, call 16: the "restart stall" of VERDICT r4 weak #8 without a profiler attached: per-step host timings (KMCP_BENCH_TRACE=1) of the
# 5-ms HiFi workload, hipEventSynchronize vs busy polling, twice each.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-extras --workload config4_hifi_uniform_sigs --steps 12 --warmup 3"
for rep in 1 2; do
  for poll in 0 1; do
    KMCP_BENCH_TRACE=1 KMCP_BENCH_POLL=$poll timeout 600 $B > $OUT/r5c16_p${poll}_$rep.json 2> $OUT/r5c16_p${poll}_$rep.err
    echo "== poll=$poll rep $rep: $(python -c "import json;j=json.load(open('$OUT/r5c16_p${poll}_$rep.json'));print('step %.3f ms k2 %.3f k1 %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kmers_kernel_ms']))")"
    grep "step" $OUT/r5c16_p${poll}_$rep.err | tail -12 | awk '{print "   ", $3, $4, "enq", $9, "wait", $12, "host", $16}' | tr -d ',' | head -12
  done
done
