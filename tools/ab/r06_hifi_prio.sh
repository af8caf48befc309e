#!/bin/bash
# HiFi: K1 of step i+1 beside K2 of step i once more, with the k-mer kernels on a high-priority stream of the handle's own (KMCPG_K1_STREAM=1)
set -u
OUT=gpurun_out/r06_hifi_prio.txt
: > $OUT
run() {  # label, workload, env...
  local label=$1 w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w [$label]: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'), d.get('sanity_batch',{}).get('hits_checksum'))" >> $OUT
}
for W in config4_hifi_uniform_sigs config4_hifi; do
for i in 1 2; do
run "one stream" $W X=1
run "two streams" $W KMCPG_WS_SLOTS=2 KMCP_BENCH_STREAMS=2
run "two streams + high-priority k-mer stream" $W KMCPG_WS_SLOTS=2 KMCP_BENCH_STREAMS=2 KMCPG_K1_STREAM=1
run "one bench stream + high-priority k-mer stream" $W KMCPG_WS_SLOTS=2 KMCPG_K1_STREAM=1
done
done
cat $OUT
