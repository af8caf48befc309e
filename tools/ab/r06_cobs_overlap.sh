#!/bin/bash
# round 6, last session: may the next batch's COBS kernel start in the previous one's ragged end?  (two kernel streams, two k-mer workspaces, no COBS chain)
set -u
OUT=gpurun_out/r06_cobs_overlap.txt
: > $OUT
run() {  # label, workload, env...
  local label=$1 w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w [$label]: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'), 'hits', d.get('hits_per_step'), d.get('sanity_batch',{}).get('hits_checksum'))" >> $OUT
}
for W in "$@"; do
for i in 1 2; do
run "one stream" $W X=1
run "two streams, chained" $W KMCPG_WS_SLOTS=2 KMCP_BENCH_STREAMS=2
run "two streams, overlapping" $W KMCPG_WS_SLOTS=2 KMCP_BENCH_STREAMS=2 KMCPG_COBS_CHAIN=0
done
done
cat $OUT
