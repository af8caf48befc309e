#!/bin/bash
# round 6, last session: sort + unique of whole-genome FracMinHash sketches (4 096 < m <= 16 384) by the distribution pass (k_dedup_bucket<16384, 2048, 1024>)
set -u
OUT=gpurun_out/r06_bigbucket.txt
: > $OUT
timeout 1200 python -m pytest "tests/test_gpu_parity.py::test_dedup_classes_at_their_boundaries" tests/test_gpu_fullsize_sketch.py tests/test_gpu_pack.py -q -x 2>&1 | tail -3 >> $OUT
KMCP_FUZZ_LONG_SEEDS=1500 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 -p no:cacheprovider -k "random_long_queries" 2>&1 | tail -1 >> $OUT
for i in 1 2; do
for B in 1 0; do
KMCPG_DEDUP_BIG_BUCKET=$B timeout 600 python bench.py --workload config2_genome_search --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('big_bucket=$B: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'))" >> $OUT
done
done
cat $OUT
