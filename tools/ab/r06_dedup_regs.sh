#!/bin/bash
# round 6, last session: k_dedup_bucket fetches a thread's elements once (registers) instead of once per pass
set -u
OUT=gpurun_out/r06_dedup_regs.txt
: > $OUT
timeout 1200 python -m pytest "tests/test_gpu_parity.py::test_dedup_classes_at_their_boundaries" "tests/test_gpu_parity.py::test_window_sketch_kernel_forms_on_long_reads" tests/test_gpu_fullsize_sketch.py -q -x 2>&1 | tail -2 >> $OUT
KMCP_FUZZ_LONG_SEEDS=1500 KMCP_FUZZ_ROLL_SEEDS=600 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 -p no:cacheprovider -k "random_long_queries or long_syncmer" 2>&1 | tail -1 >> $OUT
for W in config4_hifi_uniform_sigs config2_genome_search; do
for i in 1 2; do
timeout 600 python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$W: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'))" >> $OUT
done
done
cat $OUT
