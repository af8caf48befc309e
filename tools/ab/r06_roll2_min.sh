#!/bin/bash
# round 6, last session: k1_seg_roll2 with the pair tables at LDS offset 0 and the 32-bit FracMinHash pre-test written as v_min_u32 + one compare
set -u
OUT=gpurun_out/r06_roll2_min.txt
: > $OUT
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_genome_path_two_bit_kernel_and_its_fallback" "tests/test_gpu_parity.py::test_k1_all_forms_across_k" tests/test_gpu_pack.py -q -x 2>&1 | tail -3 >> $OUT
for i in 1 2; do
timeout 600 python bench.py --workload config2_genome_search --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('config2: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'))" >> $OUT
done
timeout 600 python tools/h2h_probe.py config2_genome_search --packed --batches 32 2>/dev/null | tail -1 >> $OUT
cat $OUT
