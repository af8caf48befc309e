#!/bin/bash
# round 6, last session: nt2_fold4 (four 2-bit codes -> one byte while a long read is staged) as v_dot4_u32_u8 instead of a 32-bit multiply + shift
set -u
OUT=gpurun_out/r06_dot4.txt
: > $OUT
timeout 1200 python -m pytest "tests/test_gpu_parity.py::test_genome_path_two_bit_kernel_and_its_fallback" "tests/test_gpu_parity.py::test_k1_all_forms_across_k" "tests/test_gpu_parity.py::test_window_sketch_kernel_forms_on_long_reads" tests/test_gpu_pack.py tests/test_gpu_fullsize_sketch.py -q -x 2>&1 | tail -3 >> $OUT
for W in config2_genome_search config4_hifi_uniform_sigs; do
for i in 1 2; do
timeout 600 python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$W: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'))" >> $OUT
done
done
cat $OUT
