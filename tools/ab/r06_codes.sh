#!/bin/bash
# round 6, last session: whole-genome batches hashed from their 2-bit codes (k1_seg_roll2 on the packed stream) against the expansion to text
set -u
OUT=gpurun_out/r06_codes.txt
: > $OUT
timeout 1500 python -m pytest tests/test_gpu_pack.py "tests/test_gpu_parity.py::test_genome_path_two_bit_kernel_and_its_fallback" tests/test_gpu_async.py -q -x 2>&1 | tail -15 >> $OUT
for i in 1 2 3; do
for C in 1 0; do
echo "KMCPG_K1_CODES=$C" >> $OUT
KMCPG_K1_CODES=$C timeout 600 python tools/h2h_probe.py config2_genome_search --packed --batches 32 2>/dev/null | tail -1 >> $OUT
done
done
cat $OUT
