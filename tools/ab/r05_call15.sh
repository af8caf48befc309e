#!/bin/bash
# round 5, call 15: random mid-width rows (the 32-lane form) against the oracle
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
( time KMCP_FUZZ_PAIRS=1 KMCP_FUZZ_SEEDS=0 KMCP_FUZZ_LONG_SEEDS=0 KMCP_FUZZ_WIDE_SEEDS=600 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 --timeout 900 -p no:cacheprovider -k mid_width ) > $OUT/r5c15_wide.txt 2>&1
grep -E "passed|failed|error" $OUT/r5c15_wide.txt | tail -2; grep real $OUT/r5c15_wide.txt
