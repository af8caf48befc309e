#!/bin/bash
# tail mode + 128-byte pitch: the new tests, the wide-row fuzz widened, then the whole GPU suite
set -u
OUT=gpurun_out
export PYTHONPATH=$PWD
( time timeout 1200 python -m pytest tests/test_gpu_tail.py tests/test_gpu_fuzz.py -m gpu -q -x -k "tail or wide_rows" -p no:cacheprovider ) > $OUT/r06_tail_tests.txt 2>&1
tail -6 $OUT/r06_tail_tests.txt
( time KMCP_FUZZ_TAIL_SEEDS=120 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 12 --timeout 900 -p no:cacheprovider -k "wide_rows" ) > $OUT/r06_tail_fuzz.txt 2>&1
tail -6 $OUT/r06_tail_fuzz.txt
( time timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $OUT/r06_pytest_gpu_6.log 2>&1
tail -6 $OUT/r06_pytest_gpu_6.log
