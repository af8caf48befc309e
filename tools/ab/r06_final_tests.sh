#!/bin/bash
# the whole -m gpu suite + a fuzz soak of the final build
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/r06_pytest_gpu_final.log 2>&1
tail -4 $OUT/r06_pytest_gpu_final.log
soak() {  # name, -k expression, env...
  local name=$1 sel=$2; shift 2
  ( time env "$@" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 --timeout 900 -p no:cacheprovider -k "$sel" ) > $OUT/r06_soak2_$name.txt 2>&1
  echo "$name ($*): $(grep -E 'passed|failed|error' $OUT/r06_soak2_$name.txt | tail -1)  $(grep real $OUT/r06_soak2_$name.txt)"
}
soak tail "wide_rows" KMCP_FUZZ_TAIL_SEEDS=1200 KMCP_FUZZ_PAIRS=1
soak default "random_configuration or random_long_queries or mid_width" KMCP_FUZZ_SEEDS=3000 KMCP_FUZZ_LONG_SEEDS=1200 KMCP_FUZZ_WIDE_SEEDS=200 KMCP_FUZZ_PAIRS=1 KMCP_FUZZ_PACKED=1
soak roll "long_syncmer" KMCP_FUZZ_ROLL_SEEDS=800 KMCP_FUZZ_PAIRS=1 KMCP_FUZZ_PACKED=1
grep -E "^FAILED|^ERROR" $OUT/r06_soak2_*.txt | head -20
