#!/bin/bash
# round 6: fuzz soak of the final build — the rolling window-sketch kernel, the packed entry (ordinary and page-locked codes), compact results
# beside the records, the chunked kernel forced onto short queries
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
soak() {  # name, -k expression, env...
  local name=$1 sel=$2; shift 2
  ( time env "$@" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 --timeout 900 -p no:cacheprovider -k "$sel" ) > $OUT/r06_soak_$name.txt 2>&1
  echo "$name ($*): $(grep -E 'passed|failed|error' $OUT/r06_soak_$name.txt | tail -1)  $(grep real $OUT/r06_soak_$name.txt)"
}
soak roll "long_syncmer" KMCP_FUZZ_ROLL_SEEDS=${1:-1500} KMCP_FUZZ_PAIRS=1 KMCP_FUZZ_PACKED=1
soak default "random_configuration or random_long_queries" KMCP_FUZZ_SEEDS=${2:-3000} KMCP_FUZZ_LONG_SEEDS=${3:-800} KMCP_FUZZ_PAIRS=1 KMCP_FUZZ_PACKED=1
soak splitmin "random_configuration" KMCP_FUZZ_SEEDS=1000 KMCPG_SPLIT_MIN=50 KMCP_FUZZ_PAIRS=1
grep -E "^FAILED|^ERROR" $OUT/r06_soak_*.txt | head -20
