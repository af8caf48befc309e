#!/bin/bash
# round 6, last session: fuzz soak of the whole-genome kernel on packed codes — every long-query draw also goes through kmcpg_submit_packed
# (KMCP_FUZZ_PACKED=1); KMCPG_K1_CODES=2 keeps the direct form however many runs of foreign bytes a batch has (default: up to one per 4 kb)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
soak() {  # name, -k expression, env...
  local name=$1 sel=$2; shift 2
  ( time env "$@" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 --timeout 900 -p no:cacheprovider -k "$sel" ) > $OUT/r06_soakc_$name.txt 2>&1
  echo "$name ($*): $(grep -E 'passed|failed|error' $OUT/r06_soakc_$name.txt | tail -1)  $(grep real $OUT/r06_soakc_$name.txt)"
}
soak forced "random_long_queries" KMCP_FUZZ_LONG_SEEDS=${1:-600} KMCP_FUZZ_PACKED=1 KMCPG_K1_CODES=2
soak default "random_long_queries" KMCP_FUZZ_LONG_SEEDS=${2:-300} KMCP_FUZZ_PACKED=1
grep -E "^FAILED|^ERROR" $OUT/r06_soakc_*.txt | head -20
