#!/bin/bash
# round 5, call 7: fuzz soak of the final build - default settings, every batch through the 2-bit packed upload, compact results
# beside the records, the two-workspace / two-stream machinery switched on.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
soak() {  # name, seeds, long seeds, env...
  local name=$1 s=$2 l=$3; shift 3
  ( time env "$@" KMCP_FUZZ_SEEDS=$s KMCP_FUZZ_LONG_SEEDS=$l timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 --timeout 600 -p no:cacheprovider ) > $OUT/r5c7_$name.txt 2>&1
  echo "$name ($*): $(grep -E 'passed|failed|error' $OUT/r5c7_$name.txt | tail -1)  $(grep real $OUT/r5c7_$name.txt)"
}
soak default 5000 1200 KMCP_FUZZ_PAIRS=1
soak pack 2500 600 KMCPG_PACK=1 KMCP_FUZZ_PAIRS=1
soak slots 1500 300 KMCPG_WS_SLOTS=2 KMCPG_KSTREAMS=2
soak fuse0 800 0 KMCPG_FUSE=0 KMCPG_PACK=1
