#!/bin/bash
# round 5, call 8: the soak again after the FPR-row cache fix (kmcpg_expand_pairs keyed its per-thread row on the handle's address),
# then the whole GPU suite and smoke() on the final tree.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
soak() {
  local name=$1 s=$2 l=$3; shift 3
  ( time env "$@" KMCP_FUZZ_SEEDS=$s KMCP_FUZZ_LONG_SEEDS=$l timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 --timeout 600 -p no:cacheprovider ) > $OUT/r5c8_$name.txt 2>&1
  echo "$name ($*): $(grep -E 'passed|failed|error' $OUT/r5c8_$name.txt | tail -1)  $(grep real $OUT/r5c8_$name.txt)"
}
soak default 6000 1500 KMCP_FUZZ_PAIRS=1
soak pack 3000 800 KMCPG_PACK=1 KMCP_FUZZ_PAIRS=1
echo "== pytest -m gpu"
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/r5c8_pytest.txt 2>&1; tail -4 $OUT/r5c8_pytest.txt
echo "== smoke"
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
