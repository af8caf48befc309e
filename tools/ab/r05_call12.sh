#!/bin/bash
# round 5, call 12: rows of 257..512 bytes on the 32-lane form of k2_cobs (two units per wave) vs the 64-lane form with half of its lanes
# idle - a 100k-chunk database as `kmcp index -j 32` cuts it (391-byte rows) - and parity on such widths.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-extras"
show() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2])); r = j["roofline"]
    print("%-28s value %10.4g  step %7.3f ms  k2 %7.3f  frac %.3f traffic %.4g checksum %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"], j["sanity_batch"]["hits_checksum"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
  for v in 0 1; do
    KMCPG_LPR32=$v timeout 600 $B --workload mid_rows --steps 8 --warmup 2 > $OUT/r5c12_mid_${v}_${rep}.json 2> $OUT/r5c12_mid_${v}_${rep}.err; show "mid_rows KMCPG_LPR32=$v rep $rep" $OUT/r5c12_mid_${v}_${rep}.json
  done
done
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 -k "remainders or medium_rows or wide_rows or mixed_block" ) > $OUT/r5c12_pytest.txt 2>&1; tail -3 $OUT/r5c12_pytest.txt
