#!/bin/bash
# round 5, call 13: split row remainders once more, where they might pay - single-hash short reads on 782-byte rows (512 + 256 + 64) -
# and parity of the default 32-lane rule.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-extras"
show() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2])); r = j["roofline"]
    print("%-36s value %10.4g  step %7.3f ms  k2 %7.3f  frac %.3f traffic %.4g checksum %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"], j["sanity_batch"]["hits_checksum"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
  for v in 0 2; do
    KMCPG_SPLIT_TILES=$v timeout 600 $B --workload mid_rows_782 --steps 6 --warmup 2 > $OUT/r5c13_782_${v}_${rep}.json 2> $OUT/r5c13_782_${v}_${rep}.err; show "mid_rows_782 KMCPG_SPLIT_TILES=$v rep $rep" $OUT/r5c13_782_${v}_${rep}.json
  done
done
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 -k "remainders" ) > $OUT/r5c13_pytest.txt 2>&1; tail -3 $OUT/r5c13_pytest.txt
