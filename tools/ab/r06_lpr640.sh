#!/bin/bash
# rows of 513-640 bytes — one 640-byte tile on the 64-lane form (40 lanes busy) vs 512 (32-lane form) + 128 (8-lane form)
set -u
OUT=gpurun_out/r06_lpr640_${1:-mid_rows_586}.txt
: > $OUT
W=${1:-mid_rows_586}
B="python bench.py --workload $W --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
for rep in 1; do
for v in 0 2; do
  KMCPG_SPLIT_TILES=$v $B > /dev/null 2>> gpurun_out/r06_lpr640.err
  python - <<PY >> $OUT
import json
j = json.load(open("bench_detail.json"))
rf = j["roofline"]
print("KMCPG_SPLIT_TILES=$v rep $rep: value %.3f M reads/s, ms_per_step %.2f, k2 kernel %.2f ms, frac %.3f (requested %.1f GB), algorithmic_over_peak %.3f, parity-free run" % (
    j["value"] / 1e6, j["ms_per_step"], rf["kernel_ms"], rf["frac"], rf["traffic"] / 1e9, rf["algorithmic_over_peak"]))
PY
done
done
cat $OUT
