#!/bin/bash
set -u
OUT=gpurun_out/r06_k1roll2.txt
: > $OUT
W=config4_hifi_uniform_sigs
run() {
  env "$@" python bench.py --workload $W --no-cpu-baseline --no-secondary --no-extras --steps 10 --warmup 3 > /dev/null 2>> gpurun_out/r06_k1roll2.err
  python - "$*" <<'PY' >> gpurun_out/r06_k1roll2.txt
import json, sys
j = json.load(open("bench_detail.json"))
rf = j["roofline"]
print("%-40s value %.3f M reads/s, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms" % (sys.argv[1], j["value"] / 1e6, j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"]))
PY
}
for rep in 1 2; do
run KMCPG_K1_FLAGS=35
run KMCPG_WR_WAVES=4
run KMCPG_WR_WAVES=2
run KMCPG_WR_WAVES=1
done
cat $OUT
