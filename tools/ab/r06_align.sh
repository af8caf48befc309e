#!/bin/bash
# rows on a 128-byte pitch (KMCPG_ROW_ALIGN=128) against the 64-byte one, on the workloads whose padded rows are an odd number of 64-byte halves
set -u
OUT=gpurun_out/r06_align.txt
: > $OUT
run() {
  local W=$1; shift
  env "$@" timeout 900 python bench.py --workload $W --no-secondary --no-extras --steps 20 --warmup 3 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_align.err
  python - "$W $*" <<'PY' >> gpurun_out/r06_align.txt
import json, sys
j = json.load(open("bench_detail.json"))
rf = j["roofline"]
print("%-56s value %.4g, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms, traffic %.4g, parity %s" % (sys.argv[1], j["value"], j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], rf.get("traffic") or 0, (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
}
for rep in 1 2; do
for W in mid_rows mid_rows_782 config2_genome_search; do
run $W KMCPG_ROW_ALIGN=64
run $W KMCPG_ROW_ALIGN=128
done
done
cat $OUT
