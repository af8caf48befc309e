#!/bin/bash
set -u
OUT=gpurun_out/r06_slotmajor.txt
: > $OUT
run() {
  local W=$1; shift
  env "$@" python bench.py --workload $W --no-secondary --no-extras --steps 40 --warmup 5 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_slotmajor.err
  python - "$W $*" <<'PY' >> gpurun_out/r06_slotmajor.txt
import json, sys
j = json.load(open("bench_detail.json"))
rf = j["roofline"]
print("%-60s value %.4g, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms, parity %s" % (sys.argv[1], j["value"], j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
}
for rep in 1 2; do
for W in config4_hifi config4_hifi_uniform_sigs config2_genome_search; do
run $W KMCPG_SLOT_MAJOR=1
run $W KMCPG_SLOT_MAJOR=2
done
done
cat $OUT
