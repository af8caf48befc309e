#!/bin/bash
set -u
OUT=gpurun_out/r06_cls.txt
: > $OUT
export PYTHONPATH=$PWD
true
true
run() {
  local W=$1; shift
  env "$@" timeout 600 python bench.py --workload $W --no-secondary --no-extras --steps 60 --warmup 5 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_cls.err
  python - "$W $*" <<'PY' >> gpurun_out/r06_cls.txt
import json, sys
j = json.load(open("bench_detail.json")); rf = j["roofline"]
print("%-52s value %.4g, ms_per_step %.3f, k1 %.3f + k2 %.3f + k3 %.3f ms, parity %s" % (sys.argv[1], j["value"], j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], rf["finalize_kernels_ms"], (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
}
for rep in 1 2; do
run config4_hifi_uniform_sigs KMCPG_CLASS_STREAMS=1 KMCPG_CLASS_ORDER=1
run config4_hifi_uniform_sigs KMCPG_CLASS_STREAMS=1
done
cat $OUT
