#!/bin/bash
set -u
OUT=gpurun_out/r06_k1roll3.txt
: > $OUT
python -m pytest tests/test_gpu_parity.py -q -x -k "window_sketch or dedup_classes or sketch_kernels" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 >> $OUT
W=${1:-config4_hifi_uniform_sigs}
run() {
  env "$@" python bench.py --workload $W --no-secondary --no-extras --steps 40 --warmup 5 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_k1roll3.err
  python - "$*" <<'PY' >> gpurun_out/r06_k1roll3.txt
import json, sys
j = json.load(open("bench_detail.json"))
rf = j["roofline"]
print("%-40s value %.3f M reads/s, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms, parity %s" % (sys.argv[1], j["value"] / 1e6, j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
}
for rep in 1 2; do
run KMCPG_K1_FLAGS=35
run KMCPG_WR_WAVES=4
run KMCPG_WR_WAVES=2
done
cat $OUT
