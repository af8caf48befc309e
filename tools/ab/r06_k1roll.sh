#!/bin/bash
set -u
OUT=gpurun_out/r06_k1roll.txt
: > $OUT
python -m pytest tests/test_gpu_parity.py -q -x -k "window_sketch or dedup_classes or sketch_kernels or k1_all_forms" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 >> $OUT
python -m pytest tests/test_gpu_fullsize_sketch.py tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head -20 >> $OUT
for W in config4_hifi_uniform_sigs config4_hifi; do
for f in 3 35 3 35; do
  KMCPG_K1_FLAGS=$f python bench.py --workload $W --no-secondary --no-extras --steps 10 --warmup 3 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_k1roll.err
  python - <<PY >> $OUT
import json
j = json.load(open("bench_detail.json"))
rf = j["roofline"]
print("$W KMCPG_K1_FLAGS=$f: value %.3f M reads/s, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms, parity %s" % (j["value"] / 1e6, j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
done
done
cat $OUT
