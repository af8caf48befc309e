#!/bin/bash
set -u
OUT=gpurun_out/r06_batch.txt
: > $OUT
for B in 256 384 512 1024; do
timeout 900 python bench.py --workload config2_genome_search --batch-reads $B --no-secondary --no-extras --steps 20 --warmup 3 --cpu-sample-reads 32 > /dev/null 2>> gpurun_out/r06_batch.err
python - "$B" <<'PY' >> gpurun_out/r06_batch.txt
import json, sys
j = json.load(open("bench_detail.json")); rf = j["roofline"]
print("batch %s: value %.4g, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms, kernel %s, parity %s" % (sys.argv[1], j["value"], j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], rf.get("kernel"), (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
done
cat $OUT
