#!/bin/bash
set -u
OUT=gpurun_out/r06_cls2.txt
: > $OUT
for v in 1 0 1; do
KMCPG_CLASS_STREAMS=$v KMCP_BENCH_CLI=0 python bench.py --no-cpu-baseline > gpurun_out/r06_cls2_line.json 2> gpurun_out/r06_cls2.err
python - "$v" <<'PY' >> gpurun_out/r06_cls2.txt
import json, sys
j = json.load(open("bench_detail.json"))
s = j["secondary"]["config4_hifi_uniform_sigs"]; rf = s["roofline"]
print("full bench, KMCPG_CLASS_STREAMS=%s: uniform value %.4g, ms_per_step %.3f, k1 %.3f + k2 %.3f + k3 %.3f" % (sys.argv[1], s["value"], s["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], rf["finalize_kernels_ms"]))
s = j["secondary"]["config2_genome_search"]; rf = s["roofline"]
print("    config2 value %.4g, ms_per_step %.3f, k2 %.3f; headline %.4g" % (s["value"], s["ms_per_step"], rf["kernel_ms"], j["value"]))
PY
done
cat $OUT
