#!/bin/bash
set -u
OUT=gpurun_out/r06_k1pre.txt
: > $OUT
export PYTHONPATH=$PWD
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_pack.py -x -q -p no:cacheprovider ) > gpurun_out/r06_k1pre_pytest.log 2>&1
tail -3 gpurun_out/r06_k1pre_pytest.log >> $OUT
( KMCP_FUZZ_LONG_SEEDS=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 12 --timeout 900 -p no:cacheprovider -k "random_long_queries" ) > gpurun_out/r06_k1pre_fuzz.log 2>&1
tail -2 gpurun_out/r06_k1pre_fuzz.log >> $OUT
for rep in 1 2; do
timeout 600 python bench.py --workload config2_genome_search --no-secondary --no-extras --steps 40 --warmup 5 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_k1pre.err
python - <<'PY' >> gpurun_out/r06_k1pre.txt
import json
j = json.load(open("bench_detail.json")); rf = j["roofline"]
print("config2 value %.4g, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms, parity %s" % (j["value"], j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
done
cat $OUT
