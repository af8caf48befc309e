#!/bin/bash
set -u
OUT=gpurun_out/r06_pair.txt
: > $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_fuzz.py -x -q -k "tail or config4 or wide_rows or mid_width or long" > gpurun_out/r06_pair_pytest.log 2>&1
tail -2 gpurun_out/r06_pair_pytest.log >> $OUT
run() {
  local W=$1; shift
  env "$@" timeout 600 python bench.py --workload $W --no-secondary --no-extras --steps 60 --warmup 5 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_pair.err
  python - "$W $*" <<'PY' >> gpurun_out/r06_pair.txt
import json, sys
j = json.load(open("bench_detail.json")); rf = j["roofline"]
print("%-44s value %.4g, ms_per_step %.3f, k1 %.3f + k2 %.3f + k3 %.3f ms, parity %s" % (sys.argv[1], j["value"], j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], rf["finalize_kernels_ms"], (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
}
for rep in 1 2; do
run config4_hifi_uniform_sigs KMCPG_PAIR=0
run config4_hifi_uniform_sigs KMCPG_PAIR=1
done
for v in 1 0; do
KMCPG_PAIR=$v KMCP_BENCH_CLI=0 python bench.py --no-cpu-baseline > gpurun_out/r06_pair_line.json 2>> gpurun_out/r06_pair.err
python - "$v" <<'PY' >> gpurun_out/r06_pair.txt
import json, sys
j = json.load(open("bench_detail.json"))
s = j["secondary"]["config4_hifi_uniform_sigs"]; rf = s["roofline"]
print("full bench, KMCPG_PAIR=%s: uniform value %.4g, ms_per_step %.3f, k1 %.3f + k2 %.3f + k3 %.3f" % (sys.argv[1], s["value"], s["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], rf["finalize_kernels_ms"]))
PY
done
cat $OUT
