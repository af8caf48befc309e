#!/bin/bash
# tail mode of the long-query COBS kernel: parity tests, then same-box A/B on the long-query workloads
set -u
OUT=gpurun_out/r06_tail.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_tail.py -x -q > gpurun_out/r06_tail_pytest.log 2>&1
tail -5 gpurun_out/r06_tail_pytest.log >> $OUT
run() {
  local W=$1; shift
  env "$@" timeout 600 python bench.py --workload $W --no-secondary --no-extras --steps 40 --warmup 5 --cpu-sample-reads 64 > /dev/null 2>> gpurun_out/r06_tail.err
  python - "$W $*" <<'PY' >> gpurun_out/r06_tail.txt
import json, sys
j = json.load(open("bench_detail.json"))
rf = j["roofline"]
print("%-64s value %.4g, ms_per_step %.3f, k1 %.3f ms, k2 %.3f ms, traffic %.4g, parity %s" % (sys.argv[1], j["value"], j["ms_per_step"], rf["kmers_kernel_ms"], rf["kernel_ms"], rf.get("traffic") or 0, (j.get("cpu_baseline") or {}).get("parity_on_sample")))
PY
}
for rep in 1 2; do
for W in config2_genome_search config4_hifi_uniform_sigs config4_hifi; do
run $W KMCPG_TAIL_SECTORS=0
run $W KMCPG_TAIL_SECTORS=4
done
done
run config2_genome_search KMCPG_TAIL_SECTORS=2
run config2_genome_search KMCPG_TAIL_SECTORS=1
run config2_genome_search KMCPG_TAIL_SECTORS=4 KMCPG_TAIL_MIN=256
run config4_hifi_uniform_sigs KMCPG_TAIL_SECTORS=2
run config4_hifi_uniform_sigs KMCPG_TAIL_SECTORS=4 KMCPG_TAIL_MIN=192
cat $OUT
