#!/bin/bash
# round 5, call 10: k1_seg_roll2 (whole genomes on 2-bit codes) - parity, then same-box A/B on the genome search against the byte kernel
# (KMCPG_K1_FLAGS=19) and a kernel-stats pass.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
echo "== tests"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_golden.py tests/test_gpu_pack.py -m gpu -x -q --timeout 900 ) > $OUT/r5c10_pytest.txt 2>&1; tail -4 $OUT/r5c10_pytest.txt
( time KMCP_FUZZ_SEEDS=200 KMCP_FUZZ_LONG_SEEDS=1500 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 --timeout 600 -p no:cacheprovider ) > $OUT/r5c10_fuzz.txt 2>&1; grep -E "passed|failed" $OUT/r5c10_fuzz.txt | tail -1
B="python $R/bench.py --no-cpu-baseline --no-secondary"
show() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2])); r = j["roofline"]
    print("%-36s value %10.4g  step %7.3f ms  k2 %7.3f  k1 %6.3f  h2h %10.4g checksum %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["kernel_ms"], r["kmers_kernel_ms"], j.get("value_host_to_host") or 0, j["sanity_batch"]["hits_checksum"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
echo "== genome search: byte kernel (KMCPG_K1_FLAGS=19) vs 2-bit kernel (default)"
for rep in 1 2 3; do
  KMCPG_K1_FLAGS=19 timeout 600 $B --workload config2_genome_search --steps 6 --warmup 2 > $OUT/r5c10_c2_byte_$rep.json 2> $OUT/r5c10_c2_byte_$rep.err; show "config2 byte kernel rep $rep" $OUT/r5c10_c2_byte_$rep.json
  timeout 600 $B --workload config2_genome_search --steps 6 --warmup 2 > $OUT/r5c10_c2_2bit_$rep.json 2> $OUT/r5c10_c2_2bit_$rep.err; show "config2 2-bit kernel rep $rep" $OUT/r5c10_c2_2bit_$rep.json
done
echo "== kernel stats"
STATS_ONLY=1 bash profiles/run_rocprof_r04.sh r05b config2 2>&1 | tail -3
grep -E "k1_seg|k2_cobs|k_unpack" $OUT/r05b_config2_stats_kernel_stats.txt | head -8 | cut -c1-140
