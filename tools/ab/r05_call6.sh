#!/bin/bash
# round 5, call 6: the 32-lane form / split row remainders (parity, then same-box A/B on the genome search), then the round's
# profiling passes (rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE of the bench commands) and the driver's bench command.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
echo "== tests"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_golden.py tests/test_gpu_pairs.py -m gpu -x -q --timeout 900 ) > $OUT/r5c6_pytest.txt 2>&1; tail -5 $OUT/r5c6_pytest.txt
B="python $R/bench.py --no-cpu-baseline --no-secondary"
show() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2])); r = j["roofline"]
    print("%-36s value %10.4g  step %7.3f ms  k2 %7.3f  k1 %6.3f  h2h %10.4g frac %.3f traffic %.4g checksum %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["kernel_ms"], r["kmers_kernel_ms"], j.get("value_host_to_host") or 0, r["frac"], r["traffic"], j["sanity_batch"]["hits_checksum"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
echo "== split row remainders, genome search"
for rep in 1 2; do
  for sp in 0 1; do
    KMCPG_SPLIT_TILES=$sp timeout 600 $B --workload config2_genome_search --steps 6 --warmup 2 > $OUT/r5c6_c2_sp${sp}_$rep.json 2> $OUT/r5c6_c2_sp${sp}_$rep.err; show "config2 KMCPG_SPLIT_TILES=$sp rep $rep" $OUT/r5c6_c2_sp${sp}_$rep.json
  done
done
BEST=$(python - <<PY
import json
def ms(sp): return min(json.load(open("$OUT/r5c6_c2_sp%d_%d.json" % (sp, r)))["ms_per_step"] for r in (1, 2))
try:
    print(1 if ms(1) < 0.985 * ms(0) else 0)
except Exception:
    print(0)
PY
)
echo "split tiles taken: $BEST"
export KMCPG_SPLIT_TILES=$BEST
echo "== rocprof passes (tag r05)"
bash profiles/run_rocprof_r04.sh r05 gtdb,config2,config4 2>&1 | tail -30
cd $R
echo "== the driver's command"
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/r5c6_bench.json 2> $OUT/r5c6_bench.err ) 2>&1 | tail -3; echo "bench bytes $(wc -c < $OUT/r5c6_bench.json)"
cp bench_detail.json $OUT/r5c6_bench_detail.json 2>/dev/null
cat $OUT/r5c6_bench.json
