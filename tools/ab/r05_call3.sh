#!/bin/bash
# round 5, call 3: K1 beside K2 (two k-mer workspaces, two kernel streams), the 2-bit packed upload, busy-polling instead of
# hipEventSynchronize and GPU_MAX_HW_QUEUES - same box, bench.py with its extras (value_host_to_host = kmcpg_submit / kmcpg_wait).
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
echo "== tests on the new paths"
( time timeout 900 python -m pytest tests/test_gpu_pack.py tests/test_gpu_parity.py tests/test_gpu_async.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_finalize_device.py tests/test_gpu_paged.py tests/test_gpu_real_families.py -m gpu -x -q --timeout 600 ) > $OUT/r5c3_pytest.txt 2>&1; tail -6 $OUT/r5c3_pytest.txt
B="python $R/bench.py --no-cpu-baseline --no-secondary"
show() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2])); r = j["roofline"]
    print("%-34s value %10.4g  step %7.3f ms  k2 %7.3f  k1 %6.3f  h2h %10.4g  frac %.3f  checksum %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["kernel_ms"], r["kmers_kernel_ms"], j.get("value_host_to_host") or 0, r["frac"], j["sanity_batch"]["hits_checksum"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for wl in config2_genome_search config4_hifi_uniform_sigs config4_hifi; do
  st=10; [ $wl = config2_genome_search ] && st=6
  for rep in 1 2; do
    KMCP_BENCH_STREAMS=1 KMCPG_KSTREAMS=1 KMCPG_PACK=0 KMCPG_WS_SLOTS=1 timeout 600 $B --workload $wl --steps $st --warmup 2 > $OUT/r5c3_${wl}_A$rep.json 2> $OUT/r5c3_${wl}_A$rep.err; show "$wl A(r04 behaviour) $rep" $OUT/r5c3_${wl}_A$rep.json
    timeout 600 $B --workload $wl --steps $st --warmup 2 > $OUT/r5c3_${wl}_B$rep.json 2> $OUT/r5c3_${wl}_B$rep.err; show "$wl B(defaults) $rep" $OUT/r5c3_${wl}_B$rep.json
  done
  KMCPG_PACK=0 timeout 600 $B --workload $wl --steps $st --warmup 2 > $OUT/r5c3_${wl}_Bnopack.json 2> $OUT/r5c3_${wl}_Bnopack.err; show "$wl B, KMCPG_PACK=0" $OUT/r5c3_${wl}_Bnopack.json
  GPU_MAX_HW_QUEUES=8 timeout 600 $B --workload $wl --steps $st --warmup 2 > $OUT/r5c3_${wl}_C.json 2> $OUT/r5c3_${wl}_C.err; show "$wl C(B + 8 hw queues)" $OUT/r5c3_${wl}_C.json
  KMCP_BENCH_POLL=1 timeout 600 $B --workload $wl --steps $st --warmup 2 > $OUT/r5c3_${wl}_D.json 2> $OUT/r5c3_${wl}_D.err; show "$wl D(B + polling)" $OUT/r5c3_${wl}_D.json
done
for wl in config1 gtdb_unchunked_k31; do
  KMCP_BENCH_STREAMS=1 KMCPG_KSTREAMS=1 KMCPG_WS_SLOTS=1 timeout 600 $B --workload $wl --steps 10 --warmup 2 > $OUT/r5c3_${wl}_A.json 2> $OUT/r5c3_${wl}_A.err; show "$wl A" $OUT/r5c3_${wl}_A.json
  timeout 600 $B --workload $wl --steps 10 --warmup 2 > $OUT/r5c3_${wl}_B.json 2> $OUT/r5c3_${wl}_B.err; show "$wl B" $OUT/r5c3_${wl}_B.json
done
cat $OUT/r5c3_*.err | grep -v "^$" | grep -iv "warn\|amdgpu.ids" | sort | uniq -c | sort -rn | head -10
