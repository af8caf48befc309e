#!/bin/bash
# round 5, call 2: (1) the row-sort gate again with an index that is not saturated (3 % of the reads planted), (2) the whole GPU
# suite on the rebuilt library, (3) v_bitop3 adders on the LPR = 8 / 8-plane form: same-box A/B on the published configuration,
# (4) a HIP + kernel trace of the first steps after an idle period (the 2-3 ms restart stall of VERDICT r4 weak #8).
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-extras"
echo "== gate (KMCP_BENCH_RANDOM_FRAC=0.97: 12 planted reads per column)"
for m in 0 1 2 0 1; do
  KMCP_BENCH_RANDOM_FRAC=0.97 KMCPG_DEBUG_ROWSORT=$m timeout 600 $B --workload config4_oneblock --steps 6 --warmup 2 > $OUT/r5c2_gate_m$m.json 2> $OUT/r5c2_gate_m$m.err
  python - <<PY
import json
try:
    j=json.load(open("$OUT/r5c2_gate_m$m.json")); r=j["roofline"]
    print("rowsort $m: k2 %.3f ms  k1 %.3f ms  traffic %.4g B  hits/step %.0f checksum %s recall %s" % (r["kernel_ms"], r["kmers_kernel_ms"], r["traffic"], j["hits_per_step"], j["sanity_batch"]["hits_checksum"], j.get("planted_recall")))
except Exception as e:
    print("rowsort $m failed", e)
PY
done
echo "== csa3 A/B on gtdb_unchunked_k31"
cp kmcp_amd/libkmcpgpu.so scratch/ab/lib_keep.so
for which in base csa3 base csa3; do
  cp scratch/ab/lib_$which.so kmcp_amd/libkmcpgpu.so
  timeout 600 $B --workload gtdb_unchunked_k31 --steps 10 --warmup 3 > $OUT/r5c2_pub_$which.json 2> $OUT/r5c2_pub_$which.err
  python - <<PY
import json
try:
    j=json.load(open("$OUT/r5c2_pub_$which.json")); r=j["roofline"]
    print("pub $which: k2 %.3f ms value %.4g frac %.3f checksum %s" % (r["kernel_ms"], j["value"], r["frac"], j["sanity_batch"]["hits_checksum"]))
except Exception as e:
    print("pub $which failed", e)
PY
done
cp scratch/ab/lib_keep.so kmcp_amd/libkmcpgpu.so
echo "== trace of the restart stall"
( cd /tmp && export TMPDIR=/tmp && d=$OUT/_prof_trace && rm -rf $d && KMCP_BENCH_TRACE=1 timeout 600 rocprofv3 --hip-trace --kernel-trace -d $d -o trace -- $B --workload config4_hifi_uniform_sigs --steps 12 --warmup 3 > $OUT/r5c2_trace.json 2> $OUT/r5c2_trace.err; python $R/profiles/extract_timeline.py $d/trace_results.db $OUT/r5c2_timeline.txt 8000 >> $OUT/r5c2_trace.err 2>&1; ls -la $d | head; rm -rf $d )
grep "step" $OUT/r5c2_trace.err | head -20
echo "== pytest -m gpu"
( time timeout 1200 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/r5c2_pytest.txt 2>&1; tail -8 $OUT/r5c2_pytest.txt
