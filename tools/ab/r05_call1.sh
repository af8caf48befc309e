#!/bin/bash
# round 5, call 1: (a) the N > 1 self-launch test + the compact bench line through the default command, (b) the row-sort gate of
# k2_cobs<4,16> on ONE narrow block with every read's k-mers in hash order (0), in row order (1: all units in flight
# sweep the block's rows in the same direction) and in row order rotated by a random offset per read (2: same per-unit locality,
# no phase coherence) - kernel time and TCC hit/miss counters.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q --timeout 800 -k "two_ranks_equal_one_rank or cpu_oracle" > $OUT/r5c1_pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/r5c1_pytest.txt
tail -3 $OUT/r5c1_pytest.txt
B="python $R/bench.py --workload config4_oneblock --no-cpu-baseline --no-secondary --no-extras --steps 4 --warmup 1"
for m in 0 1 2; do
  KMCPG_DEBUG_ROWSORT=$m timeout 600 $B > $OUT/r5c1_gate_m$m.json 2> $OUT/r5c1_gate_m$m.err
  python - <<PY
import json
j=json.load(open("$OUT/r5c1_gate_m$m.json")); r=j["roofline"]
print("rowsort $m: k2 %.3f ms  k1 %.3f ms  value %.0f  checksum %s recall %s" % (r["kernel_ms"], r["kmers_kernel_ms"], j["value"], j["sanity_batch"]["hits_checksum"], j.get("planted_recall")))
PY
done
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do
  d=$OUT/_prof_gate$m; rm -rf $d
  KMCPG_DEBUG_ROWSORT=$m timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $d -o gate$m -- $B --steps 2 > /dev/null 2> $OUT/r5c1_gate${m}_pmc.err
  python $R/profiles/extract_rocprof.py $d/gate${m}_results.db $OUT/r5c1_gate${m} >> $OUT/r5c1_gate${m}_pmc.err 2>&1
  rm -rf $d
  grep k2_cobs $OUT/r5c1_gate${m}_pmc.txt | awk -F'\t' '{s[$3]+=$4; n[$3]++} END {for (c in s) printf "  rowsort '$m' %s avg %.0f\n", c, s[c]/n[c]}'
done
cd $R
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/r5c1_bench.json 2> $OUT/r5c1_bench.err; echo "bench rc $? bytes $(wc -c < $OUT/r5c1_bench.json)"
cp bench_detail.json $OUT/r5c1_bench_detail.json 2>/dev/null
head -c 3000 $OUT/r5c1_bench.json
