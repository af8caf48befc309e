#!/bin/bash
# round 5, call 4: compact results (pairs) - tests, the real-family database (kmcpg_search_batch vs kmcpg_search_batch_pairs, the CLI
# with pairs vs the round-4 CLI with records on the same library, TSV compared byte for byte) - and one more look at K1 beside K2:
# the k-mer kernels on a high-priority stream of the handle's own (KMCPG_K1_STREAM=1).
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
echo "== tests"
( time timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_cli.py tests/test_gpu_config0.py tests/test_gpu_pack.py tests/test_gpu_real_families.py -m gpu -x -q --timeout 600 ) > $OUT/r5c4_pytest.txt 2>&1; tail -6 $OUT/r5c4_pytest.txt
echo "== real-family database"
FAM=/tmp/fam; rm -rf $FAM
timeout 900 python tools/bench_real_families.py $FAM --modes 0,1 > $OUT/r5c4_real_families.json 2> $OUT/r5c4_real_families.err; echo "rc $?"
python - <<PY
import json
j = json.load(open("$OUT/r5c4_real_families.json"))
for m in ("uniform_sigs=0", "uniform_sigs=1"):
    r = j[m]
    print(m, "kernels %.4g reads/s | search_batch records %.4g pairs %.4g | pipelined records %.4g pairs %.4g | cli rows/s %.4g (dev/null %.4g) wall %.3f s" % (
        r["reads_per_s_kernels"], r["search_batch_reads_per_s"], r["search_batch_pairs_reads_per_s"], r["pipelined_reads_per_s"], r["pipelined_pairs_reads_per_s"],
        r["cli"]["rows_per_s"], r["cli"]["to_dev_null"]["rows_per_s"], r["cli"]["wall_s"]))
PY
echo "== CLI: pairs (this build) vs records (round-4 CLI source on the same library), same reads, TSV compared"
for m in 0 1; do
  for rep in 1 2; do
    for which in pairs records; do
      bin=$R/kmcp_amd/kmcp-search; [ $which = records ] && bin=$R/scratch/ab/kmcp-search-records
      /usr/bin/time -f "%e s wall" $bin -d $FAM/mode$m $FAM/reads.fq -o /tmp/o_${which}.tsv 2> /tmp/t_${which}.txt
      echo "mode $m rep $rep $which: $(tail -1 /tmp/t_${which}.txt)  $(grep -o 'pipeline:.*' /tmp/t_${which}.txt | tail -1 | cut -c1-200)"
    done
    cmp /tmp/o_pairs.tsv /tmp/o_records.tsv && echo "  TSV identical ($(wc -l < /tmp/o_pairs.tsv) lines, $(stat -c %s /tmp/o_pairs.tsv) bytes)"
  done
done
echo "== K1 on a high-priority stream"
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-extras"
show() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2])); r = j["roofline"]
    print("%-40s value %10.4g  step %7.3f ms  k2 %7.3f  k1 %6.3f  checksum %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["kernel_ms"], r["kmers_kernel_ms"], j["sanity_batch"]["hits_checksum"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for wl in config2_genome_search config4_hifi_uniform_sigs config1; do
  st=10; [ $wl = config2_genome_search ] && st=6
  for rep in 1 2; do
    timeout 600 $B --workload $wl --steps $st --warmup 2 > $OUT/r5c4_${wl}_A$rep.json 2> $OUT/r5c4_${wl}_A$rep.err; show "$wl one stream $rep" $OUT/r5c4_${wl}_A$rep.json
    KMCP_BENCH_STREAMS=2 KMCPG_WS_SLOTS=2 KMCPG_K1_STREAM=1 timeout 600 $B --workload $wl --steps $st --warmup 2 > $OUT/r5c4_${wl}_E$rep.json 2> $OUT/r5c4_${wl}_E$rep.err; show "$wl two streams + K1 prio $rep" $OUT/r5c4_${wl}_E$rep.json
  done
done
