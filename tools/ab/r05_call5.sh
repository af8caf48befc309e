#!/bin/bash
# round 5, call 5: the whole GPU suite on the build with compact results + the trusted fast path, the real-family database again,
# and the CLI A/B that call 4 could not time (no /usr/bin/time on the box).
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
echo "== pytest -m gpu"
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/r5c5_pytest.txt 2>&1; tail -6 $OUT/r5c5_pytest.txt
echo "== real-family database"
FAM=/tmp/fam; rm -rf $FAM
timeout 900 python tools/bench_real_families.py $FAM --modes 0,1 > $OUT/r5c5_real_families.json 2> $OUT/r5c5_real_families.err; echo "rc $?"
python - <<PY
import json
j = json.load(open("$OUT/r5c5_real_families.json"))
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
      t0=$(date +%s.%N); $bin -d $FAM/mode$m $FAM/reads.fq -o /tmp/o_${which}.tsv 2> /tmp/t_${which}.txt; t1=$(date +%s.%N)
      echo "mode $m rep $rep $which: $(python -c "print('%.3f s wall' % ($t1 - $t0))")  $(grep -o 'pipeline:.*' /tmp/t_${which}.txt | tail -1 | cut -c1-220)"
    done
    cmp /tmp/o_pairs.tsv /tmp/o_records.tsv && echo "  TSV identical ($(wc -l < /tmp/o_pairs.tsv) lines, $(stat -c %s /tmp/o_pairs.tsv) bytes)"
  done
done
