#!/bin/bash
# round 5, call 18: pack threads of the 2-bit upload (host-to-host rate of the genome search)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-secondary --workload config2_genome_search --steps 6 --warmup 2"
for rep in 1 2; do
  for t in 4 8 12 16; do
    KMCPG_PACK_THREADS=$t timeout 600 $B > $OUT/r5c18_t${t}_$rep.json 2> $OUT/r5c18_t${t}_$rep.err
    python - <<PY
import json
j=json.load(open("$OUT/r5c18_t${t}_$rep.json")); d=json.load(open("$R/bench_detail.json"))
hb=d.get("host_boundary") or {}
print("pack threads $t rep $rep: value %.4g  h2h %.4g  single batch %.4g genomes/s" % (j["value"], j.get("value_host_to_host") or 0, hb.get("single_batch_reads_per_s") or 0))
PY
  done
done
