#!/bin/bash
# kmcp-search on configs[1] after the round-6 changes: early reader, quota-aware thread counts, finer writer-loop timers, quick exit
set -u
OUT=gpurun_out/r06_cli_probe2.txt
: > $OUT
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only ${1:-10000000} > gpurun_out/r06_cli_probe2_leg.json 2>> $OUT
D=/dev/shm/kmcp_cli_keep
run() {
  local label=$1; shift
  local s=$(date +%s%N)
  env "$@" 2> $D/log.txt
  local e=$(date +%s%N)
  echo "== $label: $(( (e - s) / 1000000 )) ms wall" >> $OUT
  grep -E "pipeline:|writer loop|elapsed time" $D/log.txt | sed 's/^.*\] //' >> $OUT
}
CLI=kmcp_amd/kmcp-search
for i in 1 2 3; do run "default #$i" $CLI -d $D/db $D/reads.fq -o $D/out.tsv; done
for i in 1 2; do run "/dev/null #$i" $CLI -d $D/db $D/reads.fq -o /dev/null; done
for t in 2 4; do for i in 1 2; do run "KMCPG_LOAD_THREADS=$t #$i" KMCPG_LOAD_THREADS=$t $CLI -d $D/db $D/reads.fq -o $D/out.tsv; done; done
for j in 8 12 16 24; do run "-j $j" $CLI -d $D/db $D/reads.fq -o $D/out.tsv -j $j; done
run "KMCP_READER_THREADS=12" KMCP_READER_THREADS=12 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "KMCP_READER_THREADS=4" KMCP_READER_THREADS=4 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "full teardown" KMCP_SEARCH_FULL_TEARDOWN=1 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "gpu-batch 65536" $CLI -d $D/db $D/reads.fq -o $D/out.tsv --gpu-batch 65536
run "gpu-batch 262144" $CLI -d $D/db $D/reads.fq -o $D/out.tsv --gpu-batch 262144
rm -rf $D
cat $OUT; cat gpurun_out/r06_cli_probe2_leg.json
