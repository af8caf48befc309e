#!/bin/bash
set -u
OUT=gpurun_out/r06_cli_probe4.txt
: > $OUT
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only ${1:-10000000} > /dev/null 2>> $OUT
D=/dev/shm/kmcp_cli_keep
CLI=kmcp_amd/kmcp-search
for i in 1 2 3; do
  s=$(date +%s%N)
  taskset -c 0-15,128-143 $CLI -d $D/db $D/reads.fq -o $D/out.tsv 2> $D/log.txt
  e=$(date +%s%N)
  echo "== run $i: $(( (e - s) / 1000000 )) ms wall" >> $OUT
  grep -v "^$" $D/log.txt | tr '\r' '\n' | grep -v "processed queries: [0-9]*, speed.*per minute$" | tail -25 >> $OUT
done
rm -rf $D
cat $OUT
