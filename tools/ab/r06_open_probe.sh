#!/bin/bash
# start-up of kmcp-search on configs[1]: HIP runtime floor, kmcpg_open phases, load-thread sweep
set -u
OUT=gpurun_out/r06_open_probe.txt
: > $OUT
for i in 1 2 3; do echo "-- ubench_init run $i" >> $OUT; scratch/ubench_init >> $OUT 2>&1; done
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only 1000000 > /dev/null 2>> $OUT
D=/dev/shm/kmcp_cli_keep
head -n 4000 $D/reads.fq > $D/small.fq
CLI=kmcp_amd/kmcp-search
for t in "" 2 4 8 16; do
  for i in 1 2; do
    echo "-- KMCPG_LOAD_THREADS=${t:-default} run $i" >> $OUT
    s=$(date +%s%N)
    if [ -z "$t" ]; then KMCPG_OPEN_TIMING=1 $CLI -d $D/db $D/small.fq -o $D/small.tsv 2> $D/log.txt
    else KMCPG_LOAD_THREADS=$t KMCPG_OPEN_TIMING=1 $CLI -d $D/db $D/small.fq -o $D/small.tsv 2> $D/log.txt; fi
    e=$(date +%s%N)
    grep -E "kmcpg_open:|before the search|elapsed time" $D/log.txt | sed 's/^.*\] //' >> $OUT
    echo "   wall $(( (e - s) / 1000000 )) ms" >> $OUT
  done
done
rm -rf $D
cat $OUT
