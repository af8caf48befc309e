#!/bin/bash
# does NUMA placement explain the run-to-run spread of kmcp-search's formatting time?
set -u
OUT=gpurun_out/r06_cli_probe3.txt
: > $OUT
lscpu | grep -E "Socket|NUMA|Model name|Thread|Core" >> $OUT
for c in /sys/class/drm/card*/device/numa_node; do echo "$c: $(cat $c)" >> $OUT; done
cat /sys/fs/cgroup/cpuset.cpus.effective >> $OUT 2>&1
which taskset numactl >> $OUT 2>&1
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only ${1:-10000000} > /dev/null 2>> $OUT
D=/dev/shm/kmcp_cli_keep
run() {
  local label=$1; shift
  local s=$(date +%s%N)
  "$@" 2> $D/log.txt
  local e=$(date +%s%N)
  echo "== $label: $(( (e - s) / 1000000 )) ms wall; $(grep -o "writer loop.*" $D/log.txt); $(grep -o 'pipeline: [0-9.]* s in the GPU library' $D/log.txt); $(grep -o '[0-9.]* s before the search started' $D/log.txt)" >> $OUT
}
CLI=kmcp_amd/kmcp-search
for rep in 1 2 3; do
run "no affinity" $CLI -d $D/db $D/reads.fq -o $D/out.tsv


run "NUMA off" env KMCP_SEARCH_NUMA=off $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "load threads 4" env KMCPG_LOAD_THREADS=4 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "load threads 12" env KMCPG_LOAD_THREADS=12 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "-j 16" $CLI -d $D/db $D/reads.fq -o $D/out.tsv -j 16
run "dev null" $CLI -d $D/db $D/reads.fq -o /dev/null

run "taskset 0-15,128-143" taskset -c 0-15,128-143 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
done
rm -rf $D
cat $OUT
