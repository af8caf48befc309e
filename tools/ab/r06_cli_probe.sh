#!/bin/bash
# Where does the wall clock of kmcp-search go on configs[1] (10 M reads)?  One gpurun call: keeps the leg's files, then runs variants.
set -u
OUT=gpurun_out/r06_cli_probe.txt
: > $OUT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" >> $OUT
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only ${1:-10000000} > gpurun_out/r06_cli_probe_leg.json 2>> $OUT
D=/dev/shm/kmcp_cli_keep
ls -la $D >> $OUT 2>&1
stat() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
run() {  # label, env..., -- args
  local label=$1; shift
  local s0=$(stat)
  local s=$(date +%s%N)
  env "$@" 2> $D/log.txt
  local e=$(date +%s%N)
  echo "== $label: $(( (e - s) / 1000000 )) ms wall" >> $OUT
  grep -E "pipeline:|elapsed time|before the search" $D/log.txt | sed 's/^.*\] //' >> $OUT
  echo "   cpu.stat before: $s0" >> $OUT
  echo "   cpu.stat after : $(stat)" >> $OUT
}
CLI=kmcp_amd/kmcp-search
for i in 1 2; do run "default #$i" $CLI -d $D/db $D/reads.fq -o $D/out.tsv; done
for j in 4 8 12 16 32; do run "-j $j" $CLI -d $D/db $D/reads.fq -o $D/out.tsv -j $j; done
run "parse-only" $CLI --parse-only $D/reads.fq
run "/dev/null -j 16" $CLI -d $D/db $D/reads.fq -o /dev/null -j 16
run "gpu-batch 262144" $CLI -d $D/db $D/reads.fq -o $D/out.tsv --gpu-batch 262144
run "gpu-batch 524288" $CLI -d $D/db $D/reads.fq -o $D/out.tsv --gpu-batch 524288
run "FIN_TIMING" KMCPG_FIN_TIMING=1 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
grep -c "finalize_grouped" $D/log.txt >> $OUT
grep "finalize_grouped\|piece" $D/log.txt | head -12 >> $OUT
# the library alone on this database: open + close
run "1000 reads only (start-up + teardown)" bash -c "head -n 4000 $D/reads.fq > $D/small.fq; $CLI -d $D/db $D/small.fq -o $D/small.tsv"
rm -rf $D
cat $OUT
