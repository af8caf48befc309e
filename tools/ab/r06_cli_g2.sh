#!/bin/bash
set -u
OUT=gpurun_out/r06_cli_g2.txt
: > $OUT
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only -256 > /dev/null 2>> $OUT
D=/dev/shm/kmcp_cli_keep
CLI=kmcp_amd/kmcp-search
sed "s#/dev/shm/[^/]*/#$D/#" $D/files.txt > $D/files2.txt
run() {
  local label=$1; shift
  rm -f $D/out.tsv
  local s=$(date +%s%N)
  env "${1}" $CLI ${2:-} ${3:-} -d $D/db -g -t 0.4 -s jacc --infile-list $D/files2.txt -o $D/out.tsv 2> $D/log.txt
  local e=$(date +%s%N)
  echo "== $label: $(( (e - s) / 1000000 )) ms wall; $(grep -o 'pipeline: [0-9.]* s in the GPU library' $D/log.txt); $(grep -o 'reader: [0-9.]* s parsing' $D/log.txt); $(grep -o '[0-9.]* s before the search started' $D/log.txt); $(grep -o 'elapsed time.*' $D/log.txt)" >> $OUT
}
sleep 3
for i in 1 2 3; do run "default (quick exit) #$i" X=1; done
sleep 3
for i in 1 2 3; do run "full teardown #$i" KMCP_SEARCH_FULL_TEARDOWN=1; done
sleep 3
for i in 1 2 3; do run "again #$i" X=1; done
rm -rf $D
cat $OUT
