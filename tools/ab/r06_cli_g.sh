#!/bin/bash
set -u
OUT=gpurun_out/r06_cli_g.txt
: > $OUT
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only -256 > /dev/null 2>> $OUT
D=/dev/shm/kmcp_cli_keep
CLI=kmcp_amd/kmcp-search
for i in 1 2; do
  rm -f $D/out.tsv
  s=$(date +%s%N)
  sed "s#/dev/shm/[^/]*/#$D/#" $D/files.txt > $D/files2.txt; $CLI -d $D/db -g -t 0.4 -s jacc --infile-list $D/files2.txt -o $D/out.tsv 2> $D/log.txt
  e=$(date +%s%N)
  echo "== run $i: $(( (e - s) / 1000000 )) ms wall" >> $OUT
  grep -E "pipeline|writer loop|elapsed" $D/log.txt | sed 's/^.*\] //' >> $OUT
done
# the reader alone
s=$(date +%s%N); $CLI --parse-only -g $(head -64 $D/files2.txt) > /dev/null 2> $D/log2.txt; e=$(date +%s%N)
echo "== parse-only 64 files: $(( (e - s) / 1000000 )) ms" >> $OUT
tail -2 $D/log2.txt >> $OUT
# one file through the reader in a loop: how long does a 4-Mbp FASTA take?
python - <<PY >> $OUT
import time, subprocess
t=time.time(); subprocess.run(["$CLI","--parse-only","$D/asm00000.fasta"],capture_output=True); print("parse-only one file (process incl.): %.1f ms" % ((time.time()-t)*1e3))
PY
rm -rf $D
cat $OUT
