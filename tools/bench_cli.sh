#!/bin/bash
# End-to-end run of kmcp-search on the GPU box: FASTQ file in, TSV file out, database loaded from disk.
#   usage: tools/bench_cli.sh [reads=4000000]
set -u
N=${1:-4000000}
D=/tmp/kmcp_cli_bench
rm -rf $D
python tools/make_testdata.py $D --reads $N 2>&1 | tail -2
# the reader alone (no database, no GPU): several threads on plain four-line FASTQ, then the single-threaded general reader
for i in 1 2; do kmcp_amd/kmcp-search --parse-only $D/reads.fq 2>&1 >/dev/null | tail -n 1; done
KMCP_SERIAL_READER=1 kmcp_amd/kmcp-search --parse-only $D/reads.fq 2>&1 >/dev/null | tail -n 1 | sed 's/$/ (KMCP_SERIAL_READER=1)/'
for i in 1 2 3; do
  s=$(date +%s%N)
  kmcp_amd/kmcp-search -d $D/db $D/reads.fq -o $D/out.tsv 2> $D/log.txt
  e=$(date +%s%N)
  echo "run $i: $(( (e - s) / 1000000 )) ms wall; $(grep -o 'pipeline:.*' $D/log.txt)"
done
# paired input (the same file as both mates): two parser pools, mates zipped at read 1's batch boundaries
kmcp_amd/kmcp-search --parse-only -1 $D/reads.fq -2 $D/reads.fq 2>&1 >/dev/null | tail -n 1
for i in 1 2; do
  s=$(date +%s%N)
  kmcp_amd/kmcp-search -d $D/db -1 $D/reads.fq -2 $D/reads.fq -o $D/out_pe.tsv 2> $D/log.txt
  e=$(date +%s%N)
  echo "paired run $i: $(( (e - s) / 1000000 )) ms wall; $(grep -o 'pipeline:.*' $D/log.txt)"
done
# gzip input: the first million reads through kmcp-search's own decoder, through zlib, and end to end
head -n 4000000 $D/reads.fq | gzip -6 > $D/reads1m.fq.gz
kmcp_amd/kmcp-search --parse-only $D/reads1m.fq.gz 2>&1 >/dev/null | tail -n 1 | sed 's/$/ (gzip -6, fast_gunzip)/'
KMCP_ZLIB_GUNZIP=1 kmcp_amd/kmcp-search --parse-only $D/reads1m.fq.gz 2>&1 >/dev/null | tail -n 1 | sed 's/$/ (gzip -6, zlib)/'
s=$(date +%s%N)
kmcp_amd/kmcp-search -d $D/db $D/reads1m.fq.gz -o $D/out_gz.tsv.gz 2> $D/log.txt
e=$(date +%s%N)
echo "gz in, gz out, 1 M reads: $(( (e - s) / 1000000 )) ms wall; $(grep -o 'pipeline:.*' $D/log.txt)"
kmcp_amd/kmcp-search -d $D/db $D/reads.fq -o $D/out2.tsv --gpu-batch 100000 -q
cmp $D/out.tsv $D/out2.tsv && echo "batch size does not change the output"
tail -3 $D/out.tsv
rm -rf $D
