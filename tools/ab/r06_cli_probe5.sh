#!/bin/bash
set -u
OUT=gpurun_out/r06_cli_probe5.txt
: > $OUT
python -m pytest tests/test_gpu_cli.py tests/test_gpu_config0.py tests/test_gpu_paged.py -q -x 2>&1 | tail -5 >> $OUT
KMCP_BENCH_KEEP=/dev/shm/kmcp_cli_keep python bench.py --cli-only ${1:-10000000} > gpurun_out/r06_cli_probe5_leg.json 2>> $OUT
D=/dev/shm/kmcp_cli_keep
CLI=kmcp_amd/kmcp-search
run() {
  local label=$1; shift
  rm -f $D/out.tsv
  local s=$(date +%s%N)
  "$@" 2> $D/log.txt
  local e=$(date +%s%N)
  echo "== $label: $(( (e - s) / 1000000 )) ms wall; $(grep -o "pipeline.*" $D/log.txt); $(grep -o "writer loop.*" $D/log.txt); $(grep -o "elapsed time.*" $D/log.txt)" >> $OUT
}
for i in 1 2 3; do
run "default" $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "taskset 0-15,128-143" taskset -c 0-15,128-143 $CLI -d $D/db $D/reads.fq -o $D/out.tsv
run "/dev/null" $CLI -d $D/db $D/reads.fq -o /dev/null
run "gpu-batch 262144" $CLI -d $D/db $D/reads.fq -o $D/out.tsv --gpu-batch 262144
done
rm -rf $D
cat $OUT; cat gpurun_out/r06_cli_probe5_leg.json
