#!/bin/bash
set -u
OUT=gpurun_out/r06_tail4.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_tail.py -x -q > gpurun_out/r06_tail_pytest.log 2>&1
tail -3 gpurun_out/r06_tail_pytest.log >> $OUT
S="KMCPG_TAIL_SECTORS=0 KMCPG_TAIL_SECTORS=1 KMCPG_TAIL_SECTORS=2 KMCPG_TAIL_SECTORS=3 KMCPG_TAIL_SECTORS=2,KMCPG_TAIL_MIN=256 KMCPG_TAIL_SECTORS=2,KMCPG_NT_LOADS=0"
for al in 64 128; do
  echo "== KMCPG_ROW_ALIGN=$al" >> $OUT
  KMCPG_ROW_ALIGN=$al timeout 600 python tools/ab/r06_tail_probe.py config2_genome_search $S >> $OUT 2>> gpurun_out/r06_tail2.err
done
cat $OUT
