#!/bin/bash
set -u
OUT=gpurun_out/r06_tail7.txt
: > $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_fuzz.py -x -q -k "tail or config4 or long" > gpurun_out/r06_tail_pytest.log 2>&1
tail -3 gpurun_out/r06_tail_pytest.log >> $OUT
S="KMCPG_TAIL_SECTORS=0 KMCPG_TAIL_SECTORS=1 KMCPG_TAIL_SECTORS=2 KMCPG_TAIL_SECTORS=2,KMCPG_TAIL_MIN=128 KMCPG_TAIL_SECTORS=2,KMCPG_TAIL_MIN=256 KMCPG_TAIL_SECTORS=0,KMCPG_PRUNE=0"
timeout 600 python tools/ab/r06_tail_probe.py config4_hifi $S >> $OUT 2>> gpurun_out/r06_tail2.err
cat $OUT
