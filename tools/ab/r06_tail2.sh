#!/bin/bash
set -u
OUT=gpurun_out/r06_tail8.txt
: > $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_fuzz.py -x -q -k "tail or wide_rows" > gpurun_out/r06_tail_pytest.log 2>&1
tail -2 gpurun_out/r06_tail_pytest.log >> $OUT
S="KMCPG_TAIL_SECTORS=0 KMCPG_TAIL_SECTORS=2 KMCPG_TAIL_SECTORS=0 KMCPG_TAIL_SECTORS=2"
timeout 600 python tools/ab/r06_tail_probe.py config2_genome_search $S >> $OUT 2>> gpurun_out/r06_tail2.err
timeout 600 python tools/ab/r06_tail_probe.py config4_hifi_uniform_sigs $S >> $OUT 2>> gpurun_out/r06_tail2.err
cat $OUT
