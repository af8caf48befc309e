#!/bin/bash
set -u
OUT=gpurun_out/r06_h2h3.txt
: > $OUT
python -m pytest tests/test_gpu_pack.py tests/test_gpu_async.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3 >> $OUT
for W in config2_genome_search config4_hifi_uniform_sigs config4_hifi config1; do
for i in 1 2; do
python tools/h2h_probe.py $W 2>/dev/null | tail -1 >> $OUT
if [ $W != config1 ]; then python tools/h2h_probe.py $W --packed 2>/dev/null | tail -1 | sed 's/$/ [kmcpg_host_alloc]/' >> $OUT; fi
done
done
cat $OUT
