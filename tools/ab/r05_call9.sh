#!/bin/bash
# round 5, call 9: stress of the asynchronous boundary (several host threads on one handle; single GPU, in-process multi-device with host
# merge and with the RCCL exchange - one gather slot per batch in flight since this round -, paged) on the final build, in the default
# setting and with the packed upload / two workspaces / two kernel streams switched on.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
( time timeout 600 python tools/stress_async.py 30 6 ) > $OUT/r5c9_stress_default.txt 2>&1; tail -8 $OUT/r5c9_stress_default.txt
( time KMCPG_PACK=1 KMCPG_WS_SLOTS=2 KMCPG_KSTREAMS=2 timeout 600 python tools/stress_async.py 30 6 ) > $OUT/r5c9_stress_knobs.txt 2>&1; tail -8 $OUT/r5c9_stress_knobs.txt
