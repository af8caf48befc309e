#!/bin/bash
out=gpurun_out/r02_fuse_exp.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --workload $WL 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['achieved'], j.get('planted_recall'), j.get('host_boundary',{}).get('value'))
    elif 'amdgpu.ids' not in l: print(l)
" >> $out; }
WL=gtdb run KMCPG_FUSE=1
WL=config1 run KMCPG_FUSE=1
WL=config1 run KMCPG_FUSE=0
WL=config1_wide run KMCPG_FUSE=1
cat $out
