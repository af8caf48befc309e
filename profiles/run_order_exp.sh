#!/bin/bash
# unit-order / load-policy experiment for K2 (round 2)
out=gpurun_out/r02_order_exp.txt
: > $out
for wl in gtdb config1; do
for sm in 0 1; do for nt in 1 0; do
  echo "== $wl slot_major=$sm nt=$nt" >> $out
  KMCPG_SLOT_MAJOR=$sm KMCPG_NT_LOADS=$nt timeout 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['achieved'], j.get('planted_recall'), j.get('host_boundary',{}).get('value'))
    else: print(l)
" >> $out
done; done; done
cat $out
