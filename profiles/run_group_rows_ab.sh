#!/bin/bash
# Same-box A/B of K2 variants (boxes differ by +-3 %, the size of the effects): everything runs inside ONE gpurun call.
#   - KMCPG_GROUP_ROWS=8|4 on the bench workloads and on tools/bench_shapes.py (profiles/r02_group_rows.txt);
#   - builds of libkmcpgpu.so that differ in k2_cobs only (e.g. -DK2_WAVES=N for the occupancy sweep, or an older commit) are
#     compared by swapping kmcp_amd/libkmcpgpu.so between runs: put them under scratch/ as lib_<tag>.so and list the tags.
#   usage: profiles/run_group_rows_ab.sh [tag ...]
run() { timeout 300 python bench.py --workload $1 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-extras 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('   ', round(j['value']), round(j['ms_per_step'],2), round(j['roofline']['kernel_ms'],2), round(j['roofline']['achieved']))
"; }
cp kmcp_amd/libkmcpgpu.so /tmp/lib_cur.so
for rep in 1 2; do
  for gr in 8 4; do for wl in gtdb gtdb_eighth config1; do echo "$wl KMCPG_GROUP_ROWS=$gr"; KMCPG_GROUP_ROWS=$gr run $wl; done; done
  for tag in "$@"; do cp scratch/lib_$tag.so kmcp_amd/libkmcpgpu.so; echo "gtdb build=$tag"; run gtdb; done
  cp /tmp/lib_cur.so kmcp_amd/libkmcpgpu.so
done
for gr in 8 4; do echo "== shapes KMCPG_GROUP_ROWS=$gr"; KMCPG_GROUP_ROWS=$gr timeout 200 python tools/bench_shapes.py 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin)
for k,v in j.items(): print(' ', k, round(v['kmers_ms'],2), round(v['cobs_ms'],2))"; done
