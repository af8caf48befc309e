#!/bin/bash
# same-box A/B: FLAT vs GLOBAL row loads in k2_cobs
for rep in 1 2; do
for tag in flat global; do
  cp scratch/lib_$tag.so kmcp_amd/libkmcpgpu.so
  echo "== $tag (rep $rep)"
  timeout 100 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gtdb kernel_ms', d['roofline']['kernel_ms'])"
  timeout 100 python bench.py --workload config1 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config1 kernel_ms', d['roofline']['kernel_ms'])"
  timeout 100 python tools/bench_shapes.py 2>/dev/null | python -c "
import sys,json
d=json.load(sys.stdin)
print(' '.join('%s=%.2f' % (k.split('_vs_')[0][:14], v['cobs_ms']) for k,v in d.items()))"
done; done
