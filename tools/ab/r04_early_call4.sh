#!/bin/bash
set -u
mkdir -p gpurun_out
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_finalize_device.py tests/test_gpu_parity.py tests/test_gpu_real_families.py tests/test_gpu_build.py tests/test_gpu_cli.py -x -q 2>&1 | tail -15 ) > gpurun_out/c4_pytest.txt 2>&1
tail -5 gpurun_out/c4_pytest.txt
( timeout 900 python tools/bench_real_families.py /tmp/famdb --cli-reads 100000 > gpurun_out/c4_real_families.json 2> gpurun_out/c4_real_families.err ); grep -c uniform gpurun_out/c4_real_families.json
PUB="python bench.py --workload gtdb_unchunked_k31 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-extras"
for gr in 4 8 4 8; do
  KMCPG_GROUP_ROWS=$gr timeout 300 $PUB > gpurun_out/c4_pub_gr${gr}.json 2>> gpurun_out/c4_pub.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c4_pub_gr${gr}.json')); rf=d['roofline']
print('pub GR=${gr}: k2 %.2f ms traffic %.4g value %.4g'%(rf['kernel_ms'],rf['traffic'],d['value']))
PY
done
for gr in 4 8; do
  KMCP_BENCH_RANDOM_FRAC=1 KMCPG_GROUP_ROWS=$gr timeout 300 $PUB > gpurun_out/c4_pub_random_gr${gr}.json 2>> gpurun_out/c4_pub.err
  python - <<PY
import json
d=json.load(open('gpurun_out/c4_pub_random_gr${gr}.json')); rf=d['roofline']
print('pub all-random GR=${gr}: k2 %.2f ms traffic %.4g value %.4g'%(rf['kernel_ms'],rf['traffic'],d['value']))
PY
done
( time timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err ) 2>&1 | tail -3
