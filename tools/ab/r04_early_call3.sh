#!/bin/bash
set -u
mkdir -p gpurun_out
cd /root/repo
( time timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/c3_pytest_all.txt 2>&1
tail -8 gpurun_out/c3_pytest_all.txt
( timeout 900 python tools/bench_real_families.py /tmp/famdb --cli-reads 100000 > gpurun_out/c3_real_families.json 2> gpurun_out/c3_real_families.err ); tail -3 gpurun_out/c3_real_families.err | cut -c1-400
# narrow rows with distinct NumSigs: rows between pruning tests 8 vs 4, requests per launch (live counter)
for gr in 8 4; do
  KMCPG_FUSE=0 KMCPG_GROUP_ROWS=$gr timeout 300 python bench.py --workload config1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c3_narrow_gr$gr.json 2> gpurun_out/c3_narrow_gr$gr.err
done
( time timeout 600 python bench.py --steps 5 --warmup 2 --workload config1 --no-secondary > gpurun_out/c3_bench_config1.json 2> gpurun_out/c3_bench_config1.err ) 2>&1 | tail -3
bash profiles/run_rocprof_r04.sh r04 pubsq 2>&1 | tail -12
