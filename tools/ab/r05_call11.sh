#!/bin/bash
# round 5, call 11 (final build): the whole GPU suite, the config2 profiling passes again (k1_seg_roll2 in them), the driver's command.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
echo "== pytest -m gpu"
( time timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/r5c11_pytest.txt 2>&1; tail -4 $OUT/r5c11_pytest.txt
echo "== smoke"
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
echo "== rocprof config2 (tag r05)"
bash profiles/run_rocprof_r04.sh r05 config2 2>&1 | tail -3
cd $R
echo "== the driver's command"
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/r5c11_bench.json 2> $OUT/r5c11_bench.err ) 2>&1 | tail -3; echo "bench bytes $(wc -c < $OUT/r5c11_bench.json)"
cp bench_detail.json $OUT/r5c11_bench_detail.json 2>/dev/null
cat $OUT/r5c11_bench.json
