#!/bin/bash
# round 5, call 14: the whole GPU suite + smoke + the driver's bench command on the final tree (32-lane rule on by default).
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$R
( time timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/r5c14_pytest.txt 2>&1; tail -4 $OUT/r5c14_pytest.txt
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/r5c14_bench.json 2> $OUT/r5c14_bench.err ) 2>&1 | tail -3; echo "bench bytes $(wc -c < $OUT/r5c14_bench.json)"
cp bench_detail.json $OUT/r5c14_bench_detail.json 2>/dev/null
python - <<PY
import json
j=json.load(open("$OUT/r5c14_bench.json"))
print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["cpu_baseline"]["value"], j["cpu_baseline"]["parity_on_sample"])
for k,v in j["secondary"].items(): print(k, v["value"], v["ms_per_step"], v.get("frac"), v.get("parity_on_sample"))
PY
