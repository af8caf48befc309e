#!/bin/bash
set -u
timeout 1200 python bench.py --workload config2_genome_search --no-secondary --steps 10 --warmup 2 --cpu-sample-reads 64 > gpurun_out/r06_gq_line.json 2> gpurun_out/r06_gq.err
tail -3 gpurun_out/r06_gq.err
python - <<'PY'
import json
j = json.load(open("bench_detail.json"))
print(json.dumps(j.get("whole_genome_query"), indent=1))
print("value", j["value"], "k2", j["roofline"]["kernel_ms"])
PY
