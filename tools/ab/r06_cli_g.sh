#!/bin/bash
# kmcp-search -g (256 assemblies, four batches) and the bench's genome search with the second workspace taken from the fifth batch on
set -u
OUT=gpurun_out/r06_cli_g.txt
: > $OUT
for i in 1 2; do
timeout 600 python bench.py --cli-only -256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cli -g [default]:', d.get('value'), d.get('wall_s'), d.get('value_search_phase'), d.get('parity_on_sample'))" >> $OUT
KMCPG_WS_SLOTS=1 KMCPG_KSTREAMS=1 timeout 600 python bench.py --cli-only -256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cli -g [one stream, one slot]:', d.get('value'), d.get('wall_s'), d.get('value_search_phase'), d.get('parity_on_sample'))" >> $OUT
done
timeout 600 python bench.py --workload config2_genome_search --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('config2: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k2', r.get('kernel_ms'), 'h2h', d.get('value_host_to_host'), d.get('value_host_to_host_packed'))" >> $OUT
cat $OUT
