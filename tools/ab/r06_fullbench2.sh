#!/bin/bash
# the driver's command again: does the genome search keep its two-stream gain as the fifth workload of a run?
set -u
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line_new.json 2> gpurun_out/r06_bench_new.err ) 2> gpurun_out/r06_bench_time.txt
cp bench_detail.json gpurun_out/r06_bench_detail_new.json
cat gpurun_out/r06_bench_time.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_detail_new.json'))
print('gtdb', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['secondary'].items():
    if isinstance(v,dict) and 'roofline' in v:
        r=v['roofline']; print(k,'value',v['value'],'ms',round(v['ms_per_step'],3),'k1',round(r.get('kmers_kernel_ms',0),3),'k2',round(r.get('kernel_ms',0),3),'k3',round(r.get('finalize_kernels_ms',0),3),'frac',round(r.get('frac',0),3), 'streams', v['config'].get('kernel_streams'), 'h2h', v.get('value_host_to_host'), v.get('value_host_to_host_packed'))
    else: print(k, v.get('value'), v.get('wall_s'))
PY
