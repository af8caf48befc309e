#!/bin/bash
# round 6, last session: two kernel streams + two workspace slots by default for batches of whole genomes (library lanes and bench.py alike)
set -u
OUT=gpurun_out/r06_auto_streams.txt
: > $OUT
timeout 1500 python -m pytest tests/test_gpu_pack.py tests/test_gpu_fullsize_sketch.py tests/test_gpu_cli.py tests/test_gpu_async.py "tests/test_gpu_parity.py::test_genome_path_two_bit_kernel_and_its_fallback" -q -x 2>&1 | tail -3 >> $OUT
KMCP_FUZZ_LONG_SEEDS=1500 KMCP_FUZZ_PACKED=1 KMCP_FUZZ_PAIRS=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 14 -p no:cacheprovider -k "random_long_queries" 2>&1 | tail -1 >> $OUT
run() {  # label, workload, env...
  local label=$1 w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w [$label]: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'), 'frac', r.get('frac'), 'streams', d['config'].get('kernel_streams'), 'h2h', d.get('value_host_to_host'), d.get('value_host_to_host_packed'), 'gq', (d.get('whole_genome_query') or {}).get('ms_per_query'), d.get('sanity_batch',{}).get('hits_checksum'))" >> $OUT
}
for i in 1 2; do
run "default" config2_genome_search X=1
run "one stream, one slot" config2_genome_search KMCP_BENCH_STREAMS=1 KMCPG_WS_SLOTS=1 KMCPG_KSTREAMS=1
done
run "default" config4_hifi_uniform_sigs X=1
run "default" config1 X=1
cat $OUT
