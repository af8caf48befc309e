#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests -q -m gpu -n 4 > $OUT/c28_pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/c28_pytest.log | tail -6
cp kmcp_amd/libkmcpgpu.so scratch/libkmcpgpu_new.so
run() { local label=$1; shift; local wl=$1; shift
  env "$@" python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/c28_$label.json 2> $OUT/c28_$label.err
  python - $label <<'P'
import json,sys
d=json.load(open('gpurun_out/c28_%s.json'%sys.argv[1]))
r=d['roofline']
print(sys.argv[1],'value %.2fM k2 %.2f ms k1 %.2f achieved %.0f GB/s traffic %.1f GB recall %s'%(d['value']/1e6,r['kernel_ms'],r['kmers_kernel_ms'],r['achieved'] or 0,(r['traffic'] or 0)/1e9,d.get('planted_recall')))
P
}
for v in old new old new; do
cp scratch/libkmcpgpu_$v.so kmcp_amd/libkmcpgpu.so
run ${v}_pub gtdb_unchunked_k31 A=1
run ${v}_pub_off gtdb_unchunked_k31 KMCPG_PRUNE=0
run ${v}_c1u config1 KMCPG_FUSE=0
run ${v}_c1 config1 A=1
done
for v in old new; do
cp scratch/libkmcpgpu_$v.so kmcp_amd/libkmcpgpu.so
run ${v}_gtdb gtdb A=1
python tools/bench_shapes.py > $OUT/c28_shapes_$v.json 2> $OUT/c28_shapes_$v.err; python - $v <<'P'
import json,sys
d=json.load(open('gpurun_out/c28_shapes_%s.json'%sys.argv[1]))
for k,v in d.items(): print(sys.argv[1],k, 'k1 %.2f k2 %.2f reads/s %.0f'%(v['kmers_ms'],v['cobs_ms'],v['reads_per_s']))
P
timeout 600 python tools/bench_real_families.py /tmp/family --modes 0 --cli-reads 1000 > $OUT/c28_fam_$v.json 2> $OUT/c28_fam_$v.err
python - $v <<'P'
import json,sys
d=json.load(open('gpurun_out/c28_fam_%s.json'%sys.argv[1]))['uniform_sigs=0']
print(sys.argv[1],'fam k2',d['k2_ms'],'k1',d['k1_ms'],'achieved',d['achieved_gbps'],'batch_s',d['search_batch_s'])
P
python tools/bench_uniform.py > $OUT/c28_uniform_$v.json 2> $OUT/c28_uniform_$v.err; python - $v <<'P'
import json,sys
d=json.load(open('gpurun_out/c28_uniform_%s.json'%sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and 'k2_ms' in v: print(sys.argv[1],k,'k2',v['k2_ms'])
P
done
cp scratch/libkmcpgpu_new.so kmcp_amd/libkmcpgpu.so
