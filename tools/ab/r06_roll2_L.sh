#!/bin/bash
# round 6, last session: k1_seg_roll2 variants built from the same tree (scratch/ab/lib_<variant>.so: -DKMCPG_R2_L=..., -DKMCPG_R2_BFE=...)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_roll2_L.txt
: > $OUT
cp $R/kmcp_amd/libkmcpgpu.so /tmp/lib_orig.so
for rep in 1 2; do
for V in "$@"; do
cp $R/scratch/ab/lib_$V.so $R/kmcp_amd/libkmcpgpu.so
echo "== $V" >> $OUT
cd $R
timeout 600 python bench.py --workload config2_genome_search --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('config2: value', d['value'], 'ms_per_step', d['ms_per_step'], 'k1', r.get('kmers_kernel_ms'), 'k2', r.get('kernel_ms'))" >> $OUT
if [ $rep = 1 ]; then
timeout 300 python -m pytest "tests/test_gpu_parity.py::test_genome_path_two_bit_kernel_and_its_fallback" -q -x 2>&1 | tail -1 >> $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/_p
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/_p -o p -- python $R/bench.py --workload config2_genome_search --no-cpu-baseline --no-secondary --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
python $R/profiles/extract_rocprof.py $R/gpurun_out/_p/p_results.db $R/gpurun_out/_p/x > /dev/null 2>&1
grep "k1_seg_roll2" $R/gpurun_out/_p/x_pmc.txt | awk -F'\t' '{print $3, $4, ($7-$6)/1000 " us"}' | head -5 >> $OUT
rm -rf $R/gpurun_out/_p
fi
done
done
cp /tmp/lib_orig.so $R/kmcp_amd/libkmcpgpu.so
cat $OUT
