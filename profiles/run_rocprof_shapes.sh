#!/bin/bash
# usage (on the GPU box): bash profiles/run_rocprof_shapes.sh  -> gpurun_out/shapes_prof_kernel_stats.txt + shapes.json
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k1_all_forms or sketch or syncmer or dedup" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/_prof_shapes -o shapes -- python $R/tools/bench_shapes.py > $OUT/shapes.json 2> $OUT/shapes_prof.err
python $R/profiles/extract_rocprof.py $OUT/_prof_shapes/shapes_results.db $OUT/shapes_prof >> $OUT/shapes_prof.err 2>&1
rm -rf $OUT/_prof_shapes
grep "kmcpg::k1\|k_dedup" $OUT/shapes_prof_kernel_stats.txt | head -8 | cut -c1-120
