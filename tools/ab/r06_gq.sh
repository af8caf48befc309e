#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/tools/ab/r06_gq_probe.py gtdb_unchunked_k31 4 > $OUT/r06_gq_probe.txt 2>&1
rm -rf $OUT/_prof_gq
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/_prof_gq -o gq -- python $R/tools/ab/r06_gq_probe.py gtdb_unchunked_k31 4 > /dev/null 2> $OUT/r06_gq_prof.err
python $R/profiles/extract_rocprof.py $OUT/_prof_gq/gq_results.db $OUT/r06_gq >> $OUT/r06_gq_prof.err 2>&1
rm -rf $OUT/_prof_gq
cat $OUT/r06_gq_probe.txt
grep kmcpg $OUT/r06_gq_kernel_stats.txt | head -30
