#!/bin/bash
# final artifacts of round 6, last session: the bench line (defaults) and the driver's command, then the profiling passes of the genome search (its K1 changed)
set -u
R=$PWD; OUT=$R/gpurun_out
( time python bench.py --steps 20 --warmup 5 > $OUT/r06_bench_line_final.json 2> $OUT/r06_bench_final.err ) 2> $OUT/r06_bench_final_time.txt
cp bench_detail.json $OUT/r06_bench_detail_final.json
cat $OUT/r06_bench_final_time.txt
tail -c 400 $OUT/r06_bench_final.err
bash profiles/run_rocprof_r06.sh r06d config2 > $OUT/r06d_rocprof.log 2>&1
tail -4 $OUT/r06d_rocprof.log
ls $OUT | grep r06d
