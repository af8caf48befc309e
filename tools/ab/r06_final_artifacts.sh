#!/bin/bash
# final artifacts of round 6: the driver's bench command, then the profiling passes of the workloads whose kernels changed late in the round
set -u
R=$PWD; OUT=$R/gpurun_out
( time python bench.py > $OUT/r06_bench_line_final.json 2> $OUT/r06_bench_final.err ) 2> $OUT/r06_bench_final_time.txt
cp bench_detail.json $OUT/r06_bench_detail_final.json
cat $OUT/r06_bench_final_time.txt
tail -c 400 $OUT/r06_bench_final.err
bash profiles/run_rocprof_r06.sh r06c config2,config4 > $OUT/r06c_rocprof.log 2>&1
tail -8 $OUT/r06c_rocprof.log
