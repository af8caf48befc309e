#!/bin/bash
set -u
( time python bench.py > gpurun_out/r06_bench_line_new.json 2> gpurun_out/r06_bench_new.err ) 2> gpurun_out/r06_bench_time.txt
cp bench_detail.json gpurun_out/r06_bench_detail_new.json
cat gpurun_out/r06_bench_time.txt
tail -c 600 gpurun_out/r06_bench_new.err
wc -c gpurun_out/r06_bench_line_new.json
