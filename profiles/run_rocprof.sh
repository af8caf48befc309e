#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root.  Produces the text summaries that get committed under
# profiles/: rocprofv3 --kernel-trace --stats of the bench command, and (separate passes, counters only with
# --kernel-trace as the pool requires) --pmc FETCH_SIZE for both workloads.  The rocpd .db files are large and are
# deleted after extraction.
#   usage: profiles/run_rocprof.sh <tag>      e.g. r01
set -u
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  local d=$OUT/_prof_$name
  rm -rf $d
  timeout 900 rocprofv3 "$@" > $OUT/${TAG}_${name}_bench.json 2> $OUT/${TAG}_${name}.err
  python $R/profiles/extract_rocprof.py $d/${name}_results.db $OUT/${TAG}_${name} >> $OUT/${TAG}_${name}.err 2>&1
  rm -rf $d
}
run gtdb_stats --kernel-trace --stats -d $OUT/_prof_gtdb_stats -o gtdb_stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-extras
run gtdb_pmc --pmc FETCH_SIZE --kernel-trace -d $OUT/_prof_gtdb_pmc -o gtdb_pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-extras
run config1_stats --kernel-trace --stats -d $OUT/_prof_config1_stats -o config1_stats -- python $R/bench.py --workload config1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-extras
run config1_pmc --pmc FETCH_SIZE --kernel-trace -d $OUT/_prof_config1_pmc -o config1_pmc -- python $R/bench.py --workload config1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-extras
ls -la $OUT | head -40
# extra counter passes for the dominant kernel (each its own run): wave occupancy / stall picture and L2 hit rate
run gtdb_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/_prof_gtdb_pmc_sq -o gtdb_pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-extras
run gtdb_pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/_prof_gtdb_pmc_l2 -o gtdb_pmc_l2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-extras
ls $OUT | head -60
# the 10k-chunk index with every block on its own (KMCPG_FUSE=0: what a database with distinct NumSigs gets)
KMCPG_FUSE=0 run config1_ungrouped_pmc --pmc FETCH_SIZE --kernel-trace -d $OUT/_prof_config1_ungrouped_pmc -o config1_ungrouped_pmc -- python $R/bench.py --workload config1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-extras
