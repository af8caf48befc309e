#!/bin/bash
# Round-3 profiling passes (on the GPU box via gpurun, from the repo root): rocprofv3 --kernel-trace --stats and, in separate
# counter-only passes as the pool requires, --pmc FETCH_SIZE — for the default bench command, the same with sector pruning off
# (KMCPG_PRUNE=0: every algorithmic byte fetched), the hit-heavy families workload (F = 64) and paired-end reads at GTDB scale;
# plus the 10k-chunk workloads and the K1 shapes.  The rocpd .db files are deleted after extraction.
#   usage: profiles/run_rocprof_r03.sh [tag=r03] [which="all"]
set -u
TAG=${1:-r03}
WHICH=${2:-all}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-secondary --no-extras"
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  local d=$OUT/_prof_$name
  rm -rf $d
  local t0=$(date +%s)
  timeout 900 rocprofv3 "$@" > $OUT/${TAG}_${name}_out.json 2> $OUT/${TAG}_${name}.err
  python $R/profiles/extract_rocprof.py $d/${name}_results.db $OUT/${TAG}_${name} >> $OUT/${TAG}_${name}.err 2>&1
  rm -rf $d
  echo "$name: $(( $(date +%s) - t0 )) s"
}
stats() { local n=$1; shift; run ${n}_stats --kernel-trace --stats -d $OUT/_prof_${n}_stats -o ${n}_stats -- "$@"; }
pmc() { local n=$1; shift; run ${n}_pmc --pmc FETCH_SIZE --kernel-trace -d $OUT/_prof_${n}_pmc -o ${n}_pmc -- "$@"; }
want() { [ "$WHICH" = all ] || [[ ",$WHICH," == *",$1,"* ]]; }

if want gtdb; then
  stats gtdb $BENCH --steps 3 --warmup 1
  pmc gtdb $BENCH --steps 2 --warmup 1
fi
if want pruneoff; then
  KMCPG_PRUNE=0 stats gtdb_pruneoff $BENCH --steps 3 --warmup 1
  KMCPG_PRUNE=0 pmc gtdb_pruneoff $BENCH --steps 2 --warmup 1
fi
if want families; then
  NB=2 stats families64 python $R/tools/bench_families.py 64
  NB=2 pmc families64 python $R/tools/bench_families.py 64
fi
if want pe; then
  stats pe_gtdb python $R/tools/bench_shapes.py gtdb
  pmc pe_gtdb python $R/tools/bench_shapes.py gtdb
fi
if want config1; then
  stats config1 $BENCH --workload config1 --steps 3 --warmup 1
  pmc config1 $BENCH --workload config1 --steps 2 --warmup 1
  KMCPG_FUSE=0 pmc config1_ungrouped $BENCH --workload config1 --steps 2 --warmup 1
fi
if want shapes; then
  stats shapes python $R/tools/bench_shapes.py
fi
if want hifi; then
  # the long-read sketch kernel: kernel stats + the SQ counters behind "bound by VALU issue"
  stats hifi python $R/tools/bench_shapes.py hifi
  run hifi_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace -d $OUT/_prof_hifi_pmc_sq -o hifi_pmc_sq -- python $R/tools/bench_shapes.py hifi
fi
if want sq; then
  run gtdb_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/_prof_gtdb_pmc_sq -o gtdb_pmc_sq -- $BENCH --steps 2 --warmup 1
  run gtdb_pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/_prof_gtdb_pmc_l2 -o gtdb_pmc_l2 -- $BENCH --steps 2 --warmup 1
fi
if want pub; then
  # the reference's published benchmark configuration: unchunked GTDB, k = 31, 47 blocks of 128-byte rows, -t 0.8
  PUB="$BENCH --workload gtdb_unchunked_k31"
  stats pub $PUB --steps 3 --warmup 1
  pmc pub $PUB --steps 2 --warmup 1
  KMCPG_PRUNE=0 pmc pub_pruneoff $PUB --steps 2 --warmup 1
  KMCPG_PRUNE=0 run pub_pruneoff_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace -d $OUT/_prof_pub_pruneoff_pmc_sq -o pub_pruneoff_pmc_sq -- $PUB --steps 2 --warmup 1
  KMCPG_PRUNE=0 run pub_pruneoff_pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/_prof_pub_pruneoff_pmc_l2 -o pub_pruneoff_pmc_l2 -- $PUB --steps 2 --warmup 1
fi
ls $OUT | grep "^${TAG}_" | head -60
