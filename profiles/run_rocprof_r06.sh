#!/bin/bash
# Round-6 profiling passes (same passes as round 4 + the ungrouped 10k-chunk index; usage: profiles/run_rocprof_r06.sh [tag=r06] [which=all])
# Round-4 profiling passes (on the GPU box via gpurun, from the repo root).  rocprofv3 --kernel-trace --stats and, in separate
# counter-only passes as the pool requires, --pmc ... for:
#   gtdb     the default bench command (GTDB-scale): kernel stats + FETCH_SIZE
#   order    the same launch in both unit orders (KMCPG_SLOT_MAJOR=1/0) under FETCH_SIZE, the TCC->EA request counters (how many
#            requests, how many "destined for DRAM", the in-flight level whose quotient with the request count is the average EA read
#            latency) and the UTCL1 translation counters: what rocprofv3 on gfx950 offers towards "how much of the fabric traffic did
#            HBM itself serve" (profiles/r04_counters.txt is the full counter list: nothing in it separates Infinity-Cache hits)
#   config2  BASELINE configs[2] (genome search, FracMinHash, 3 hashes) with matching queries: kernel stats + FETCH_SIZE
#   config4  BASELINE configs[4] (HiFi, Closed Syncmer) with matching reads, both index variants: kernel stats + FETCH_SIZE
#   pubsq    the published configuration (128-byte rows) WITH pruning: SQ counters
# STATS_ONLY=1 skips the FETCH_SIZE passes.  The rocpd .db files are deleted after extraction.    usage: profiles/run_rocprof_r04.sh [tag=r04] [which="all"]
set -u
TAG=${1:-r06}
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
pmc() { [ -n "${STATS_ONLY:-}" ] && return 0; local n=$1; shift; run ${n}_pmc --pmc FETCH_SIZE --kernel-trace -d $OUT/_prof_${n}_pmc -o ${n}_pmc -- "$@"; }
ctr() { local n=$1; local c=$2; shift; shift; run ${n} --pmc $c --kernel-trace -d $OUT/_prof_${n} -o ${n} -- "$@"; }
want() { [ "$WHICH" = all ] || [[ ",$WHICH," == *",$1,"* ]]; }

if want gtdb; then
  stats gtdb $BENCH --steps 3 --warmup 1
  pmc gtdb $BENCH --steps 2 --warmup 1
fi
if want order; then
  for sm in 1 0; do
    export KMCPG_SLOT_MAJOR=$sm
    stats order_sm${sm} $BENCH --steps 2 --warmup 1
    pmc order_sm${sm} $BENCH --steps 2 --warmup 1
    ctr order_sm${sm}_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUBBLE_sum GRBM_GUI_ACTIVE" $BENCH --steps 2 --warmup 1
    ctr order_sm${sm}_tlb "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" $BENCH --steps 2 --warmup 1
  done
  unset KMCPG_SLOT_MAJOR
fi
if want config2; then
  stats config2 $BENCH --workload config2_genome_search --steps 3 --warmup 1
  pmc config2 $BENCH --workload config2_genome_search --steps 2 --warmup 1
fi
if want config4; then
  stats config4 $BENCH --workload config4_hifi --steps 3 --warmup 1
  pmc config4 $BENCH --workload config4_hifi --steps 2 --warmup 1
  stats config4_uniform $BENCH --workload config4_hifi_uniform_sigs --steps 3 --warmup 1
  pmc config4_uniform $BENCH --workload config4_hifi_uniform_sigs --steps 2 --warmup 1
fi
if want config1u; then  # BASELINE configs[1] with every block on its own (KMCPG_FUSE=0): 39-byte rows, the 4-lane form
  export KMCPG_FUSE=0
  stats config1_ungrouped $BENCH --workload config1 --steps 3 --warmup 1
  pmc config1_ungrouped $BENCH --workload config1 --steps 2 --warmup 1
  unset KMCPG_FUSE
fi
if want config1; then
  stats config1 $BENCH --workload config1 --steps 3 --warmup 1
  pmc config1 $BENCH --workload config1 --steps 2 --warmup 1
fi
if want pubsq; then
  PUB="$BENCH --workload gtdb_unchunked_k31"
  stats pub $PUB --steps 3 --warmup 1
  pmc pub $PUB --steps 2 --warmup 1
  ctr pub_pmc_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" $PUB --steps 2 --warmup 1
  ctr pub_pmc_sq2 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" $PUB --steps 2 --warmup 1
fi
ls $OUT | grep "^${TAG}_" | head -80
