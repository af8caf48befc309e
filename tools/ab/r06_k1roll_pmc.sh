#!/bin/bash
# SQ counters of k1_windows_roll vs k1_windows_wave on the HiFi workload (separate counter-only passes)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --workload config4_hifi_uniform_sigs --no-cpu-baseline --no-secondary --no-extras --steps 3 --warmup 1"
run() {
  local name=$1; shift
  local d=$OUT/_prof_$name
  rm -rf $d
  timeout 600 rocprofv3 "$@" > /dev/null 2> $OUT/r06_${name}.err
  python $R/profiles/extract_rocprof.py $d/${name}_results.db $OUT/r06_${name} >> $OUT/r06_${name}.err 2>&1
  rm -rf $d
}
for f in 3; do
  export KMCPG_K1_FLAGS=$f
  run k1f${f}_stats --kernel-trace --stats -d $OUT/_prof_k1f${f}_stats -o k1f${f}_stats -- $BENCH
  run k1f${f}_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace -d $OUT/_prof_k1f${f}_sq -o k1f${f}_sq -- $BENCH
  run k1f${f}_sq2 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/_prof_k1f${f}_sq2 -o k1f${f}_sq2 -- $BENCH
done
cd $R
python - <<'PY'
import collections, glob
for f in sorted(glob.glob("gpurun_out/r06_k1f*_kernel_stats.txt")):
    print("==", f)
    for ln in open(f):
        if "k1_windows" in ln and ln.count("\t") == 4:
            print("  ", ln.strip()[:160])
for f in sorted(glob.glob("gpurun_out/r06_k1f*_pmc.txt")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    hdr = None
    for ln in open(f):
        if ln.startswith("#"):
            continue
        p = ln.rstrip("\n").split("\t")
        if hdr is None:
            hdr = p
            continue
        d = dict(zip(hdr, p))
        kn = d.get("kernel_name") or d.get("name")
        if "k1_windows" not in kn:
            continue
        if float(d["end"]) - float(d["start"]) < 300000:  # (the planting calls leave every read to the fallback: the rolling kernel exits at once there)
            continue
        acc[kn[:40]][d["counter_name"]].append(float(d["value"]))
        acc[kn[:40]]["duration_ns"].append(float(d["end"]) - float(d["start"]))
    print("==", f)
    for kn, cs in acc.items():
        print("  ", kn, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
