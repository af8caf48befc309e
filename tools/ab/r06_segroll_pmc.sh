#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --workload config2_genome_search --no-cpu-baseline --no-secondary --no-extras --steps 3 --warmup 1"
run() {
  local name=$1; shift
  local d=$OUT/_prof_$name
  rm -rf $d
  timeout 600 rocprofv3 "$@" > /dev/null 2> $OUT/r06_${name}.err
  python $R/profiles/extract_rocprof.py $d/${name}_results.db $OUT/r06_${name} >> $OUT/r06_${name}.err 2>&1
  rm -rf $d
}
run g_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace -d $OUT/_prof_g_sq -o g_sq -- $BENCH
run g_sq2 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/_prof_g_sq2 -o g_sq2 -- $BENCH
cd $R
python - <<'PY'
import collections, glob
for f in sorted(glob.glob("gpurun_out/r06_g_sq*_pmc.txt")):
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
        if "k1_seg_roll2" not in kn and "k2_cobs" not in kn:
            continue
        if float(d["end"]) - float(d["start"]) < 500000:
            continue
        acc[kn[:44]][d["counter_name"]].append(float(d["value"]))
        acc[kn[:44]]["duration_ns"].append(float(d["end"]) - float(d["start"]))
    print("==", f)
    for kn, cs in acc.items():
        print("  ", kn, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
