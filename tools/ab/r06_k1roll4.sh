#!/bin/bash
set -u
OUT=gpurun_out/r06_k1roll4.txt
: > $OUT
W=config4_hifi_uniform_sigs
for f in 35 3; do
  echo "== KMCPG_K1_FLAGS=$f" >> $OUT
  KMCP_BENCH_TRACE=1 KMCPG_K1_FLAGS=$f python bench.py --workload $W --no-cpu-baseline --no-secondary --no-extras --steps 8 --warmup 3 2>&1 >/dev/null | grep "rank 0 step" | tail -8 >> $OUT
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 35 3; do
KMCPG_K1_FLAGS=$f rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/_k1t -o k1t -- python $R/bench.py --workload $W --no-cpu-baseline --no-secondary --no-extras --steps 4 --warmup 2 > /dev/null 2>&1
python - <<PY >> $R/$OUT
import csv, glob
ev = []
for f in glob.glob("$R/gpurun_out/_k1t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("kmcpg::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:36]))
ev.sort()
# the last two timed steps: from the second-to-last k2_cobs<64 back
k2 = [i for i, e in enumerate(ev) if e[2].startswith("k2_cobs<64")]
lo = k2[-3] + 1
t0 = ev[lo][0]
print("== timeline KMCPG_K1_FLAGS=$f (ms from the first kernel after a COBS kernel)")
prev_end = None
for s, e, n in ev[lo:lo + 40]:
    gap = "" if prev_end is None else f"  gap {(s - prev_end) / 1e6:6.3f}"
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}  {n}{gap}")
    prev_end = max(prev_end or 0, e)
PY
rm -rf $R/gpurun_out/_k1t
done
cat $R/$OUT
