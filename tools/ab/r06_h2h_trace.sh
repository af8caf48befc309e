#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_h2h_trace.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/_h2h_trace -o h2h -- python $R/tools/h2h_probe.py ${1:-config2_genome_search} --packed --batches 12 > /dev/null 2>&1
python - <<PY >> $OUT 2>&1
import csv, glob
base = "$R/gpurun_out/_h2h_trace"
ev = []
for f in glob.glob(base + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].replace("kmcpg::", "").replace("void ", "")[:34]))
for f in glob.glob(base + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")))
ev.sort()
t_end = ev[-1][1]
sel = [e for e in ev if e[0] > t_end - 40_000_000 and (e[1] - e[0]) > 30_000]
t0 = sel[0][0]
last_k_end = None
for s, e, n in sel:
    gap = ""
    if n.startswith("K"):
        if last_k_end is not None and s - last_k_end > 50_000:
            gap = f"   <- {(s-last_k_end)/1e6:.3f} ms after the previous kernel"
        last_k_end = max(last_k_end or 0, e)
    print(f"{(s-t0)/1e6:9.3f} .. {(e-t0)/1e6:9.3f} ms  {(e-s)/1e6:7.3f}  {n}{gap}")
PY
rm -rf $R/gpurun_out/_h2h_trace
cat $OUT
