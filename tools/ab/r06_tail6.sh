#!/bin/bash
set -u
OUT=$PWD/gpurun_out
export PYTHONPATH=$PWD
rm -f $OUT/r06_tail_report.txt
( time KMCP_FUZZ_TAIL_REPORT=$OUT/r06_tail_report.txt KMCP_FUZZ_TAIL_SEEDS=400 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 12 --timeout 900 -p no:cacheprovider -k "wide_rows" ) > $OUT/r06_tail_fuzz.txt 2>&1
tail -4 $OUT/r06_tail_fuzz.txt
python - <<'PY'
import collections
rows = [l.split() for l in open("gpurun_out/r06_tail_report.txt")]
by = collections.defaultdict(lambda: [0, 0, 0])
for r in rows:
    tw = int(r[-1].split("=")[1])
    b = by[r[1] + " " + r[2]]
    b[0] += 1; b[1] += tw > 0; b[2] += tw
for k in sorted(by):
    print(k, "draws %d, with tail waves %d, waves %d" % tuple(by[k]))
PY
