#!/bin/bash
# timeline of a HiFi step on one and on two kernel streams (rocprofv3 --kernel-trace): does k1_windows_roll of step i+1 run beside k2_cobs_pair of step i?
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for M in 1 2; do
  d=$OUT/_prof_h$M
  rm -rf $d
  if [ $M = 2 ]; then export KMCPG_WS_SLOTS=2 KMCP_BENCH_STREAMS=2; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $d -o h$M -- python $R/bench.py --workload config4_hifi_uniform_sigs --no-cpu-baseline --no-secondary --no-extras --steps 8 --warmup 3 > /dev/null 2> $OUT/r06_hifi_trace_$M.err
  python $R/profiles/extract_rocprof.py $d/h${M}_results.db $OUT/r06_hifi_trace_$M >> $OUT/r06_hifi_trace_$M.err 2>&1
  rm -rf $d
done
cd $R
python - <<'PY'
for M in (1, 2):
    rows = []
    on = False
    for ln in open(f"gpurun_out/r06_hifi_trace_{M}_kernel_stats.txt"):
        if ln.startswith("# dispatches"):
            on = True
            continue
        if not on or ln.startswith("name"):
            continue
        p = ln.rstrip("\n").split("\t")
        if len(p) < 9:
            continue
        rows.append((p[0], int(p[1]), int(p[2]), p[3], p[4], p[5], p[6]))
    # the last 3 steps: from the third-last k1_windows_roll on
    idx = [i for i, r in enumerate(rows) if "k1_windows_roll" in r[0]]
    lo = idx[-3] if len(idx) >= 3 else 0
    t0 = rows[lo][1]
    print(f"== {M} stream(s)")
    for r in rows[lo:]:
        if r[2] < 20000 and "k2_cobs" not in r[0] and "windows_roll" not in r[0] and "dedup_bucket<2048" not in r[0]:
            continue
        print(f"  {r[0][:52]:52s} start {(r[1]-t0)/1e6:8.3f} ms  dur {r[2]/1e6:7.3f} ms  grid {r[3]:>9s} wg {r[4]:>5s} lds {r[5]:>6s} vgpr {r[6]}")
PY
