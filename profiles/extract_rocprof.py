#!/usr/bin/env python3
"""Turns a rocprofv3 results .db (rocpd sqlite) into the small text summaries committed under profiles/.

usage: extract_rocprof.py <results.db> <out_prefix>
writes <out_prefix>_kernel_stats.txt (the --stats table + one line per dispatch of our kernels) and, if the run
collected counters, <out_prefix>_pmc.txt (per-dispatch counter values of our kernels).
"""
import sqlite3
import sys


def short(name, n=90):
    return name if len(name) <= n else name[:n - 3] + "..."


def main(db, prefix):
    con = sqlite3.connect(db)
    cur = con.cursor()
    with open(prefix + "_kernel_stats.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (top_kernels view); durations in ns\n")
        f.write("name\tcalls\ttotal_ns\taverage_ns\tpercent\n")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration*1000,average*1000,percentage from top_kernels"):
            f.write(f"{short(name)}\t{calls}\t{total:.0f}\t{avg:.0f}\t{pct:.4f}\n")
        f.write("\n# dispatches of kmcpg kernels\nname\tstart_ns\tduration_ns\tgrid_x\tworkgroup_x\tlds\tvgpr\tsgpr\tscratch\n")
        for r in cur.execute("select name,start,duration,grid_x,workgroup_x,lds_size,vgpr_count,sgpr_count,scratch_size from kernels "
                             "where name like '%kmcpg%' and name not like '%synth_fill%' order by start"):
            f.write("\t".join(str(x) if i else short(str(x), 60) for i, x in enumerate(r)) + "\n")
    n = cur.execute("select count(*) from counters_collection").fetchone()[0]
    if n:
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        with open(prefix + "_pmc.txt", "w") as f:
            f.write("# rocprofv3 --pmc per-dispatch counter values (counters_collection view), kmcpg kernels only\n")
            want = [c for c in ("dispatch_id", "kernel_name", "name", "counter_name", "value", "grid_size", "start", "end") if c in cols]
            f.write("\t".join(want) + "\n")
            kn = "kernel_name" if "kernel_name" in cols else "name"
            for r in cur.execute(f"select {','.join(want)} from counters_collection where {kn} like '%kmcpg%' and {kn} not like '%synth_fill%'"):
                f.write("\t".join(short(str(x), 60) for x in r) + "\n")
        print("pmc columns:", cols)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
