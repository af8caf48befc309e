#!/usr/bin/env python3
"""Timeline of a rocprofv3 --hip-trace --kernel-trace run (rocpd sqlite): kmcpg kernels and HIP API calls after the last
planting kernel (= the warm-up and timed steps of bench.py), merged by start time, times in us relative to the first line.

usage: extract_timeline.py <results.db> <out.txt> [max_lines=6000]"""
import sqlite3
import sys


def main(db, out, max_lines=6000):
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    with open(out, "w") as f:
        f.write("# objects: " + " ".join(n for n in names if not n.startswith("rocpd_info")) [:3000] + "\n")
        ks = list(cur.execute("select name,start,end from kernels order by start"))
        t0 = 0
        for nm, st, en in ks:
            if "k_plant" in nm or "synth_fill" in nm:
                t0 = en
        rows = [(st, en, "K", nm.replace("kmcpg::", "").replace("void ", "")[:70], 0) for nm, st, en in ks if st >= t0 and "kmcpg" in nm]
        api = []
        for view in ("regions", "region", "api_calls"):
            if view in names:
                cols = [d[1] for d in cur.execute(f"pragma table_info({view})")]
                f.write(f"# {view} columns: {cols}\n")
                if {"name", "start", "end"} <= set(cols):
                    tid = "tid" if "tid" in cols else "0"
                    api = list(cur.execute(f"select name,start,end,{tid} from {view} where start >= ? order by start", (t0,)))
                    break
        skip = ("hipGetLastError", "hipGetDevice", "hipSetDevice", "hipPeekAtLastError", "hipGetDeviceCount", "hipDeviceGetAttribute", "__hip")
        rows += [(st, en, "A", nm[:70], tid) for nm, st, en, tid in api if not any(nm.startswith(s) for s in skip)]
        rows.sort()
        if not rows:
            f.write("# nothing found\n")
            return
        base = rows[0][0]
        f.write("start_us\tdur_us\tkind\tname\ttid\n")
        for st, en, kind, nm, tid in rows[:max_lines]:
            f.write(f"{(st - base) / 1e3:.1f}\t{(en - st) / 1e3:.1f}\t{kind}\t{nm}\t{tid}\n")
        f.write(f"# {len(rows)} rows in window, {min(len(rows), max_lines)} written\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6000)
