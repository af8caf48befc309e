#!/usr/bin/env python3
"""What searching an index in passes costs (kmcpg_open_paged: one shard resident at a time, a batch pays passes - 1 uploads):
the real-genome family database of tools/family_db.py searched resident and in 2 / 4 passes, for two batch sizes.

usage: bench_paged.py OUT_DIR [--ecoli-strains 600] [--small-strains 100]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import family_db  # noqa: E402
from kmcp_amd import Database, default_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--ecoli-strains", type=int, default=600)
    ap.add_argument("--small-strains", type=int, default=100)
    a = ap.parse_args()
    info, reads = family_db.build(a.out_dir, a.ecoli_strains, a.small_strains, n_reads=524288, log=lambda m: print(m, file=sys.stderr))
    params = default_params(top_n_scores=1)  # (-n 1: the result stays small, what is timed is the search)
    out = {"database": {k: v for k, v in info.items() if k != "species"}}
    for B in (131072, 524288):
        h_reads = np.ascontiguousarray(reads[:B]).reshape(-1)
        h_offs = np.arange(B + 1, dtype=np.uint64) * family_db.READ_LEN
        for passes in (1, 2, 4):
            t0 = time.time()
            db = Database.open(info["db_dir"], device=0) if passes == 1 else Database.open_paged(info["db_dir"], device=0, passes=passes)
            open_s = time.time() - t0
            with db:
                m0 = db.search_packed_count(h_reads, h_offs, params=params)
                t0 = time.time()
                reps = 3
                for _ in range(reps):
                    m = db.search_packed_count(h_reads, h_offs, params=params)
                dt = (time.time() - t0) / reps
                assert m == m0
                p, up = db.paged_info()
            out[f"batch={B},passes={passes}"] = dict(open_s=open_s, s_per_batch=dt, reads_per_s=B / dt, matches=m, uploads=up,
                                                     index_bytes=int(info.get("index_bytes", 0)) or None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
