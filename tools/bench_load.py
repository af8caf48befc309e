#!/usr/bin/env python3
"""Database load rate (kmcpg_open: .uniki files -> padded rows resident in HBM) on GTDB-shaped blocks.

Writes ONE block of 14 976 columns x 968 700 rows (1.8 GB, through the oracle's .uniki writer with NumSigs forced) into
/tmp, hard-links it N times (so the page cache holds it once and the run measures the loader, not the disk), and times
Database.open for a few loader-thread counts.

usage: bench_load.py [--blocks 8] [--threads 1,4,8]
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402  (test-data generation only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--threads", default="1,4,8")
    ap.add_argument("--cols", type=int, default=14976)
    ap.add_argument("--num-sigs", type=int, default=968700)
    ap.add_argument("--dir", default="/tmp/kmcp_load_bench")
    a = ap.parse_args()
    O.build()
    shutil.rmtree(a.dir, ignore_errors=True)
    cfg = O.sketch_cfg(k=21)
    rng = np.random.default_rng(1)
    cols = [(f"c{i:05d}", 1000, 0, 1, np.sort(rng.integers(1, 2**63, 8, dtype=np.uint64))) for i in range(a.cols)]
    db = O.build_db(a.dir, cfg, cols, threads=1, block_size=a.cols)
    arr = (O.Column * len(cols))()
    keep = []
    for i, (name, gsize, ci, nch, h) in enumerate(cols):
        keep.append(h)
        arr[i] = O.Column(name.encode(), gsize, ci, nch, h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h))
    first = os.path.join(db, "_block001.uniki")
    rc = O.lib().ko_write_block(first.encode(), 21, 1, 1, 0.3, a.num_sigs, arr, len(cols))
    assert rc == 0
    names = ["_block%03d.uniki" % (i + 1) for i in range(a.blocks)]
    for n in names[1:]:
        os.link(first, os.path.join(db, n))
    yml = open(os.path.join(db, "__db.yml")).read()
    head = yml[:yml.index("files:")]
    open(os.path.join(db, "__db.yml"), "w").write(head + "files:\n" + "".join(f"- {n}\n" for n in names))
    size = os.path.getsize(first) * a.blocks
    with open(first, "rb") as fh:  # warm the page cache
        while fh.read(1 << 26):
            pass
    from kmcp_amd import Database
    out = {"blocks": a.blocks, "bytes": size, "runs": []}
    for t in [int(x) for x in a.threads.split(",")]:
        os.environ["KMCPG_LOAD_THREADS"] = str(t)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            d = Database.open(db, device=0)
            dt = time.perf_counter() - t0
            assert d.info.n_blocks == a.blocks and d.info.matrix_bytes_local == a.blocks * a.num_sigs * ((a.cols + 7) // 8)
            d.close()
            best = dt if best is None else min(best, dt)
        out["runs"].append({"threads": t, "seconds": round(best, 3), "GB_per_s": round(size / best / 1e9, 2)})
    print(json.dumps(out))
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
