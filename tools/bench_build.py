#!/usr/bin/env python3
"""Index-building throughput: kmcpg_build_db (GPU scatter) vs the oracle's CPU restatement of `kmcp index`, on the same hash lists."""
import filecmp
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmcp_amd import lib  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    n_cols, n_hashes = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 345510
    rng = np.random.default_rng(1)
    cols = []
    for c in range(n_cols):
        h = np.unique(rng.integers(1, 1 << 63, size=n_hashes + c % 7, dtype=np.uint64))
        cols.append((f"ref{c:06d}", 4000000, c % 10, 10, h))
    total = sum(len(c[4]) for c in cols)
    out = {"columns": n_cols, "kmers": total}
    with tempfile.TemporaryDirectory() as tmp:
        lib.build_db(os.path.join(tmp, "warm"), cols[:8], k=21, block_size=8)  # HIP context, first-touch
        t0 = time.perf_counter()
        g = lib.build_db(os.path.join(tmp, "gpu"), cols, k=21, num_hashes=1, fpr=0.3, block_size=n_cols)
        out["gpu_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        r = O.build_db(os.path.join(tmp, "cpu"), O.sketch_cfg(k=21), cols, num_hashes=1, fpr=0.3, block_size=n_cols)
        out["cpu_oracle_s"] = time.perf_counter() - t0
        out["identical"] = filecmp.cmp(os.path.join(g, "_block001.uniki"), os.path.join(r, "_block001.uniki"), shallow=False)
        out["block_bytes"] = os.path.getsize(os.path.join(g, "_block001.uniki"))
    out["gpu_Mkmers_per_s"] = total / out["gpu_s"] / 1e6
    out["cpu_Mkmers_per_s"] = total / out["cpu_oracle_s"] / 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
