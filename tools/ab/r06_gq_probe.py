"""whole-genome queries (all k-mers) against the published-configuration index: one per call and four in one call"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from kmcp_amd import Database, default_params, lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "gtdb_unchunked_k31"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = dict(bench.WORKLOADS[name])
spec = lib.SynthSpec(k=wl["k"], num_hashes=wl["num_hashes"], fpr=wl["fpr"], n_blocks=wl["n_blocks"], cols_per_block=wl["cols_per_block"],
                     num_sigs=wl["num_sigs"], kmers_per_col=wl["kmers_per_col"], seed=42, sigs_step=wl.get("sigs_step", 0), scale=wl.get("scale", 0),
                     syncmer_s=wl.get("syncmer_s", 0), minimizer_w=wl.get("minimizer_w", 0))
db = Database.open_synthetic(spec, device=0)
rng = np.random.default_rng(1)
lens = [int(x) for x in np.linspace(4600000, 5600000, n)]
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=sum(lens))]
offs = np.zeros(n + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
p = default_params()
p.min_qcov = 0.5
for rep in range(3):
    per = []
    for i in range(n):
        s_i = np.ascontiguousarray(seq[int(offs[i]):int(offs[i + 1])])
        o_i = np.array([0, len(s_i)], dtype=np.uint64)
        t = time.perf_counter()
        db.search_packed_count(s_i, o_i, params=p)
        per.append(time.perf_counter() - t)
    t = time.perf_counter()
    db.search_packed_count(seq, offs, params=p)
    tb = time.perf_counter() - t
    print("rep %d: one per call %s ms; %d in one call %.2f ms (%.2f per genome)" % (rep, " ".join("%.2f" % (x * 1e3) for x in per), n, tb * 1e3, tb * 1e3 / n), flush=True)
db.close()
