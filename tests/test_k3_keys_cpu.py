"""K3's per-match decisions compiled for the host (kmcp_amd/csrc/k3_keys.hpp: the -T test and the 128-bit sort keys that stand for
Matches.Less / SortByTCov / SortByJacc, util-db-search.go:105-145, :7471-7473) against the order and the filter of the host half
(kmcpg_finalize on a metadata-only handle — itself compared with the oracle on the GPU box, tests/test_gpu_finalize_device.py):
equal counts, equal sizes and equal scores abound, so every tie-break is exercised."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def k3(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("k3") / "k3_keys_check.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "k3_keys_check.cpp")], check=True)
    lib = C.CDLL(so)
    lib.k3_order.restype = C.c_uint32
    lib.k3_order.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_void_p]
    return lib


def test_keys_reproduce_the_host_half_order_and_the_T_filter(k3, oracle_lib, tmp_path):
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    rng = np.random.default_rng(2026)
    cfg = O.sketch_cfg(k=21)
    sizes_pool = [200, 200, 200, 333, 333, 500, 1000, 1000, 4096, 70000]  # many columns of one size: tcov / jacc ties at equal counts
    cols = []
    for c in range(240):
        h = np.unique(rng.integers(1, 2**63, size=int(rng.choice(sizes_pool)) + 50, dtype=np.int64).astype(np.uint64))[: int(rng.choice(sizes_pool))]
        cols.append((f"c{c}", 1000, 0, 1, np.sort(h)))
    db_dir = O.build_db(str(tmp_path / "db"), cfg, cols, num_hashes=1, fpr=0.3, threads=4, block_size=64)
    with Database.open(db_dir, device=-1) as db:
        ncols = int(db.info.n_cols)
        size = np.array([db.col_info(c)[3] for c in range(ncols)], dtype=np.uint64)
        assert len(set(size.tolist())) < 12 < ncols
        n_reads = 300
        qk = rng.integers(100, 180, size=n_reads).astype(np.int32)
        ql = (qk + 20).astype(np.int32)
        reads, cs, counts = [], [], []
        for r in range(n_reads):
            m = int(rng.integers(1, 200))
            c = rng.permutation(ncols)[:m]
            k = rng.integers(1, 6, size=m) * (int(qk[r]) // 6)  # few distinct counts per read: qcov ties
            reads.append(np.full(m, r, np.uint32)); cs.append(c.astype(np.uint32)); counts.append(k.astype(np.uint32))
        hits = np.empty(sum(len(x) for x in cs), dtype=lib.HIT_DTYPE)
        hits["read"], hits["col"], hits["count"] = np.concatenate(reads), np.concatenate(cs), np.concatenate(counts)
        hits = hits[rng.permutation(len(hits))]
        total = 0
        for kw, mode in ((dict(), 0), (dict(sort_by=1), 1), (dict(sort_by=2), 2), (dict(do_not_sort=1), 3), (dict(min_tcov=0.05), 0), (dict(min_tcov=0.11, sort_by=2), 2),
                         (dict(min_tcov=0.3, sort_by=1), 1)):
            p = default_params(min_qcov=0.0, min_matched=1, max_fpr=1.0, **kw)
            want = db.finalize(hits, qk, ql, params=p)
            for r in range(n_reads):
                sel = hits["read"] == r
                pairs = np.ascontiguousarray(np.stack([hits["col"][sel], hits["count"][sel]], axis=1).astype(np.uint32))
                out = np.zeros_like(pairs)
                kept = k3.k3_order(mode, size.ctypes.data, pairs.ctypes.data, len(pairs), float(qk[r]), float(kw.get("min_tcov", 0.0)), out.ctypes.data)
                ms = want.read(r)
                assert kept == len(ms), (kw, r, kept, len(ms))
                assert out[:kept, 0].tolist() == [int(x) for x in ms["col"]], (kw, r)
                assert out[:kept, 1].tolist() == [int(x) for x in ms["mkmers"]], (kw, r)
                total += kept
        assert total > 50000
