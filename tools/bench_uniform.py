#!/usr/bin/env python3
"""What kmcpg_build_cfg.uniform_sigs buys on a database shaped like BASELINE configs[1] but with UNEVEN chunks (the real case):
9 984 columns, `kmcp index -j 32` => 32 blocks of 312 columns (39-byte rows), k-mer counts per column spread 3x.  The reference's
sizing gives every block its own NumSigs => 32 lone blocks, one 64-byte request per (k-mer, block); uniform_sigs = 1 / 2 lets the
resident layout put blocks side by side.  Builds the three databases with kmcpg_build_db (columns = random 64-bit hash lists +
the k-mers of planted 150-bp fragments), searches the same 1 M reads against each and prints sizes, groups and rates.

usage: bench_uniform.py [kmers_per_col_mean=200000] [out_dir=/tmp/kmcp_uniform]
"""
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kmcp_amd import Database, default_params, lib  # noqa: E402

N_COLS, READ_LEN, K = 9984, 150, 21


def main():
    mean_kmers = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    out_dir = sys.argv[2] if len(sys.argv) > 2 else "/tmp/kmcp_uniform"
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    # planted fragments and their k-mer hashes (K1 of the product, through a throw-away synthetic handle)
    F = 16384
    frag = acgt[rng.integers(0, 4, size=(F, READ_LEN))]
    spec = lib.SynthSpec(k=K, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=8, num_sigs=1024, kmers_per_col=100, seed=1)
    with Database.open_synthetic(spec) as tiny:
        d_seq = torch.from_numpy(frag.reshape(-1).copy()).to(dev)
        d_off = (torch.arange(F + 1, device=dev, dtype=torch.int64) * READ_LEN).contiguous()
        d_h = torch.zeros(F * READ_LEN, dtype=torch.int64, device=dev)
        d_ko = torch.zeros(F, dtype=torch.int64, device=dev)
        d_nk = torch.zeros(F, dtype=torch.int32, device=dev)
        tiny.kmers_device(d_seq.data_ptr(), d_off.data_ptr(), F, F * READ_LEN, READ_LEN, d_h.data_ptr(), F * READ_LEN, d_ko.data_ptr(), d_nk.data_ptr())
        torch.cuda.synchronize()
        h = d_h.cpu().numpy().view(np.uint64).reshape(F, READ_LEN)[:, :READ_LEN - K + 1]
        assert int(d_nk.min().item()) == READ_LEN - K + 1
    frag_col = rng.integers(0, N_COLS, size=F)
    planted = {}
    for f in range(F):
        planted.setdefault(int(frag_col[f]), []).append(h[f])
    sizes = rng.integers(mean_kmers // 2, mean_kmers * 3 // 2, size=N_COLS)
    t0 = time.time()
    cols = []
    for c in range(N_COLS):
        hs = np.random.default_rng(1000 + c).integers(1, 2**63, size=int(sizes[c]), dtype=np.int64).view(np.uint64)
        if c in planted:
            hs = np.concatenate([hs] + planted[c])
        cols.append((f"ref{c // 10:05d}", 4000000, c % 10, 10, hs))
    gen_s = time.time() - t0
    # the reads: 90 % mutated copies of planted fragments (1 % substitutions, half reverse-complemented), 10 % random
    B = 1 << 20
    src = rng.integers(0, F, size=B)
    code = np.searchsorted(acgt, frag[src])  # A C G T -> 0..3
    sub = rng.random((B, READ_LEN)) < 0.01
    code = np.where(sub, rng.integers(0, 4, size=(B, READ_LEN)), code)
    rnd = rng.random(B) < 0.10
    code[rnd] = rng.integers(0, 4, size=(int(rnd.sum()), READ_LEN))
    rc = rng.random(B) < 0.5
    code[rc] = 3 - code[rc][:, ::-1]
    reads = acgt[code].reshape(-1).copy()
    offs = (np.arange(B + 1, dtype=np.uint64) * READ_LEN)
    want_col = np.where(rnd, -1, frag_col[src])
    res = {"columns": N_COLS, "kmers_per_col": [int(sizes.min()), int(sizes.max())], "hash_gen_s": gen_s, "batch_reads": B}
    params = default_params()
    for mode in (0, 1, 2):
        d = os.path.join(out_dir, f"mode{mode}")
        shutil.rmtree(d, ignore_errors=True)
        t0 = time.time()
        db_dir = lib.build_db(d, cols, k=K, threads=32, uniform_sigs=mode)
        build_s = time.time() - t0
        t0 = time.time()
        with Database.open(db_dir, device=0) as db:
            load_s = time.time() - t0
            nb = int(db.info.n_blocks)
            bi = [db.block_info(b) for b in range(nb)]
            d_reads = torch.from_numpy(reads).to(dev)
            d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
            cap = 8 * B
            d_hits = torch.empty((cap, 3), dtype=torch.int32, device=dev)
            d_cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            d_qk = torch.zeros(B, dtype=torch.int32, device=dev)
            d_ql = torch.zeros(B, dtype=torch.int32, device=dev)
            ms = []
            for level in (1, 1, 1, 1, 2):  # timed at level 1; the last run counts the row loads (one atomic per wave and row group: slow)
                db.set_profiling(level)
                db.query_device(d_reads.data_ptr(), d_offs.data_ptr(), B, B * READ_LEN, READ_LEN, d_hits.data_ptr(), cap, d_cnt.data_ptr(), d_qk.data_ptr(),
                                d_ql.data_ptr(), params=params)
                torch.cuda.synchronize()
                if level == 1:
                    ms.append(db.last_timing())
            gathered = db.last_gathered_bytes()
            db.set_profiling(1)
            n_hits = int(d_cnt[0].item())
            hh = d_hits[:n_hits].cpu().numpy()
            # the builder sorts the columns by k-mer count: global column ids are found through (name, chunk index)
            key_of = {}
            for c in range(int(db.info.n_cols)):
                name, tidx, _, _ = db.col_info(c)
                key_of[(name, tidx & 0xffff)] = c
            got = set(zip(hh[:, 0].tolist(), hh[:, 1].tolist()))
            pl = np.nonzero(want_col >= 0)[0][:20000]
            recall = sum((int(r), key_of[(f"ref{int(want_col[r]) // 10:05d}", int(want_col[r]) % 10)]) in got for r in pl) / len(pl)
            # the whole boundary once (host buffers in, finalized matches out)
            t0 = time.time()
            nm = db.search_packed_count(reads, offs, params=params)
            t0 = time.time()
            nm = db.search_packed_count(reads, offs, params=params)
            search_s = time.time() - t0
            k2 = min(m[1] for m in ms[1:])
            alg = B * (READ_LEN - K + 1) * sum(b["row_bytes"] for b in bi)
            res[f"uniform_sigs={mode}"] = dict(
                index_bytes=int(db.info.matrix_bytes), distinct_num_sigs=len({b["num_sigs"] for b in bi}), strides=sorted({b["stride"] for b in bi}),
                build_s=build_s, load_s=load_s, k1_ms=min(m[0] for m in ms[1:]), k2_ms=k2, reads_per_s_kernels=B / ((k2 + min(m[0] for m in ms[1:])) * 1e-3),
                gathered_bytes=gathered, algorithmic_bytes=alg, effective_gbps=alg / (k2 * 1e-3) / 1e9, achieved_gbps=gathered / (k2 * 1e-3) / 1e9,
                hits=n_hits, planted_recall=recall, matches=nm, search_batch_reads_per_s=B / search_s)
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
