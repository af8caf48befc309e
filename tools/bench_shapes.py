#!/usr/bin/env python3
"""Kernel timings (HIP events inside libkmcpgpu) for the BASELINE.json configs that are parity cases rather than the
headline bench line: paired-end 2x150 (sort+unique for every query), HiFi 10-kb reads against a Closed-Syncmer index
(config 4) and whole-genome queries against a FracMinHash 3-hash index of 50 k references (config 2)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmcp_amd import Database, default_params, lib  # noqa: E402

dev = torch.device("cuda:0")
ACGT = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)


def rand_reads(n, lens, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(lens.to(dev), 0)
    total = int(offs[-1].item())
    return ACGT[torch.randint(0, 4, (total,), generator=g, device=dev)].contiguous(), offs.contiguous(), total


def run(db, seqs, offs, n, total, maxlen, params, seqs2=None, offs2=None, reps=3):
    cap = 8 * n + 4096
    hits = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    qk = torch.zeros(n, dtype=torch.int32, device=dev)
    ql = torch.zeros(n, dtype=torch.int32, device=dev)
    db.set_profiling(True)
    t = []
    for _ in range(reps):
        db.query_device(seqs.data_ptr(), offs.data_ptr(), n, total, maxlen, hits.data_ptr(), cap, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(),
                        params=params, d_seqs2=seqs2.data_ptr() if seqs2 is not None else None, d_offs2=offs2.data_ptr() if offs2 is not None else None)
        torch.cuda.synchronize()
        t.append(db.last_timing())
    k1, k2 = min(x[0] for x in t), min(x[1] for x in t)
    return dict(reads=n, bases=total, kmers_ms=k1, cobs_ms=k2, reads_per_s=n / ((k1 + k2) * 1e-3), mean_qkmers=float(qk.float().mean().item()),
                hits=int(cnt[0].item()))


def main():
    out = {}
    only = sys.argv[1] if len(sys.argv) > 1 else ""  # "gtdb": the two GTDB-scale shapes only
    # sigs_step 0: every block has the same NumSigs (equal-length chunks, as BASELINE configs[1] is built) and the 32 narrow
    # blocks share one group of 1248-byte rows; sigs_step 7: a different NumSigs per block, every block on its own (39-byte rows)
    for tag, step in ((("", 0), ("_distinct_numsigs", 7)) if only not in ("gtdb", "hifi") else ((("", 0),) if only == "hifi" else ())):
        # paired-end 2 x 150 against the 10k-chunk index
        spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=312, num_sigs=1121470, kmers_per_col=400000, seed=1, sigs_step=step)
        with Database.open_synthetic(spec) as db:
            if only != "hifi":
                n = 262144
                lens = torch.full((n,), 150, dtype=torch.int64)
                s1, o1, t1 = rand_reads(n, lens, 1)
                s2, o2, t2 = rand_reads(n, lens, 2)
                out["pe_2x150_vs_10k_chunks" + tag] = run(db, s1, o1, n, t1 + t2, 150, default_params(), s2, o2)
        # HiFi ~10 kb against a Closed-Syncmer (s=11) 10k-chunk index
        spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=312, num_sigs=300000, kmers_per_col=100000, seed=2, syncmer_s=11,
                             sigs_step=step)
        with Database.open_synthetic(spec) as db:
            n = 16384
            g = torch.Generator()
            g.manual_seed(3)
            lens = torch.clamp((torch.randn(n, generator=g) * 2000 + 10000).long(), 2000, 20000)
            s, o, t = rand_reads(n, lens, 4)
            out["hifi_10kb_syncmer_vs_10k_chunks" + tag] = run(db, s, o, n, t, int(lens.max()), default_params())
    if only == "hifi":
        print(json.dumps(out, indent=1))
        return
    # genome search: 4-Mbp queries against a FracMinHash (scale 1000), 3-hash, fpr 0.001 index of 50 k references
    spec = lib.SynthSpec(k=21, num_hashes=3, fpr=0.001, n_blocks=8, cols_per_block=6256, num_sigs=431000, kmers_per_col=10000, seed=3, scale=1000, sigs_step=13)
    with Database.open_synthetic(spec) as db:
        n = 64 if only != "gtdb" else 1
        lens = torch.full((n,), 4000000, dtype=torch.int64)
        s, o, t = rand_reads(n, lens, 5)
        out["genome_4Mbp_fracminhash_vs_50k_refs"] = run(db, s, o, n, t, 4000000, default_params(min_qcov=0.4, sort_by=2))
    # whole-genome query, all k-mers (`-g`), against the GTDB-scale index: the reference's "whole-genome query" benchmark
    # (benchmarks/searching/README.md:139-163: 12.7-13.7 s hot on 40 threads against an unchunked 55 GB index)
    free_b, _ = torch.cuda.mem_get_info(dev)
    if free_b > 80e9:
        spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=14976, num_sigs=967708, kmers_per_col=345510, seed=4, sigs_step=64)
        with Database.open_synthetic(spec) as db:
            # paired-end 2 x 150 at GTDB scale: 260 k-mers per pair => 16 counter planes
            n = 131072
            lens = torch.full((n,), 150, dtype=torch.int64)
            s1, o1, t1 = rand_reads(n, lens, 11)
            s2, o2, t2 = rand_reads(n, lens, 12)
            out["pe_2x150_vs_gtdb_scale"] = run(db, s1, o1, n, t1 + t2, 150, default_params(), s2, o2)
            n = 2
            lens = torch.full((n,), 5000000, dtype=torch.int64)
            s, o, t = rand_reads(n, lens, 6)
            out["genome_5Mbp_all_kmers_vs_gtdb_scale"] = run(db, s, o, n, t, 5000000, default_params(min_qcov=0.5, dedup_threshold=256))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
