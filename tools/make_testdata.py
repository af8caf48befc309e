#!/usr/bin/env python3
"""Builds a synthetic kmcp database (.uniki files + __db.yml, through the oracle's `compute`+`index` restatement) and a FASTQ
file of reads sampled from it: inputs for end-to-end runs of kmcp-search.

usage: make_testdata.py <out_dir> [--genomes 200] [--genome-len 200000] [--chunks 10] [--reads 2000000] [--threads 32]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--genomes", type=int, default=200)
    ap.add_argument("--genome-len", type=int, default=200000)
    ap.add_argument("--chunks", type=int, default=10)
    ap.add_argument("--reads", type=int, default=2000000)
    ap.add_argument("--threads", type=int, default=32, help="`kmcp index -j`: decides the block size")
    ap.add_argument("--block-size", type=int, default=0)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    genomes = synth.random_genomes(a.genomes, a.genome_len, seed=42)
    db = synth.make_db(os.path.join(a.out, "db"), genomes, k=21, n_chunks=a.chunks, overlap=150, threads=a.threads, block_size=a.block_size)
    print("db:", db)
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    G = np.stack([np.frombuffer(g, dtype=np.uint8) for g in genomes])
    n = a.reads
    gi = rng.integers(0, a.genomes, n)
    pos = rng.integers(0, a.genome_len - 150, n)
    idx = pos[:, None] + np.arange(150)[None, :]
    reads = G[gi[:, None], idx]
    sub = rng.random((n, 150)) < 0.01
    reads = np.where(sub, acgt[rng.integers(0, 4, (n, 150))], reads)
    rnd = rng.random(n) < 0.1
    reads[rnd] = acgt[rng.integers(0, 4, (int(rnd.sum()), 150))]
    qual = b"I" * 150
    with open(os.path.join(a.out, "reads.fq"), "wb") as fh:
        for i in range(n):
            fh.write(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + qual + b"\n")
    print("reads:", os.path.join(a.out, "reads.fq"), n)


if __name__ == "__main__":
    main()
