#!/usr/bin/env python3
"""Makes shim/testdata/: a tiny kmcp database (2 .uniki blocks, 12 columns), 48 reads and the TSV `kmcp search` prints for them
with default flags — the fixture of shim/kmcp_gpu_test.go (the Go-side test of the cgo binding a maintainer runs with
shim/build.sh) and of tests/test_shim_fixture_cpu.py / tests/test_gpu_cli.py, which keep it honest here: the CPU oracle must
reproduce expected.tsv from db/ + reads.fq, and kmcp-search on the GPU must print it.

  python tests/golden/make_shim_fixture.py        (deterministic: seeds below)"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests import synth  # noqa: E402
from tests.test_gpu_cli import oracle_tsv, write_fastq  # noqa: E402

HEADER = "#query\tqLen\tqKmers\tFPR\thits\ttarget\tchunkIdx\tchunks\ttLen\tkSize\tmKmers\tqCov\ttCov\tjacc\tqueryIdx"


def fixture(out):
    genomes = synth.random_genomes(6, 2400, seed=404)
    genomes[1] = genomes[0][:1200] + genomes[1][1200:]  # two references share half of their sequence: reads with two matches
    shutil.rmtree(os.path.join(out, "db"), ignore_errors=True)
    db_dir = synth.make_db(os.path.join(out, "db"), genomes, k=21, n_chunks=2, overlap=100, threads=2)  # 12 columns, blocks of 8
    reads = synth.sample_reads(genomes, 44, 150, sub_rate=0.01, seed=405, frac_random=0.15) + [genomes[2][:25], genomes[5][7:36], genomes[3][100:160], b"ACGT" * 40]  # two reads below -m 30
    ids = [f"read{i}" for i in range(len(reads))]
    odb = O.OracleDB(db_dir)
    rows, trailer = oracle_tsv(O, odb, ids, reads)
    odb.close()
    return db_dir, ids, reads, [HEADER] + rows + trailer


def main():
    out = os.path.join(ROOT, "shim", "testdata")
    os.makedirs(out, exist_ok=True)
    db_dir, ids, reads, lines = fixture(out)
    write_fastq(os.path.join(out, "reads.fq"), ids, reads)
    with open(os.path.join(out, "expected.tsv"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    n = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(out) for f in fs)
    print(f"{out}: {len(reads)} reads, {len(lines) - 4} rows, {n} bytes")


if __name__ == "__main__":
    main()
