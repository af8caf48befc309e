"""BASELINE.json configs[0]: the demo-profiling database — `kmcp compute -k 21 --split-number 10 --split-overlap 150
--seq-name-filter plasmid` + `kmcp index -n 1 -f 0.3` over the reference's 15 demo genomes, searched with single-end
150-bp reads given as two files (demo-profiling/README.md:232-275: "paired information are not used").

The genomes are the reference's own data (demo-profiling/refs/*.fa.gz), cut to their first 300 kb by
tests/golden/make_demo_profiling.py; the reads of the demo (mock_1/2.fastq.gz) are not in the reference tree, so reads are
sampled from the genomes (seed 1, 2 % substitutions, both strands) as SURVEY.md §8d "Config 0" says.  Real genomes bring
what random sequences do not: several contigs per file joined by k-1 N's, plasmid records to filter, close relatives
(E. coli and two Shigella) whose chunks tie for a read.

Checked: kmcp-search's TSV (15 columns + the trailer `kmcp profile` parses) equals the oracle's line for line; the
C-ABI batch search agrees tuple for tuple; the GPU index builder writes the oracle's .uniki files byte for byte."""
import filecmp
import gzip
import os

import numpy as np
import pytest

from tests import synth
from tests.test_gpu_cli import compare, oracle_tsv, run_cli, write_fastq

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "demo_profiling_refs_300k.fa.gz")
K, SPLIT, OVERLAP = 21, 10, 150


def load_genomes():
    """{accession: [(record name, sequence bytes), ...]} in file order."""
    out, name, seq = {}, None, []

    def flush():
        if name is not None:
            acc, rec = name.split("|", 1)
            out.setdefault(acc, []).append((rec, "".join(seq).encode()))

    with gzip.open(FIXTURE, "rt") as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith(">"):
                flush()
                name, seq = line[1:], []
            else:
                seq.append(line)
    flush()
    return out


def compute_columns(O, genomes):
    """`kmcp compute` in --split-number mode (compute.go:571-627): records not matching --seq-name-filter are concatenated with
    kMax-1 N's, genome size = length of that sequence, then 10 windows with 150 bases of overlap (synth.split_chunks)."""
    cfg = O.sketch_cfg(k=K)
    cols, big = [], {}
    for acc in sorted(genomes):
        kept = [s for n, s in genomes[acc] if "plasmid" not in n]
        assert kept, acc
        seq = (b"N" * (K - 1)).join(kept)
        big[acc] = seq
        chunks = synth.split_chunks(seq, SPLIT, OVERLAP)
        assert len(chunks) == SPLIT
        for ci, c in enumerate(chunks):
            cols.append((acc, len(seq), ci, SPLIT, O.sort_unique(O.generate_kmers(c, cfg))))
    return cols, big


@pytest.fixture(scope="module")
def demo(oracle_lib, tmp_path_factory):
    O = oracle_lib
    tmp = tmp_path_factory.mktemp("config0")
    genomes = load_genomes()
    assert len(genomes) == 15 and sum(len(v) for v in genomes.values()) == 23
    cols, big = compute_columns(O, genomes)
    db_dir = O.build_db(str(tmp / "refs-k21-n10.kmcp"), O.sketch_cfg(k=K), cols, num_hashes=1, fpr=0.3, threads=16)  # kmcp index -j 16 -n 1 -f 0.3
    # reads: SURVEY.md §8d Config 0
    rng = np.random.default_rng(1)
    accs = sorted(big)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    reads = []
    for i in range(12000):
        g = big[accs[int(rng.integers(0, len(accs)))]]
        p = int(rng.integers(0, len(g) - 150))
        r = np.frombuffer(g[p:p + 150], dtype=np.uint8).copy()
        m = rng.random(150) < 0.02
        r[m] = acgt[rng.integers(0, 4, int(m.sum()))]
        b = r.tobytes()
        if rng.random() < 0.5:
            b = b.translate(comp)[::-1]
        reads.append(b)
    reads += [bytes(acgt[rng.integers(0, 4, 150)]) for _ in range(600)]  # reads of something that is not in the database
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    return dict(tmp=tmp, cols=cols, db_dir=db_dir, reads=reads)


def test_config0_cli_matches_oracle_tsv(demo, oracle_lib):
    O = oracle_lib
    tmp, db_dir, reads = demo["tmp"], demo["db_dir"], demo["reads"]
    half = len(reads) // 2
    ids = [f"mock{i}/1" for i in range(half)] + [f"mock{i}/2" for i in range(len(reads) - half)]
    f1, f2 = str(tmp / "mock_1.fastq.gz"), str(tmp / "mock_2.fastq.gz")
    write_fastq(f1, ids[:half], reads[:half], gz=True)
    write_fastq(f2, ids[half:], reads[half:], gz=True)
    odb = O.OracleDB(db_dir)
    assert odb.nblocks == 10 and odb.ncols == 150
    want, trailer = oracle_tsv(O, odb, ids, reads)
    matched = int(trailer[1].split(": ")[1])
    assert 0.8 * len(reads) < matched < 0.97 * len(reads)  # the demo log reports 88.5 % on its own reads
    hits_per_read = len(want) / matched
    assert hits_per_read > 1.05  # close relatives share k-mers: reads with several hits exist
    got = run_cli(["-d", os.path.dirname(db_dir), f1, f2], str(tmp / "mock.kmcp.gz"))
    compare(got, want, trailer)
    # the same with -K (unmatched rows kept), small GPU batches and two lanes' worth of searcher threads
    want_k, trailer_k = oracle_tsv(O, odb, ids, reads, keep_unmatched=True)
    compare(run_cli(["-d", os.path.dirname(db_dir), f1, f2, "-K", "--gpu-batch", "1000"], str(tmp / "mock.K.tsv")), want_k, trailer_k)
    odb.close()


def test_config0_batch_api_parity(demo, oracle_lib):
    from kmcp_amd import Database, default_params
    O = oracle_lib
    odb = O.OracleDB(demo["db_dir"])
    with Database.open(demo["db_dir"]) as db:
        assert db.info.k == K and db.info.n_blocks == 10 and int(db.info.n_cols) == 150
        res = db.search(demo["reads"], params=default_params())
        assert synth.assert_parity(odb, res, demo["reads"]) > 12000
        # sorted by target coverage / jaccard with the top score kept, as `kmcp profile` users do
        for sb in (1, 2):
            res = db.search(demo["reads"][:3000], params=default_params(sort_by=sb, top_n_scores=1, min_qcov=0.4))
            synth.assert_parity(odb, res, demo["reads"][:3000], oparams=O.default_params(sort_by=sb, top_n_scores=1, min_qcov=0.4))
    odb.close()


def test_config0_gpu_index_builder_writes_the_same_files(demo, oracle_lib, tmp_path):
    from kmcp_amd import lib
    out = lib.build_db(str(tmp_path / "gpu.kmcp"), demo["cols"], k=K, num_hashes=1, fpr=0.3, threads=16)
    files = sorted(f for f in os.listdir(demo["db_dir"]) if f.endswith(".uniki"))
    assert len(files) == 10 and files == sorted(f for f in os.listdir(out) if f.endswith(".uniki"))
    for f in files:
        assert filecmp.cmp(os.path.join(demo["db_dir"], f), os.path.join(out, f), shallow=False), f
