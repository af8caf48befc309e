"""The reference's two golden tables (demo-searching/README.md:61-68, :102-109) through the GPU path:
whole-genome query -> K1 (FracMinHash / Closed Syncmer + scale) -> sort+unique -> K2 with 3 hash functions."""
import pytest

from tests.test_oracle_golden import build_demo_db, meta, read_query, sketches  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["minhash", "syncmer"])
def test_demo_searching_table_on_gpu(oracle_lib, meta, sketches, tmp_path, mode):  # noqa: F811
    from kmcp_amd import Database, default_params
    db_dir = build_demo_db(oracle_lib, tmp_path, meta, sketches, mode)
    q = read_query()
    with Database.open(db_dir, device=0) as db:
        res = db.search([q], params=default_params(min_qcov=0.5, sort_by=2))
        rows = [[db.col_info(int(m["col"]))[0], "%.4f" % m["qcov"], "%.4f" % m["tcov"], "%.4f" % m["jacc"]] for m in res.read(0)]
    assert rows == meta["tables"][mode]["rows"]
    assert int(res.qkmers[0]) == meta["genomes"]["NC_018658.1"][f"{mode}_kmers"]
    assert int(res.qlen[0]) == len(q)


def test_reads_from_a_real_genome(oracle_lib, tmp_path):  # noqa: F811
    """150-bp single-end and 2x150 paired reads sampled from the demo's E. coli genome (real repeats, rRNA operons, IS elements:
    duplicated k-mers inside reads and columns that share k-mers) against a 10-chunk k=21 index of it plus decoys."""
    import numpy as np

    from kmcp_amd import Database, default_params
    from tests import synth
    O = oracle_lib
    g = read_query()
    rng = np.random.default_rng(5)
    decoys = synth.random_genomes(3, 400000, seed=6)
    shuffled = np.frombuffer(g[:1000000], dtype=np.uint8).copy()
    rng.shuffle(shuffled)
    db_dir = synth.make_db(tmp_path, [g] + decoys + [shuffled.tobytes()], k=21, n_chunks=10, overlap=150, threads=8,
                           names=["NC_018658.1", "decoy1", "decoy2", "decoy3", "shuffled"])
    reads = synth.sample_reads([g], 3000, 150, sub_rate=0.01, seed=7, frac_random=0.05)
    # reads over the most repetitive stretches: positions whose 21-mer occurs again elsewhere in the genome
    a = np.frombuffer(g, dtype=np.uint8)
    h = O.nthash_all(g, 21)
    order = np.argsort(h, kind="stable")
    dup = order[1:][h[order][1:] == h[order][:-1]]
    for p in rng.choice(dup, 300):
        p = int(min(max(0, p - 60), len(g) - 150))
        reads.append(a[p:p + 150].tobytes())
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as db:
            res = db.search(reads, params=default_params())
            assert synth.assert_parity(odb, res, reads) > 3000
            r2 = synth.sample_reads([g], len(reads), 150, sub_rate=0.02, seed=8, frac_random=0.3)
            res = db.search(reads, r2, params=default_params(try_se=1))
            assert synth.assert_parity(odb, res, reads, r2, O.default_params(try_se=1)) > 2000
    finally:
        odb.close()
