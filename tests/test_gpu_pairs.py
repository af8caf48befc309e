"""Compact results (kmcpg_search_batch_pairs / kmcpg_wait_pairs + kmcpg_expand_pairs, round 5): the final matches of every query as
(column, mKmers) pairs, expanded by the caller - must be, bit for bit and in order, what kmcpg_search_batch returns as records, on
every path a batch can take: plain, every sort mode, -T, -f, --keep-top-scores, pieces of a large batch, paired reads with
--try-se and multi-k databases (which collect records and are converted), submit / wait, the oracle as the referee."""
import re

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _expanded(db, pr):
    parts = [db.expand_pairs(int(pr.qkmers[i]), pr.read(i)) for i in range(len(pr)) if pr.offs[i + 1] > pr.offs[i]]
    from kmcp_amd.lib import MATCH_DTYPE
    return np.concatenate(parts) if parts else np.zeros(0, dtype=MATCH_DTYPE)


def _same(db, rec, pr):
    for f in ("qlen", "qkmers", "ksize", "offs"):
        assert np.array_equal(getattr(rec, f), getattr(pr, f)), f
    assert rec.k == pr.k
    assert _expanded(db, pr).tobytes() == rec.matches.tobytes()


@pytest.fixture(scope="module")
def family(oracle_lib, tmp_path_factory):
    """30 strains of one species + a few others: dozens of matches per read"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import family_db
    tmp = tmp_path_factory.mktemp("pairs")
    info, reads = family_db.build(str(tmp / "db"), ecoli_strains=30, small_strains=2, n_reads=6000, threads=8)
    return info["db_dir"], [reads[i].tobytes() for i in range(reads.shape[0])]


def test_pairs_equal_records_on_match_heavy_data(family, oracle_lib):
    from kmcp_amd import Database, default_params, lib
    db_dir, reads = family
    O = oracle_lib
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir) as db:
            for kw in (dict(), dict(sort_by=1), dict(sort_by=2), dict(do_not_sort=1), dict(top_n_scores=2), dict(sort_by=2, top_n_scores=1),
                       dict(min_tcov=0.0002), dict(min_qcov=0.3, min_matched=5), dict(max_fpr=1e-12)):
                p = default_params(**kw)
                rec = db.search(reads[:1500], params=p)
                pr = db.search_pairs(reads[:1500], params=p)
                _same(db, rec, pr)
                if not kw:
                    assert len(rec.matches) > 10 * 1500
                    synth.assert_parity(odb, rec, reads[:300])
            # submit / wait_pairs, several in flight; the caller's buffers are the caller's again at once
            seqs, offs = lib.pack_reads(reads[:2000])
            t1 = db.submit(seqs, offs, params=default_params())
            t2 = db.submit(seqs, offs, params=default_params(sort_by=2))
            r2 = db.wait_pairs(t2)
            r1 = db.wait_pairs(t1)
            _same(db, db.search(reads[:2000], params=default_params()), r1)
            _same(db, db.search(reads[:2000], params=default_params(sort_by=2)), r2)
    finally:
        odb.close()


def test_pairs_through_the_pieces_of_a_large_batch(family, monkeypatch):
    """a batch large enough to be cut into pieces (KMCPG_PIECE_MIN lowers the bar): every piece lands its pairs in the one result"""
    from kmcp_amd import Database, default_params
    db_dir, reads = family
    monkeypatch.setenv("KMCPG_PIECE_MIN", "500")
    with Database.open(db_dir) as db:
        db.search(reads[:3000], params=default_params())  # the first large batch of a handle goes through whole
        rec = db.search(reads, params=default_params())
        pr = db.search_pairs(reads, params=default_params())
        _same(db, rec, pr)
        pr2 = db.search_pairs(reads, params=default_params(top_n_scores=1))
        _same(db, db.search(reads, params=default_params(top_n_scores=1)), pr2)
        assert len(pr2.pairs) < len(pr.pairs)


def test_pairs_on_the_paths_that_collect_records(oracle_lib, tmp_path):
    """--try-se on pairs and the smaller k of a multi-k database splice sub-results record by record: such batches are converted"""
    from kmcp_amd import Database, default_params
    O = oracle_lib
    genomes = synth.random_genomes(12, 8000, seed=277)
    cols = []
    for gi, g in enumerate(genomes):
        h = np.concatenate([O.generate_kmers(g, O.sketch_cfg(k=k)) for k in (21, 31)])
        cols.append((f"g{gi}", len(g), 0, 1, O.sort_unique(h)))
    db_dir = O.build_db(str(tmp_path / "db"), O.sketch_cfg(k=31), cols, num_hashes=1, fpr=0.1, threads=4, block_size=8)
    yml = open(db_dir + "/__db.yml").read()
    open(db_dir + "/__db.yml", "w").write(re.sub(r"ks:\n- 31\n", "ks:\n- 21\n- 31\n", yml))
    rng = np.random.default_rng(3)
    reads, reads2 = [], []
    for i in range(300):
        g = genomes[i % len(genomes)]
        pos = int(rng.integers(0, len(g) - 150))
        r = bytearray(g[pos:pos + 150])
        if i % 3 == 2:
            for j in range(5, 150, 28):
                r[j] = ord("A") if r[j] != ord("A") else ord("C")
        reads.append(bytes(r))
        reads2.append(genomes[(i + 5) % len(genomes)][40:190] if i % 4 else bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)))
    with Database.open(db_dir) as db:
        assert db.ks == [31, 21]
        p = default_params(min_qcov=0.2)
        rec = db.search(reads, params=p)
        _same(db, rec, db.search_pairs(reads, params=p))
        assert set(rec.ksize.tolist()) == {21, 31}
        pe = default_params(min_qcov=0.4, try_se=1, fpr_buf_size=499)
        rec2 = db.search(reads, reads2, params=pe)
        _same(db, rec2, db.search_pairs(reads, reads2, params=pe))
        assert len(rec2.matches) > 50
        # empty batch
        e = db.search_pairs([], params=p)
        assert len(e) == 0 and len(e.pairs) == 0


def test_pairs_when_short_queries_take_the_chunked_kernel(family, oracle_lib, monkeypatch):
    """KMCPG_SPLIT_MIN below the read length sends 130-k-mer queries through k2_cobs<SPLIT> + k_threshold_long.  The compact path takes a
    short query's segment as final only when the query call reported that the -f bound covered it (Lane::bound_n) - and the chunked
    form applies the bound too: with -t just above the database's FPR the bound is the stricter threshold, so a path that skipped it
    would print matches that fail -f (finalize.cpp trusted path)."""
    from kmcp_amd import Database, default_params
    db_dir, reads = family
    O = oracle_lib
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir) as db:
            fpr = db.info.fpr
            for kw in (dict(min_qcov=fpr + 0.02, min_matched=1), dict(min_qcov=fpr + 0.02, min_matched=1, max_fpr=1e-6), dict()):
                p = default_params(**kw)
                monkeypatch.delenv("KMCPG_SPLIT_MIN", raising=False)
                want = db.search(reads[:400], params=p)
                monkeypatch.setenv("KMCPG_SPLIT_MIN", "50")
                rec = db.search(reads[:400], params=p)
                pr = db.search_pairs(reads[:400], params=p)
                assert rec.matches.tobytes() == want.matches.tobytes()
                _same(db, rec, pr)
                # ... and with the bound switched off for the call, the host half applies -f itself
                monkeypatch.setenv("KMCPG_FPR_BOUND", "0")
                _same(db, want, db.search_pairs(reads[:400], params=p))
                monkeypatch.delenv("KMCPG_FPR_BOUND")
            synth.assert_parity(odb, want, reads[:100])
    finally:
        odb.close()
