"""A real-genome, hit-heavy database (tools/family_db.py: the reference's demo genomes + mutated family members, 10 chunks per
strain, `-j 32` block rules): what synthetic i.i.d. indexes do not have — uneven column densities (a block's filter is sized for
its fullest column, index.go:936-946), relatives that share sectors, tens to hundreds of hits per read.  Reduced size here
(a few hundred columns); test_full_size_sample builds the 21 000-column / 1.3 GB one itself and checks a sample of reads against
the oracle (tools/bench_real_families.py measures on the same database)."""
import json
import os
import sys

import numpy as np
import pytest

from tests import synth
from tests.test_gpu_cli import compare, oracle_tsv, run_cli

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _reads_list(reads):
    return [reads[i].tobytes() for i in range(reads.shape[0])]


@pytest.mark.parametrize("uniform_sigs", [0, 1])
def test_family_database_reduced(oracle_lib, tmp_path, uniform_sigs):
    import family_db
    from kmcp_amd import Database, default_params
    O = oracle_lib
    info, reads = family_db.build(str(tmp_path / "db"), ecoli_strains=6, small_strains=3, n_reads=3000, uniform_sigs=uniform_sigs, threads=8)
    assert info["columns"] == (6 + 15 * 3) * 10 and info["kmers_per_col_max"] > 10 * info["kmers_per_col_min"]  # uneven columns
    rl = _reads_list(reads)
    odb = O.OracleDB(info["db_dir"])
    try:
        # the GPU builder's columns = the oracle's restatement of `kmcp compute` on the same strain sequences: spot-check one strain
        with Database.open(info["db_dir"], device=0) as db:
            res = db.search(rl, params=default_params())
            n = synth.assert_parity(odb, res, rl)
            hits_per_read = n / len(rl)
            assert hits_per_read > 2, hits_per_read  # families: a read matches several strains (and overlapping chunks)
            if uniform_sigs:
                assert len({db.block_info(b)["num_sigs"] for b in range(db.info.n_blocks)}) < db.info.n_blocks
        # end to end: the TSV with many rows per read, line for line
        fq = str(tmp_path / "r.fq")
        family_db.write_fastq(fq, reads[:1500])
        ids = [f"r{i}" for i in range(1500)]
        want, trailer = oracle_tsv(O, odb, ids, rl[:1500])
        compare(run_cli(["-d", os.path.dirname(info["db_dir"]), fq], str(tmp_path / "o.tsv")), want, trailer)
        # a kept-top-scores run (what `kmcp profile` users add for families): -n 2
        want, trailer = oracle_tsv(O, odb, ids, rl[:1500], params=O.default_params(top_n_scores=2))
        compare(run_cli(["-d", os.path.dirname(info["db_dir"]), fq, "-n", "2"], str(tmp_path / "o2.tsv")), want, trailer)
    finally:
        odb.close()


def test_many_matches_every_sort_mode(oracle_lib, tmp_path):
    """Dozens of matches per read (30 strains of one species): kmcpg_finalize's paths for reads with many matches — the counting
    sort by mKmers of `-s qcov`, the index sorts of `-s tcov` / `-s jacc` / `-S`, `-n`, `-T` — give the oracle's matches in the
    oracle's order."""
    import family_db
    from kmcp_amd import Database, default_params
    O = oracle_lib
    info, reads = family_db.build(str(tmp_path / "db"), ecoli_strains=30, small_strains=1, n_reads=1200, threads=8)
    rl = _reads_list(reads)
    odb = O.OracleDB(info["db_dir"])
    try:
        with Database.open(info["db_dir"], device=0) as db:
            for kw in (dict(), dict(sort_by=1), dict(sort_by=2), dict(do_not_sort=1), dict(top_n_scores=3), dict(sort_by=2, top_n_scores=2),
                       dict(min_tcov=0.0002), dict(min_qcov=0.3, min_matched=5)):
                res = db.search(rl, params=default_params(**kw))
                n = synth.assert_parity(odb, res, rl, oparams=O.default_params(**kw))
                if not kw:
                    per_read = np.diff(np.asarray(res.offs, dtype=np.int64))
                    assert per_read.max() > 30 and np.count_nonzero(per_read > 8) > len(rl) // 4, (per_read.max(), n)
    finally:
        odb.close()


def test_strain_columns_equal_the_oracles_compute(oracle_lib):
    """The generator's columns (K1 on the GPU + torch sort/unique) are what the oracle's `kmcp compute` restatement yields for the
    same sequence: k-mers of 10 overlapping chunks of a real genome with N's and several records."""
    import torch

    import family_db
    O = oracle_lib
    B = family_db.Builder(0, seed=1)
    try:
        acc, seq = family_db.base_genomes()[4]  # Enterococcus faecalis: two records joined by k-1 N's
        base = torch.frombuffer(bytearray(seq), dtype=torch.uint8).to(B.dev)
        s = B.strain(base, 4, 3)
        B.add_genome("x", s)
        hs = s.cpu().numpy().tobytes()
        cfg = O.sketch_cfg(k=family_db.K)
        bounds = family_db.split_bounds(len(hs))
        for (a, b), col in zip(bounds, B.columns):
            want = O.sort_unique(O.generate_kmers(hs[a:b], cfg))
            assert np.array_equal(col[4], want)
        assert [c for c in synth.split_chunks(hs, 10, 150)] == [hs[a:b] for a, b in bounds]
    finally:
        B.close()


def test_full_size_sample(oracle_lib, tmp_path_factory):
    """Parity against the oracle on a sample of reads at FULL size: 21 000 real-genome columns (600 E. coli strains + 15 species x
    100 strains, 10 chunks each; 1.3 GB of index, 28 distinct NumSigs, ~200 matches per read).  The test builds the database itself
    (tools/family_db.py: ~7 s of strain generation + ~6 s of kmcpg_build_db on the GPU box), for the reference's block layout (and
    for uniform_sigs = 1 under KMCP_FAMILY_BOTH=1); KMCP_FAMILY_DB may point at one tools/bench_real_families.py left behind instead.  Skips only when the
    box lacks the room (HBM, scratch disk)."""
    import shutil

    import torch

    import family_db
    from kmcp_amd import Database, default_params
    O = oracle_lib
    root = os.environ.get("KMCP_FAMILY_DB")
    if root and os.path.exists(os.path.join(root, "family_db.json")):
        info = json.load(open(os.path.join(root, "family_db.json")))
        reads = []
        with open(os.path.join(root, "reads.fq"), "rb") as fh:
            for i, line in enumerate(fh):
                if i % 4 == 1:
                    reads.append(line.rstrip(b"\n"))
                if len(reads) == 400:
                    break
        db_dirs = info["db_dirs"]
    else:
        free_b, _ = torch.cuda.mem_get_info(0)
        tmp = tmp_path_factory.mktemp("family_full")
        if free_b < 40e9:
            pytest.skip("needs 40 GB of free HBM (strain columns + builder + 1.3 GB of index)")
        if shutil.disk_usage(str(tmp)).free < 6e9:
            pytest.skip("needs 6 GB of scratch disk for two 1.3 GB databases")
        cols, reads_a, info = family_db.generate(600, 100, 20000)
        assert info["columns"] == 21000
        from kmcp_amd import lib
        # the reference's block layout always; uniform_sigs = 1 as well under KMCP_FAMILY_BOTH=1 (the reduced test above covers both)
        modes = (0, 1) if os.environ.get("KMCP_FAMILY_BOTH") else (0,)
        db_dirs = {str(m): lib.build_db(str(tmp / f"u{m}"), cols, k=family_db.K, threads=32, uniform_sigs=m, alias="family-db") for m in modes}
        del cols
        reads = _reads_list(reads_a[:400])
    for mode, db_dir in db_dirs.items():
        odb = O.OracleDB(db_dir)
        try:
            with Database.open(db_dir, device=0) as db:
                res = db.search(reads, params=default_params())
                assert int(db.info.n_cols) == 21000
            n = synth.assert_parity(odb, res, reads)
            assert n > 20 * len(reads)  # families: dozens to hundreds of matches per planted read
            print(f"full-size family database, uniform_sigs={mode}: {len(reads)} reads, {n} (read, column) tuples identical to the oracle")
        finally:
            odb.close()
