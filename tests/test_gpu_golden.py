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
