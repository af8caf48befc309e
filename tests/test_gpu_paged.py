"""An index larger than the HBM at hand (reference: mmap / --low-mem, util-db-search.go:1238-1280, search.go:80):
kmcpg_open says what is needed and what is free (KMCPG_ENOMEM), kmcpg_open_paged searches every batch against one resident shard
after the other, and kmcp-search falls back to it by itself — same results as the resident database and the oracle.  The "small
GPU" is made with KMCPG_HBM_LIMIT_MB (caps what the library believes to be free)."""
import os
import subprocess

import numpy as np
import pytest

from tests import synth
from tests.test_gpu_cli import CLI, compare, oracle_tsv, run_cli, write_fastq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(oracle_lib, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("paged")
    # genomes of different lengths: every block gets a NumSigs of its own (equal ones would be laid side by side in one group)
    lens = np.random.default_rng(9).integers(16000, 32000, size=48)
    genomes = [g[:n] for g, n in zip(synth.random_genomes(48, 32000, seed=90), lens)]
    db_dir = synth.make_db(tmp / "db", genomes, k=21, n_chunks=2, overlap=150, threads=12)  # 96 columns in blocks of 8
    odb = oracle_lib.OracleDB(db_dir)
    yield tmp, genomes, db_dir, odb
    odb.close()


class small_gpu:
    def __init__(self, mb):
        self.env = {"KMCPG_HBM_LIMIT_MB": str(mb), "KMCPG_WORKSPACE_RESERVE_MB": "0"}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_open_reports_needed_and_free_then_paged_search_equals_resident(oracle_lib, case):
    from kmcp_amd import Database, default_params
    from kmcp_amd.lib import KmcpGpuError
    O = oracle_lib
    tmp, genomes, db_dir, odb = case
    with Database.open(db_dir, device=0) as db:
        assert db.paged_info() == (0, 0)
        bi = [db.block_info(b) for b in range(db.info.n_blocks)]
        assert len({b["num_sigs"] for b in bi}) == len(bi)  # nothing grouped: the sum below is what is resident
        need_mb = sum(b["num_sigs"] * b["stride"] for b in bi) / 2**20
    assert need_mb > 4
    reads = synth.sample_reads(genomes, 3000, 150, sub_rate=0.01, seed=91, frac_random=0.1)
    reads2 = synth.sample_reads(genomes, 3000, 150, sub_rate=0.01, seed=92, frac_random=0.6)
    with small_gpu(int(need_mb * 0.45)):
        with pytest.raises(KmcpGpuError) as e:
            Database.open(db_dir, device=0)
        assert e.value.code == -5 and "does not fit in HBM" in str(e.value) and "GB free on device 0" in str(e.value) and "needs" in str(e.value)
        with pytest.raises(KmcpGpuError) as e:
            Database.open_paged(db_dir, device=0, passes=2)  # 2 shards of 0.5 x need do not fit 0.45 x need
        assert e.value.code == -5 and "at least" in str(e.value)
        with Database.open_paged(db_dir, device=0) as db:
            passes, uploads = db.paged_info()
            assert 3 <= passes <= 6 and uploads == 0
            res = db.search(reads, params=default_params())
            assert db.paged_info() == (passes, passes)
            assert synth.assert_parity(odb, res, reads) > 2000
            # the shard searched last is still resident: the next batch uploads passes - 1
            p = default_params(try_se=1, fpr_buf_size=499)
            res2 = db.search(reads, reads2, params=p)
            n_up = db.paged_info()[1]
            assert n_up >= 2 * passes - 1  # + the retries of --try-se, each a round of passes - 1 uploads
            assert synth.assert_parity(odb, res2, reads, reads2, O.default_params(try_se=1, fpr_buf_size=499)) > 1000
            # asynchronous spelling: the search happens inside kmcpg_submit on a paged handle
            seqs, offs = __import__("kmcp_amd").lib.pack_reads(reads[:500])
            t = db.submit(seqs, offs, params=default_params())
            res3 = db.wait(t)
            assert synth.assert_parity(odb, res3, reads[:500]) > 300
    # an explicit number of passes on a GPU where everything fits
    with Database.open_paged(db_dir, device=0, passes=5) as db:
        assert db.paged_info()[0] == 5
        res = db.search(reads[:800], params=default_params())
        assert synth.assert_parity(odb, res, reads[:800]) > 500
    with Database.open_paged(db_dir, device=0) as db:  # fits: an ordinary resident handle
        assert db.paged_info() == (0, 0)


def test_cli_falls_back_to_passes(oracle_lib, case):
    O = oracle_lib
    tmp, genomes, db_dir, odb = case
    reads = synth.sample_reads(genomes, 2500, 150, sub_rate=0.02, seed=93, frac_random=0.1)
    ids = [f"r{i}" for i in range(len(reads))]
    fq = str(tmp / "r.fq")
    write_fastq(fq, ids, reads)
    want, trailer = oracle_tsv(O, odb, ids, reads)
    db_root = os.path.dirname(db_dir)
    one = run_cli(["-d", db_root, fq], str(tmp / "one.tsv"))
    compare(one, want, trailer)
    from kmcp_amd import Database
    with Database.open(db_dir, device=0) as db:
        need_mb = sum(db.block_info(b)["num_sigs"] * db.block_info(b)["stride"] for b in range(db.info.n_blocks)) / 2**20
    env = dict(os.environ, KMCPG_HBM_LIMIT_MB=str(max(1, int(need_mb * 0.6))), KMCPG_WORKSPACE_RESERVE_MB="0")
    r = subprocess.run([CLI, "-d", db_root, fq, "-o", str(tmp / "auto.tsv"), "--gpu-batch", "700"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "does not fit in HBM" in r.stderr and "passes per batch" in r.stderr
    assert open(tmp / "auto.tsv").read().split("\n") == one
    # round 6: the reader runs before the database is open, so a paged index meets batches that were cut for a resident one (here ~40 of
    # ~60 reads each, through the multi-threaded parser): the searcher joins them into the large batches a paged index wants (Batch::append)
    env2 = dict(env, KMCP_PARALLEL_MIN_BYTES="1", KMCP_READER_CHUNK="20000")
    r = subprocess.run([CLI, "-d", db_root, fq, "-o", str(tmp / "auto2.tsv")], capture_output=True, text=True, timeout=300, env=env2)
    assert r.returncode == 0, r.stderr
    assert "passes per batch" in r.stderr
    assert open(tmp / "auto2.tsv").read().split("\n") == one
    r = subprocess.run([CLI, "-d", db_root, fq, "-o", str(tmp / "p3.tsv"), "--gpu-passes", "3", "-q"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(tmp / "p3.tsv").read().split("\n") == one


def test_a_batch_whose_workspace_does_not_fit_is_searched_in_halves(oracle_lib, case):
    """a batch whose device workspace cannot be allocated is not an error of kmcpg_search_batch — it is
    searched as two halves (recursively) and answered as one result; kmcpg_batch_hint tells a host how large a batch may be.
    KMCPG_TEST_MAX_BASES makes kmcpg_query_device refuse larger batches with KMCPG_ENOMEM, as a failed hipMalloc would."""
    from kmcp_amd import Database, default_params
    O = oracle_lib
    tmp, genomes, db_dir, odb = case
    reads = synth.sample_reads(genomes, 1500, 150, sub_rate=0.01, seed=95, frac_random=0.1)
    reads2 = synth.sample_reads(genomes, 1500, 150, sub_rate=0.01, seed=96, frac_random=0.5)
    with Database.open(db_dir, device=0) as db:
        hint = db.batch_hint()
        assert hint > 1 << 20  # an (almost) empty MI355X: gigabases
        whole = db.search(reads, params=default_params())
        whole_pe = db.search(reads, reads2, params=default_params(try_se=1, fpr_buf_size=499))
        os.environ["KMCPG_TEST_MAX_BASES"] = str(150 * 200)  # 1500 reads -> 8 pieces; pairs -> 16
        try:
            res = db.search(reads, params=default_params())
            res_pe = db.search(reads, reads2, params=default_params(try_se=1, fpr_buf_size=499))
        finally:
            os.environ.pop("KMCPG_TEST_MAX_BASES")
    for a, b in ((whole, res), (whole_pe, res_pe)):
        assert np.array_equal(a.qlen, b.qlen) and np.array_equal(a.qkmers, b.qkmers) and np.array_equal(a.offs, b.offs)
        assert np.array_equal(a.ksize, b.ksize) and a.matches.tobytes() == b.matches.tobytes()
    assert synth.assert_parity(odb, res, reads) > 1000
    assert synth.assert_parity(odb, res_pe, reads, reads2, O.default_params(try_se=1, fpr_buf_size=499)) > 500
    with small_gpu(64):
        with Database.open_paged(db_dir, device=0, passes=4) as db:
            assert 0 < db.batch_hint() <= (64 << 20) // 40 + 1
