"""K3 (kmcp_amd/csrc/k3_finalize.hip): the grouping by read, the -T filter and the per-query order of the hit list on the GPU, with
kmcpg_finalize_grouped expanding (column, count) pairs to Match records on the host — against the round-3 host half
(kmcpg_finalize on the raw hit list, KMCPG_DEVICE_FINALIZE=0) and against the oracle's order (Matches.Less / SortByTCov /
SortByJacc, util-db-search.go:105-145; --keep-top-scores :285-311).  Segment classes of the sort: one wave (2..512 matches), one
workgroup (513..4096), host (more)."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

MODES = (dict(), dict(sort_by=1), dict(sort_by=2), dict(do_not_sort=1), dict(top_n_scores=1), dict(top_n_scores=3), dict(sort_by=2, top_n_scores=2),
         dict(sort_by=1, top_n_scores=2), dict(min_tcov=0.02), dict(min_tcov=0.05, sort_by=1), dict(min_qcov=0.3, min_matched=5), dict(max_fpr=1e-12))


class host_finalize:
    """KMCPG_DEVICE_FINALIZE=0 for handles that make their first search inside the block"""

    def __enter__(self):
        self.old = os.environ.get("KMCPG_DEVICE_FINALIZE")
        os.environ["KMCPG_DEVICE_FINALIZE"] = "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("KMCPG_DEVICE_FINALIZE", None)
        else:
            os.environ["KMCPG_DEVICE_FINALIZE"] = self.old


def _same(a, b):
    for f in ("qlen", "qkmers", "ksize", "offs"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.matches.tobytes() == b.matches.tobytes()


def _family(O, tmp, n_cols, base_len, seed, extra_max, second=0):
    """n_cols columns that all hold the k-mers of one sequence (+ a different number of private k-mers each: different sizes, so
    tcov and jacc differ from column to column while mKmers ties abound), in blocks of 512 columns; `second` more columns hold a
    second sequence (reads from it have that many matches)"""
    rng = np.random.default_rng(seed)
    bases = synth.random_genomes(2, base_len, seed=seed)
    cfg = O.sketch_cfg(k=21)
    cols = []
    for fam, (base, ncol) in enumerate(zip(bases, (n_cols, second))):
        shared = O.sort_unique(O.generate_kmers(base, cfg))
        for c in range(ncol):
            drop = rng.random(len(shared)) < (0.0 if c % 7 == 0 else rng.uniform(0, 0.3))  # some relatives lack part of the sequence
            extra = rng.integers(1, 2**63, size=int(rng.integers(0, extra_max)), dtype=np.int64).astype(np.uint64)
            cols.append((f"f{fam}c{c}", base_len, c % 10, 10, O.sort_unique(np.concatenate([shared[~drop], extra]))))
    db_dir = O.build_db(str(tmp), cfg, cols, num_hashes=1, fpr=0.3, threads=8, block_size=512)
    reads = synth.sample_reads(bases if second else bases[:1], 300, 150, sub_rate=0.01, seed=seed + 1, frac_random=0.1)
    return db_dir, reads


def test_device_finalize_equals_host_finalize_and_the_oracle(oracle_lib, tmp_path):
    from kmcp_amd import Database, default_params
    O = oracle_lib
    db_dir, reads = _family(O, tmp_path / "db", 700, 4000, seed=31, extra_max=3000, second=90)  # 700 / 90 matches per read: workgroup and wave sorts
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as dev, Database.open(db_dir, device=0) as host:
            with host_finalize():
                host.search(reads[:2])
            for kw in MODES:
                a = dev.search(reads, params=default_params(**kw))
                with host_finalize():
                    b = host.search(reads, params=default_params(**kw))
                _same(a, b)
                n = synth.assert_parity(odb, a, reads, oparams=O.default_params(**kw))
                if not kw:
                    per_read = np.diff(a.offs.astype(np.int64))
                    assert per_read.max() > 512 and np.count_nonzero((per_read > 1) & (per_read <= 512)) > 20, (per_read.max(), n)
            # paired reads, --try-se retries (the retry lane runs K3 too)
            reads2 = synth.sample_reads(synth.random_genomes(2, 4000, seed=31), 300, 150, sub_rate=0.01, seed=77, frac_random=0.5)
            pe = dict(try_se=1, fpr_buf_size=499)
            a = dev.search(reads, reads2, params=default_params(**pe))
            with host_finalize():
                b = host.search(reads, reads2, params=default_params(**pe))
            _same(a, b)
            synth.assert_parity(odb, a, reads, reads2, O.default_params(**pe))
    finally:
        odb.close()


def test_segments_beyond_the_workgroup_sort_are_ordered_by_the_host(oracle_lib, tmp_path):
    from kmcp_amd import Database, default_params
    O = oracle_lib
    db_dir, reads = _family(O, tmp_path / "db", 4700, 1500, seed=41, extra_max=400)  # > 4096 matches per read
    reads = reads[:40]
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as dev, Database.open(db_dir, device=0) as host:
            for kw in (dict(), dict(sort_by=1), dict(sort_by=2, top_n_scores=2), dict(do_not_sort=1)):
                a = dev.search(reads, params=default_params(**kw))
                with host_finalize():
                    b = host.search(reads, params=default_params(**kw))
                _same(a, b)
                synth.assert_parity(odb, a, reads, oparams=O.default_params(**kw))
                if not kw:
                    assert np.diff(a.offs.astype(np.int64)).max() > 4096
    finally:
        odb.close()


def test_group_device_on_a_raw_hit_list(oracle_lib, tmp_path):
    """The two halves through the C ABI on device pointers, as a host that gathers the shards' hit lists on one GPU uses them:
    kmcpg_query_device -> kmcpg_group_device -> kmcpg_finalize_grouped == kmcpg_finalize on the same (shuffled) hit list; offsets,
    the bad-hit word, an empty batch and hits that name no read."""
    import torch
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    db_dir, reads = _family(O, tmp_path / "db", 300, 3000, seed=51, extra_max=2000)
    dev = torch.device("cuda:0")
    seqs, offs = lib.pack_reads(reads)
    n = len(reads)
    t_seqs = torch.from_numpy(seqs).to(dev)
    t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    cap = 400 * n
    with Database.open(db_dir, device=0) as db:
        for kw in (dict(), dict(sort_by=2), dict(min_tcov=0.03, sort_by=1), dict(do_not_sort=1)):
            p = default_params(**kw)
            hits = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            qk = torch.zeros(n, dtype=torch.int32, device=dev)
            ql = torch.zeros(n, dtype=torch.int32, device=dev)
            db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), n, int(offs[-1]), 150, hits.data_ptr(), cap, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(), params=p)
            torch.cuda.synchronize()
            m = int(cnt[0].item())
            assert 0 < m <= cap
            # any order in: shuffle the list on the device
            perm = torch.randperm(m, device=dev)
            hits[:m] = hits[:m][perm]
            pairs = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
            roffs = torch.full((n + 2,), -1, dtype=torch.int64, device=dev)
            db.group_device(hits.data_ptr(), cnt.data_ptr(), cap, qk.data_ptr(), n, pairs.data_ptr(), roffs.data_ptr(), params=p)
            torch.cuda.synchronize()
            ro = roffs.cpu().numpy().view(np.uint64)
            assert ro[0] == 0 and ro[n + 1] == 0 and np.all(np.diff(ro[:n + 1].astype(np.int64)) >= 0) and ro[n] <= m
            if not kw.get("min_tcov"):
                assert ro[n] == m
            a = db.finalize_grouped(pairs[:int(ro[n])].cpu().numpy().view(np.uint32), ro, qk.cpu().numpy(), ql.cpu().numpy(), params=p)
            b = db.finalize(hits[:m].cpu().numpy().view(np.uint32).reshape(-1).view(lib.HIT_DTYPE), qk.cpu().numpy(), ql.cpu().numpy(), params=p)
            _same(a, b)
            assert len(a.matches) > 50 * n // 2
        # hits that name a read / column that does not exist are counted, not followed
        bad = torch.tensor([[n + 5, 0, 50], [0, 10**6, 50], [1, 2, 60]], dtype=torch.int32, device=dev)
        cnt = torch.tensor([3, 0], dtype=torch.int64, device=dev)
        db.group_device(bad.data_ptr(), cnt.data_ptr(), 3, qk.data_ptr(), n, pairs.data_ptr(), roffs.data_ptr(), params=default_params())
        torch.cuda.synchronize()
        ro = roffs.cpu().numpy().view(np.uint64)
        assert ro[n + 1] == 2 and ro[n] == 1 and ro[1] == 0 and ro[2] == 1
        with pytest.raises(lib.KmcpGpuError):
            db.finalize_grouped(pairs[:1].cpu().numpy().view(np.uint32), ro, qk.cpu().numpy(), ql.cpu().numpy())
        # more hits produced than the buffer holds: only hit_cap are looked at
        cnt = torch.tensor([10, 0], dtype=torch.int64, device=dev)
        db.group_device(bad[2:].data_ptr(), cnt.data_ptr(), 1, qk.data_ptr(), n, pairs.data_ptr(), roffs.data_ptr(), params=default_params())
        torch.cuda.synchronize()
        assert int(roffs[n].item()) == 1 and int(roffs[n + 1].item()) == 0
        # an empty batch
        db.group_device(None, cnt.data_ptr(), 0, None, 0, None, roffs.data_ptr(), params=default_params())
        torch.cuda.synchronize()
        assert int(roffs[0].item()) == 0 and int(roffs[1].item()) == 0
        r = db.finalize_grouped(np.zeros((0, 2), np.uint32), np.zeros(2, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.int32))
        assert len(r) == 0 and len(r.matches) == 0
