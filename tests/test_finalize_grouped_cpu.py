"""kmcpg_finalize_grouped (the host half behind K3) on a metadata-only handle — no GPU: fed with the (column, count) pairs of
kmcpg_finalize's own output (grouped by read, in final order, -T applied: what K3 hands over) it must reproduce that output bit for bit
for every sort mode, --keep-top-scores, -T and -f; a segment longer than K3 orders (> 4096 pairs) arrives unordered and is sorted here;
offsets that run backwards, columns that do not exist and a non-zero bad-hit word are refused."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "shim", "testdata", "db", "R001")


def _pairs_of(res):
    """what K3 would hand over for the matches of `res`: pairs in result order + CSR offsets (+ the bad-hit word)"""
    pairs = np.stack([res.matches["col"].astype(np.uint32), res.matches["mkmers"].astype(np.uint32)], axis=1) if len(res.matches) else np.zeros((0, 2), np.uint32)
    return np.ascontiguousarray(pairs), np.concatenate([res.offs.astype(np.uint64), [np.uint64(0)]])


@pytest.mark.parametrize("kw", [dict(), dict(sort_by=1), dict(sort_by=2), dict(do_not_sort=1), dict(top_n_scores=1), dict(sort_by=2, top_n_scores=2),
                                dict(min_tcov=0.09), dict(max_fpr=1e-14), dict(min_qcov=0.3, min_matched=5)])
def test_grouped_input_reproduces_the_host_half(kw):
    from kmcp_amd import Database, default_params, lib
    rng = np.random.default_rng(11)
    n_reads, per = 3000, 9
    with Database.open(DB, device=-1) as db:
        ncols = int(db.info.n_cols)
        hits = np.empty(n_reads * per, dtype=lib.HIT_DTYPE)
        hits["read"] = np.repeat(np.arange(n_reads, dtype=np.uint32), per)
        hits["col"] = np.concatenate([rng.permutation(ncols)[:per] for _ in range(n_reads)]).astype(np.uint32)
        hits["count"] = rng.integers(60, 131, size=n_reads * per).astype(np.uint32)
        hits = hits[rng.permutation(len(hits))]  # any order in
        qk = rng.integers(120, 131, size=n_reads).astype(np.int32)
        hits["count"] = np.minimum(hits["count"], qk[hits["read"]].astype(np.uint32))
        ql = (qk + 20).astype(np.int32)
        p = default_params(**kw)
        want = db.finalize(hits, qk, ql, params=p)
        # K3 applies -T and the order but not --keep-top-scores / -f: hand over the untruncated, FPR-unfiltered list in final order
        q = default_params(**{k: v for k, v in kw.items() if k not in ("top_n_scores", "max_fpr")})
        if "max_fpr" in kw:
            q.max_fpr = 1.0
        full = db.finalize(hits, qk, ql, params=q)
        pairs, roffs = _pairs_of(full)
        got = db.finalize_grouped(pairs, roffs, qk, ql, params=p)
        for f in ("qlen", "qkmers", "ksize", "offs"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), f
        assert got.matches.tobytes() == want.matches.tobytes()
        assert len(want.matches) > 1000 or "max_fpr" in kw


def test_long_segments_are_sorted_here_and_bad_input_is_refused():
    from kmcp_amd import Database, default_params, lib
    rng = np.random.default_rng(12)
    with Database.open(DB, device=-1) as db:
        ncols = int(db.info.n_cols)
        m = 5000  # > K3_WG_CAP: arrives grouped but unordered
        hits = np.empty(m + 3, dtype=lib.HIT_DTYPE)
        hits["read"] = np.concatenate([np.zeros(m, np.uint32), np.ones(3, np.uint32)])
        hits["col"] = rng.integers(0, ncols, size=m + 3).astype(np.uint32)
        hits["count"] = rng.integers(80, 131, size=m + 3).astype(np.uint32)
        qk = np.array([130, 130], np.int32)
        ql = np.array([150, 150], np.int32)
        for kw in (dict(), dict(sort_by=1), dict(sort_by=2, top_n_scores=3), dict(do_not_sort=1)):
            p = default_params(**kw)
            want = db.finalize(hits, qk, ql, params=p)
            pairs = np.stack([hits["col"], hits["count"]], axis=1).astype(np.uint32)
            # read 1's three pairs in final order (short segments arrive ordered), read 0's in any order
            tail = db.finalize(hits[m:], qk, ql, params=default_params(**{k: v for k, v in kw.items() if k != "top_n_scores"}))
            pairs[m:] = np.stack([tail.matches["col"], tail.matches["mkmers"]], axis=1).astype(np.uint32)
            roffs = np.array([0, m, m + 3, 0], np.uint64)
            got = db.finalize_grouped(pairs, roffs, qk, ql, params=p)
            assert np.array_equal(got.offs, want.offs) and got.matches.tobytes() == want.matches.tobytes(), kw
        good = np.array([[0, 100]], np.uint32)
        with pytest.raises(lib.KmcpGpuError):
            db.finalize_grouped(good, np.array([0, 1, 1, 5], np.uint64), qk, ql)  # the bad-hit word
        with pytest.raises(lib.KmcpGpuError):
            db.finalize_grouped(np.array([[ncols + 7, 100]], np.uint32), np.array([0, 1, 1, 0], np.uint64), qk, ql)  # no such column
        with pytest.raises(lib.KmcpGpuError):
            db.finalize_grouped(good, np.array([0, 1, 0, 0], np.uint64), qk, ql)  # offsets run backwards / past the end
        with pytest.raises(lib.KmcpGpuError):
            db.finalize_grouped(good, np.array([1, 1, 1, 0], np.uint64), qk, ql)  # must start at 0


@pytest.mark.parametrize("kw", [dict(), dict(sort_by=1), dict(sort_by=2, top_n_scores=2), dict(do_not_sort=1), dict(top_n_scores=1)])
def test_short_segments_out_of_order_are_sorted_here_too(kw):
    """a caller of the public kmcpg_finalize_grouped whose short segments are NOT in K3's order (a list that did not come
    from kmcpg_group_device) used to get them streamed as given - wrong order, and --keep-top-scores cutting in the wrong place.
    Every record is now compared with its predecessor; a segment with one out of place goes through the host sort."""
    from kmcp_amd import Database, default_params, lib
    rng = np.random.default_rng(13)
    n_reads, per = 500, 40  # > 8 per read: the scratch-array path; the first 50 reads get 5 each: the in-place path
    with Database.open(DB, device=-1) as db:
        ncols = int(db.info.n_cols)
        per = min(per, ncols)
        assert per > 8
        per_read = np.where(np.arange(n_reads) < 50, 5, per)
        reads = np.repeat(np.arange(n_reads, dtype=np.uint32), per_read)
        hits = np.empty(len(reads), dtype=lib.HIT_DTYPE)
        hits["read"] = reads
        hits["col"] = np.concatenate([rng.permutation(ncols)[:c] for c in per_read]).astype(np.uint32)
        hits["count"] = rng.integers(75, 131, size=len(reads)).astype(np.uint32)
        qk = np.full(n_reads, 130, np.int32)
        ql = np.full(n_reads, 150, np.int32)
        p = default_params(**kw)
        want = db.finalize(hits, qk, ql, params=p)
        # grouped by read (hits are), but in the random order they were drawn in
        pairs = np.ascontiguousarray(np.stack([hits["col"], hits["count"]], axis=1).astype(np.uint32))
        roffs = np.concatenate([[0], np.cumsum(per_read), [0]]).astype(np.uint64)
        got = db.finalize_grouped(pairs, roffs, qk, ql, params=p)
        assert np.array_equal(got.offs, want.offs) and got.matches.tobytes() == want.matches.tobytes()
        assert len(want.matches) > 400


def test_expand_pairs_rebuilds_the_records_bit_for_bit():
    """kmcpg_expand_pairs (the caller-side half of the compact results, round 5): the Match records derived from a query's final
    (column, mKmers) pairs are the very bytes kmcpg_finalize wrote for them - qCov, tCov, jacc, the FPR column, column metadata."""
    from kmcp_amd import Database, default_params, lib
    rng = np.random.default_rng(17)
    n_reads, per = 400, 9
    with Database.open(DB, device=-1) as db:
        ncols = int(db.info.n_cols)
        per = min(per, ncols)
        hits = np.empty(n_reads * per, dtype=lib.HIT_DTYPE)
        hits["read"] = np.repeat(np.arange(n_reads, dtype=np.uint32), per)
        hits["col"] = np.concatenate([rng.permutation(ncols)[:per] for _ in range(n_reads)]).astype(np.uint32)
        qk = rng.integers(100, 3000, size=n_reads).astype(np.int32)  # short reads and long ones (FPR rows beyond the cached triangle)
        hits["count"] = (qk[hits["read"]] * rng.uniform(0.56, 1.0, size=len(hits))).astype(np.uint32) + 1
        hits["count"] = np.minimum(hits["count"], qk[hits["read"]].astype(np.uint32))
        want = db.finalize(hits, qk, (qk + 20).astype(np.int32), params=default_params())
        assert len(want.matches) > 1000
        for i in range(n_reads):
            ms = want.read(i)
            pairs = np.stack([ms["col"].astype(np.uint32), ms["mkmers"].astype(np.uint32)], axis=1) if len(ms) else np.zeros((0, 2), np.uint32)
            got = db.expand_pairs(int(qk[i]), pairs)
            assert got.tobytes() == ms.tobytes(), i
        with pytest.raises(lib.KmcpGpuError):
            db.expand_pairs(130, np.array([[ncols + 3, 100]], np.uint32))
