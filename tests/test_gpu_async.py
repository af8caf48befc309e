"""kmcpg_submit / kmcpg_wait (the asynchronous form of kmcpg_search_batch): several batches in flight, waited for out of
order and from several threads, give the oracle's results; the caller's buffers are free as soon as submit returns; retries
(--try-se) inside kmcpg_wait work while other batches are in flight; multi-k databases fall back to the smaller k
(util-db-search.go:764, :1016-1022)."""
import re
import threading

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(oracle_lib, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("async")
    genomes = synth.random_genomes(12, 20000, seed=31)
    db_dir = synth.make_db(tmp, genomes, k=21, n_chunks=2, threads=4)
    return genomes, db_dir


def _tuples(res, i):
    return (int(res.qlen[i]), int(res.qkmers[i]), [(int(m["col"]), int(m["mkmers"]), float(m["qcov"])) for m in res.read(i)])


def test_batches_in_flight_out_of_order(world, oracle_lib):
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    genomes, db_dir = world
    odb = O.OracleDB(db_dir)
    batches = [synth.sample_reads(genomes, 200 + 37 * i, 150, sub_rate=0.01, seed=100 + i, frac_random=0.2) for i in range(7)]
    with Database.open(db_dir) as db:
        tickets, results = {}, {}

        def submit(i):
            seqs, offs = lib.pack_reads(batches[i])
            tickets[i] = db.submit(seqs, offs, params=default_params())
            seqs[:] = ord("N")  # the caller's buffers belong to the caller again
            offs[:] = 0

        for i in range(4):
            submit(i)
        assert lib.load().kmcpg_close(db._h) == -7  # tickets outstanding: the handle stays open
        with pytest.raises(lib.KmcpGpuError) as e:  # four lanes, all in flight: submit reports it instead of blocking
            submit(4)
        assert e.value.code == -7
        for done, nxt in ((3, 4), (0, 5), (4, 6), (1, None), (6, None), (2, None), (5, None)):  # out of order
            results[done] = db.wait(tickets[done])
            if nxt is not None:
                submit(nxt)
        total = 0
        for i, reads in enumerate(batches):
            total += synth.assert_parity(odb, results[i], reads)
        assert total > 500
    odb.close()


def test_two_threads_share_a_handle(world, oracle_lib):
    from kmcp_amd import Database, default_params
    O = oracle_lib
    genomes, db_dir = world
    odb = O.OracleDB(db_dir)
    work = {t: [synth.sample_reads(genomes, 300, 150, seed=7 * t + j, frac_random=0.1) for j in range(4)] for t in range(3)}
    out, errs = {}, []
    with Database.open(db_dir) as db:
        def run(t):
            try:
                out[t] = [db.search(r, params=default_params()) for r in work[t]]
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=run, args=(t,)) for t in work]
        [x.start() for x in th]
        [x.join() for x in th]
    assert not errs, errs
    for t in work:
        for reads, res in zip(work[t], out[t]):
            synth.assert_parity(odb, res, reads)
    odb.close()


def test_try_se_retries_with_batches_in_flight(world, oracle_lib):
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    genomes, db_dir = world
    odb = O.OracleDB(db_dir)
    rng = np.random.default_rng(4)
    r1 = synth.sample_reads(genomes, 240, 150, sub_rate=0.01, seed=1, frac_random=0.0, both_strands=False)
    r2 = []
    for i in range(len(r1)):  # a third of the mates are junk: the pair fails, read 1 alone matches; some mates are too short
        if i % 3 == 0:
            r2.append(bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)))
        elif i % 7 == 0:
            r2.append(b"ACGT" * 4)
        else:
            r2.append(synth.sample_reads(genomes, 1, 150, seed=1000 + i, frac_random=0.0)[0])
    other = synth.sample_reads(genomes, 500, 150, seed=9)
    p = default_params(try_se=1)
    op = O.default_params(try_se=1, fpr_buf_size=499)
    with Database.open(db_dir) as db:
        s1, o1 = lib.pack_reads(r1)
        s2, o2 = lib.pack_reads(r2)
        so, oo = lib.pack_reads(other)
        t_other1 = db.submit(so, oo, params=default_params())
        t_pe = db.submit(s1, o1, s2, o2, params=p)
        t_other2 = db.submit(so, oo, params=default_params())
        res_pe = db.wait(t_pe)  # its retries run while t_other2 is still on the GPU and t_other1 holds a lane
        a, b = db.wait(t_other2), db.wait(t_other1)
    assert synth.assert_parity(odb, res_pe, r1, r2, oparams=op) > 100
    assert [_tuples(a, i) for i in range(len(other))] == [_tuples(b, i) for i in range(len(other))]
    synth.assert_parity(odb, a, other)
    odb.close()


def test_multi_k_database_falls_back_to_smaller_k(oracle_lib, tmp_path):
    """`kmcp compute -k 21 -k 31`: every column holds the k-mers of both sizes, `ks: [21, 31]` in __db.yml, the .uniki headers
    carry the largest k (util-db-search.go:690).  A read whose 31-mers are all broken by substitutions still matches with k=21;
    kSize says which k answered."""
    from kmcp_amd import Database, default_params
    O = oracle_lib
    genomes = synth.random_genomes(10, 8000, seed=77)
    cols = []
    for gi, g in enumerate(genomes):
        h = np.concatenate([O.generate_kmers(g, O.sketch_cfg(k=k)) for k in (21, 31)])
        cols.append((f"g{gi}", len(g), 0, 1, O.sort_unique(h)))
    db_dir = O.build_db(str(tmp_path), O.sketch_cfg(k=31), cols, num_hashes=1, fpr=0.3, threads=2)
    yml = open(db_dir + "/__db.yml").read()
    yml2 = re.sub(r"ks:\n- 31\n", "ks:\n- 21\n- 31\n", yml)
    assert yml2 != yml
    open(db_dir + "/__db.yml", "w").write(yml2)

    rng = np.random.default_rng(2)
    reads, reads2 = [], []
    for i in range(300):
        g = genomes[i % len(genomes)]
        pos = int(rng.integers(0, len(g) - 150))
        r = bytearray(g[pos:pos + 150])
        if i % 3 == 1:  # a substitution every 24 bases: no intact 31-mer, 21-mers survive... but too few for -t 0.55
            for j in range(12, 150, 24):
                r[j] = ord("A") if r[j] != ord("A") else ord("C")
        elif i % 3 == 2:  # every 28 bases: no intact 31-mer (28 < 31); runs of 27 leave 7 intact 21-mers per run
            for j in range(5, 150, 28):
                r[j] = ord("A") if r[j] != ord("A") else ord("C")
        reads.append(bytes(r))
        pos2 = int(rng.integers(0, len(g) - 150))
        reads2.append(g[pos2:pos2 + 150] if i % 5 else bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)))
    odb = O.OracleDB(db_dir)
    with Database.open(db_dir) as db:
        assert db.info.k == 31
        for kw, okw, r2 in ((dict(min_qcov=0.2), dict(min_qcov=0.2), None),
                            (dict(min_qcov=0.4, try_se=1), dict(min_qcov=0.4, try_se=1, fpr_buf_size=499), reads2)):
            res = db.search(reads, r2, params=default_params(**kw))
            n = synth.assert_parity(odb, res, reads, r2, oparams=O.default_params(**okw))
            ks = [odb.search(r, r2[i] if r2 else None, params=O.default_params(**okw))["k"] for i, r in enumerate(reads)]
            assert list(res.ksize) == ks
            assert n > 150 and 21 in ks and 31 in ks
        # an explicit k searches with that size only
        res31 = db.search(reads, params=default_params(min_qcov=0.2, k=31))
        assert set(res31.ksize.tolist()) == {31}
        with pytest.raises(Exception):
            db.search(reads, params=default_params(k=25))
    odb.close()


def test_async_stress_short():
    """tools/stress_async.py for a few seconds: four threads, random submit / wait / search_batch patterns on a single-GPU
    handle and on a two-shard in-process handle, every result compared with the batch searched alone (a 60-second run of the
    same script checked 375 000 batches on the final build)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_async.py"), "6", "4"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "stress ok" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


@pytest.mark.parametrize("budget_mb", [None, "0"])
def test_hit_heavy_batches_follow_the_data(oracle_lib, tmp_path, monkeypatch, budget_mb):
    """(budget_mb = "0": KMCPG_HIT_BUDGET_MB=0 — the lanes' hit buffers may not grow beyond the plain size, every heavy batch
    overflows and is searched a second time with room for its hits: same results.)
    A database of 70 close relatives: every read matches ~60 columns.  The hit buffers of the lanes learn that from the first
    large batch (kmcpg_wait reruns it once with room for every hit), later batches fit at once; the eager read-back stays
    bounded at 32 hits per read, so the rest of every batch's hits arrives through the late copy in kmcpg_wait.  Batches of
    2 000 reads in flight on all lanes, a small one (below the size that updates the estimate) and a hit-free one in between."""
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    rng = np.random.default_rng(91)
    base = synth.random_genomes(1, 6000, seed=90)[0]
    genomes = []
    for i in range(70):
        g = np.frombuffer(base, dtype=np.uint8).copy()
        m = rng.random(len(g)) < 0.004
        g[m] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(m.sum()))
        genomes.append(g.tobytes())
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=1)
    odb = O.OracleDB(db_dir)
    heavy = [synth.sample_reads([base], 2000, 150, sub_rate=0.0, seed=200 + i, frac_random=0.0) for i in range(6)]
    small = synth.sample_reads([base], 40, 150, sub_rate=0.0, seed=300, frac_random=0.0)
    empty = synth.sample_reads([base], 1500, 150, seed=301, frac_random=1.0)
    order = [heavy[0], heavy[1], small, heavy[2], empty, heavy[3], heavy[4], heavy[5]]
    if budget_mb is not None:
        monkeypatch.setenv("KMCPG_HIT_BUDGET_MB", budget_mb)
    with Database.open(db_dir) as db:
        alone = [db.search(b, params=default_params()) for b in order]  # one at a time (also the first overflow + rerun)
        assert alone[0].offs[-1] > 40 * len(order[0]) and alone[4].offs[-1] == 0
        tickets = []
        got = []
        for b in order:
            seqs, offs = lib.pack_reads(b)
            while True:
                try:
                    tickets.append(db.submit(seqs, offs, params=default_params()))
                    break
                except lib.KmcpGpuError as e:
                    assert e.code == -7
                    got.append(db.wait(tickets.pop(0)))
        while tickets:
            got.append(db.wait(tickets.pop(0)))
        for a, g in zip(alone, got):
            for f in ("qlen", "qkmers", "offs", "matches"):
                assert np.array_equal(getattr(a, f), getattr(g, f)), f
        # the oracle on a sample of a heavy batch
        sub = order[3][:80]
        assert synth.assert_parity(odb, db.search(sub, params=default_params()), sub) > 40 * len(sub)
    odb.close()


def test_search_batch_cuts_large_batches_into_pieces(oracle_lib, tmp_path):
    """kmcpg_search_batch sends a batch of >= 32 768 queries through the lanes as up to 4 pieces (upload / kernels / copy / expansion
    of consecutive pieces overlap) and assembles ONE result in place: byte for byte the result of the unsplit call (KMCPG_PIECES=0),
    for single reads, pairs, every sort mode with --keep-top-scores, a hit buffer that overflows inside a piece (hundreds of chance
    hits per read at -t 0.31 -f 1), and from two threads at once (a thread blocks for a lane only while it holds none)."""
    import os
    import threading
    from kmcp_amd import Database, default_params
    O = oracle_lib
    genomes = synth.random_genomes(60, 6000, seed=171)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=2, overlap=150, threads=8)  # 120 columns in blocks of 16
    rng = np.random.default_rng(172)
    pool = synth.sample_reads(genomes, 3000, 150, sub_rate=0.01, seed=173, frac_random=0.2)
    pool2 = synth.sample_reads(genomes, 3000, 150, sub_rate=0.01, seed=174, frac_random=0.5)
    idx = rng.integers(0, len(pool), size=40000)
    reads = [pool[i] for i in idx]
    reads2 = [pool2[i] for i in idx]

    def same(a, b):
        for f in ("qlen", "qkmers", "ksize", "offs"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert a.matches.tobytes() == b.matches.tobytes()

    cases = [(dict(), False), (dict(sort_by=2, top_n_scores=2), False), (dict(min_tcov=0.01, sort_by=1), False), (dict(fpr_buf_size=499), True),
             (dict(min_qcov=0.31, max_fpr=1.0, min_matched=1), False)]
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as db:
            for kw, paired in cases:
                p = default_params(**kw)
                os.environ["KMCPG_PIECES"] = "0"
                try:
                    whole = db.search(reads, reads2 if paired else None, params=p)
                finally:
                    os.environ.pop("KMCPG_PIECES")
                pieces = db.search(reads, reads2 if paired else None, params=p)
                same(whole, pieces)
                os.environ["KMCPG_PIECES"] = "2"
                try:
                    same(whole, db.search(reads, reads2 if paired else None, params=p))
                finally:
                    os.environ.pop("KMCPG_PIECES")
                if "max_fpr" in kw:
                    assert len(whole.matches) > 8 * len(reads)  # more hits than the plain hit buffer of a piece holds: the rerun path
            # the oracle on a sample of the default case
            res = db.search(reads, params=default_params())
            sample = slice(0, 600)
            sub = type(res)(res.qlen[sample], res.qkmers[sample], res.offs[:601], res.matches[:int(res.offs[600])], res.k, res.ksize[sample])
            assert synth.assert_parity(odb, sub, reads[sample]) > 300
            # two threads at once
            out = [None, None]

            def work(i):
                out[i] = db.search(reads[i * 3:] + reads[:i * 3], params=default_params())

            th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
            [t.start() for t in th]
            [t.join() for t in th]
            same(out[0], res)
            assert np.array_equal(out[1].qkmers, np.concatenate([res.qkmers[3:], res.qkmers[:3]]))
    finally:
        odb.close()
