"""GPU parity: libkmcpgpu (HIP, through the C ABI) vs the CPU oracle on identical seeded inputs.

Bit-exact per-read (qLen, qKmers, {(target column, mKmers, qCov, tCov, jacc)}); FPR with tolerance.
"""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from kmcp_amd import Database, default_params, lib
    lib.load()
    return dict(Database=Database, default_params=default_params, lib=lib)


def _run(G, O, db_dir, reads, reads2=None, oracle_kw=None, gpu_kw=None):
    odb = O.OracleDB(db_dir)
    try:
        with G["Database"].open(db_dir, device=0) as db:
            res = db.search(reads, reads2, params=G["default_params"](**(gpu_kw or {})))
        n = synth.assert_parity(odb, res, reads, reads2, O.default_params(**(oracle_kw or {})))
    finally:
        odb.close()
    return n, res


def test_narrow_rows_single_hash(G, oracle_lib, tmp_path):
    """config-1 shape scaled down: many small blocks with 3-byte rows (LPR=4 kernel), 150-bp reads, k=21."""
    O = oracle_lib
    genomes = synth.random_genomes(24, 20000, seed=1)
    db_dir = synth.make_db(tmp_path, genomes, k=21, n_chunks=4, overlap=150, threads=4)  # 96 columns, sBlock 24
    reads = synth.sample_reads(genomes, 1500, 150, sub_rate=0.01, seed=2, frac_random=0.1)
    n, res = _run(G, O, db_dir, reads)
    assert n > 1000
    assert (res.qkmers == 130).sum() > 1000


def test_medium_rows_lpr16(G, oracle_lib, tmp_path):
    """One block of 1000 columns: 125-byte rows -> 128-byte stride (LPR=16 kernel)."""
    O = oracle_lib
    genomes = synth.random_genomes(1000, 1500, seed=3)
    db_dir = synth.make_db(tmp_path, genomes, k=21, block_size=1000)
    reads = synth.sample_reads(genomes, 800, 150, sub_rate=0.02, seed=4, frac_random=0.2)
    n, _ = _run(G, O, db_dir, reads)
    assert n > 400


def test_wide_rows_two_tiles(G, oracle_lib, tmp_path):
    """One block of 9000 columns: 1125-byte rows -> two 1-KiB tiles per row (LPR=64 kernel), plus a ragged
    second block of 37 columns."""
    O = oracle_lib
    genomes = synth.random_genomes(9037, 600, seed=5)
    db_dir = synth.make_db(tmp_path, genomes, k=21, block_size=9000)
    reads = synth.sample_reads(genomes, 600, 150, sub_rate=0.01, seed=6, frac_random=0.1)
    n, _ = _run(G, O, db_dir, reads)
    assert n > 300


def test_multi_hash_scaled(G, oracle_lib, tmp_path):
    """FracMinHash database with 3 hash functions (the demo-searching shape): AND of 3 rows, dedup path."""
    O = oracle_lib
    genomes = synth.random_genomes(12, 200000, seed=7)
    db_dir = synth.make_db(tmp_path, genomes, k=31, num_hashes=3, fpr=0.01, scale=20, threads=2)
    # queries = mutated genomes, each ~10 k sketch hashes -> sort+unique on the GPU, 16-plane counters
    rng = np.random.default_rng(8)
    queries = []
    for g in genomes[:6]:
        a = np.frombuffer(g, dtype=np.uint8).copy()
        m = rng.random(len(a)) < 0.002
        a[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
        queries.append(a.tobytes())
    kw = dict(min_qcov=0.4, sort_by=2)
    n, res = _run(G, O, db_dir, queries, oracle_kw=kw, gpu_kw=kw)
    assert n >= 6
    assert (res.qkmers > 256).all()


def test_paired_end_dedup_and_try_se(G, oracle_lib, tmp_path):
    """Paired-end 2x150: 260 k-mers > -u 256 so every query goes through sort+unique; --try-se retries."""
    O = oracle_lib
    genomes = synth.random_genomes(16, 30000, seed=9)
    db_dir = synth.make_db(tmp_path, genomes, k=21, n_chunks=2, overlap=150, threads=2)
    r1 = synth.sample_reads(genomes, 400, 150, sub_rate=0.01, seed=10, frac_random=0.0)
    # mates: half from the same genome population, half random so that only one mate matches
    r2 = synth.sample_reads(genomes, 400, 150, sub_rate=0.01, seed=11, frac_random=0.5)
    for try_se in (0, 1):
        kw = dict(try_se=try_se)
        okw = dict(try_se=try_se, fpr_buf_size=499)
        n, res = _run(G, O, db_dir, r1, r2, oracle_kw=okw, gpu_kw=kw)
        assert n > 100


def test_edge_cases(G, oracle_lib, tmp_path):
    """Empty, shorter than k, shorter than -m, exactly k, N-rich, lower-case, IUPAC, and too-few-k-mers reads."""
    O = oracle_lib
    genomes = synth.random_genomes(10, 5000, seed=12)
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=2)
    g0 = genomes[0]
    reads = [
        b"",                         # empty
        g0[:10],                     # shorter than k and than -m
        g0[:21],                     # one k-mer, shorter than -m 30
        g0[:30],                     # == -m: 10 k-mers == -c
        g0[:29],                     # < -m
        g0[100:250],                 # exact
        g0[100:250].lower(),         # lower case hashes like upper case
        g0[100:170] + b"N" + g0[171:250],  # one N: 21 k-mers contain it
        b"N" * 150,                  # all N: every hash is 0 and dropped
        g0[300:380] + b"RYKM" + g0[384:450],  # IUPAC codes (rc seed quirk)
        g0[500:540],                 # 20 k-mers
        genomes[3][1000:1150],
    ]
    reads += synth.sample_reads(genomes, 200, 150, sub_rate=0.03, seed=13, frac_random=0.2, n_rate=0.01)
    n, res = _run(G, O, db_dir, reads)
    assert n > 50
    assert int(res.qkmers[0]) == 0 and int(res.qkmers[8]) == 0


def test_thresholds_and_sorting(G, oracle_lib, tmp_path):
    """Non-default -t/-T/-c/-f, --sort-by tcov/jacc, --keep-top-scores."""
    O = oracle_lib
    genomes = synth.random_genomes(20, 8000, seed=14)
    # related genomes so a read matches several targets with different scores
    rng = np.random.default_rng(15)
    rel = []
    for g in genomes[:5]:
        for rate in (0.0, 0.01, 0.02, 0.04):
            a = np.frombuffer(g, dtype=np.uint8).copy()
            m = rng.random(len(a)) < rate
            a[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
            rel.append(a.tobytes())
    db_dir = synth.make_db(tmp_path, rel + genomes[5:], k=21, threads=2)
    reads = synth.sample_reads(rel, 300, 150, sub_rate=0.005, seed=16, frac_random=0.1)
    for kw in (dict(min_qcov=0.4, min_tcov=0.001, min_matched=5, max_fpr=0.05), dict(sort_by=1), dict(sort_by=2, top_n_scores=1),
               dict(top_n_scores=2), dict(min_qcov=0.9)):
        n, _ = _run(G, O, db_dir, reads, oracle_kw=kw, gpu_kw=kw)
        assert n > 0


def test_kmer_kernel_matches_oracle(G, oracle_lib, tmp_path):
    """K1 alone: device hashes (plain and FracMinHash) equal the oracle's generateKmers in order."""
    import torch
    O = oracle_lib
    genomes = synth.random_genomes(4, 3000, seed=17)
    for scale in (1, 8):
        d = tmp_path / f"s{scale}"
        db_dir = synth.make_db(d, genomes, k=21, threads=2, scale=scale)
        reads = synth.sample_reads(genomes, 100, 150, seed=18, n_rate=0.02) + [genomes[0][:1000], b"ACGT"]
        seqs, offs = G["lib"].pack_reads(reads)
        with G["Database"].open(db_dir, device=0) as db:
            dev = torch.device("cuda:0")
            t_seqs = torch.from_numpy(seqs).to(dev)
            t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
            t_h = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
            t_nk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
            p = G["default_params"](min_qlen=0, min_matched=1, dedup_threshold=1 << 30)
            db.kmers_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), max(len(r) for r in reads),
                            t_h.data_ptr(), t_h.numel(), None, t_nk.data_ptr(), params=p)
            torch.cuda.synchronize()
            h = t_h.cpu().numpy().view(np.uint64)
            nk = t_nk.cpu().numpy()
        cfg = O.sketch_cfg(k=21, scale=scale)
        for i, r in enumerate(reads):
            want = O.generate_kmers(r, cfg)
            got = h[int(offs[i]):int(offs[i]) + int(nk[i])]
            assert len(want) == nk[i]
            assert np.array_equal(got, want), i


def test_syncmer_and_minimizer_databases(G, oracle_lib, tmp_path):
    """Closed-Syncmer (k=21 s=11: BASELINE config 4 shape) and Minimizer (w=10) databases: short reads (no dedup: the
    emission multiplicity matters) and 5-kb reads (sort+unique, 16-plane counters)."""
    O = oracle_lib
    genomes = synth.random_genomes(12, 40000, seed=30)
    for name, kw in (("syn", dict(syncmer_s=11)), ("min", dict(minimizer_w=10)), ("synscaled", dict(syncmer_s=9, scale=4))):
        db_dir = synth.make_db(tmp_path / name, genomes, k=21, n_chunks=2, overlap=150, threads=2, **kw)
        short = synth.sample_reads(genomes, 300, 150, sub_rate=0.01, seed=31, frac_random=0.1, n_rate=0.003)
        short += [genomes[0][:29], genomes[0][:30], genomes[0][:31], genomes[0][:40], b""]
        long_ = synth.sample_reads(genomes, 40, 5000, sub_rate=0.005, seed=32, frac_random=0.1)
        for reads in (short, long_):
            n, _ = _run(G, O, db_dir, reads)
            assert n > 20


def test_sketch_kernels_match_oracle_in_order(G, oracle_lib, tmp_path):
    """K1 in syncmer / minimizer mode emits exactly the oracle's list (values, order, multiplicity)."""
    import torch
    O = oracle_lib
    genomes = synth.random_genomes(3, 4000, seed=33)
    for kw in (dict(syncmer_s=11), dict(syncmer_s=21), dict(syncmer_s=1), dict(minimizer_w=1), dict(minimizer_w=7), dict(minimizer_w=50)):
        d = tmp_path / ("k" + "_".join(f"{a}{b}" for a, b in kw.items()))
        db_dir = synth.make_db(d, genomes, k=21, threads=2, **kw)
        reads = synth.sample_reads(genomes, 60, 150, seed=34, n_rate=0.02) + [genomes[0][:1500], genomes[1][:69], genomes[1][:70], genomes[1][:71], b"ACGT"]
        seqs, offs = G["lib"].pack_reads(reads)
        with G["Database"].open(db_dir, device=0) as db:
            dev = torch.device("cuda:0")
            t_seqs = torch.from_numpy(seqs).to(dev)
            t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
            t_h = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
            t_nk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
            p = G["default_params"](min_qlen=0, min_matched=1, dedup_threshold=1 << 30)
            db.kmers_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), max(len(r) for r in reads),
                            t_h.data_ptr(), t_h.numel(), None, t_nk.data_ptr(), params=p)
            torch.cuda.synchronize()
            h = t_h.cpu().numpy().view(np.uint64)
            nk = t_nk.cpu().numpy()
        cfg = O.sketch_cfg(k=21, **kw)
        for i, r in enumerate(reads):
            want = O.generate_kmers(r, cfg)
            got = h[int(offs[i]):int(offs[i]) + int(nk[i])]
            assert len(want) == nk[i], (kw, i, len(want), nk[i])
            assert np.array_equal(got, want), (kw, i)


@pytest.mark.parametrize("k,s,scale", [(21, 11, 1), (31, 15, 1), (31, 15, 3), (25, 15, 1), (21, 15, 1), (21, 13, 2), (31, 19, 1)])
def test_rolling_window_sketch_kernel_at_its_boundaries(G, oracle_lib, monkeypatch, k, s, scale):
    """k1_windows_roll (round 6: closed syncmers of long reads by per-lane rolling on 2-bit codes, windows of 12 / 16 / 20 / 24 / 32 s-mers) against the oracle
    and against k1_windows_wave (KMCPG_K1_FLAGS=35) at the edges of what it takes: reads of exactly WR_MIN_WINDOWS windows and one fewer, window
    counts around multiples of 64 x 16 (the lanes' runs are multiples of 16: the last lanes idle or hold one window), soft-masked reads, one N at
    the very end / start (left to the wave kernel), a low-complexity read whose emissions repeat for hundreds of windows (the adjacent-repeat
    stitching across lanes), a homopolymer (hash 0 never: all kept), the longest read the LDS takes; -u above and below the emission count."""
    import torch
    O = oracle_lib
    lib = G["lib"]
    dev = torch.device("cuda:0")
    Lw = 2 * k - s - 1
    g = synth.random_genomes(2, 40000, seed=500 + k + s)
    lens = [Lw + 1022, Lw + 1023, Lw + 1024, Lw - 1 + 64 * 16, Lw - 1 + 64 * 16 + 1, Lw - 1 + 64 * 32 - 1, Lw - 1 + 64 * 32, 5000, 9999, 16384, 20011, 29000]
    reads = [g[0][7:7 + n] for n in lens]
    reads += [g[1][:6000].lower(), g[1][100:7000] + b"N", b"N" + g[1][200:7000], (b"ACGTTGCAAT" * 700)[:6500] + g[1][:3000], b"A" * 5000, b"AC" * 4000,
              g[1][:3000] + g[1][:3000] + g[1][:3000]]
    spec = lib.SynthSpec(k=k, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=8, num_sigs=1000, kmers_per_col=10, seed=1, syncmer_s=s, scale=scale)
    cfg = O.sketch_cfg(k=k, syncmer_s=s, scale=scale)
    seqs, offs = lib.pack_reads(reads)
    t_seqs = torch.from_numpy(seqs).to(dev)
    t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    out = {}
    with G["Database"].open_synthetic(spec) as db:
        for flags in ("3", "35"):
            monkeypatch.setenv("KMCPG_K1_FLAGS", flags)
            for thr in (256, 3000):
                t_h = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
                t_nk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
                p = G["default_params"](min_qlen=0, min_matched=1, dedup_threshold=thr)
                db.kmers_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), max(len(r) for r in reads), t_h.data_ptr(), t_h.numel(), None,
                                t_nk.data_ptr(), params=p)
                torch.cuda.synchronize()
                out[(flags, thr)] = (t_h.cpu().numpy().view(np.uint64), t_nk.cpu().numpy())
    for thr in (256, 3000):
        h3, n3 = out[("3", thr)]
        h35, n35 = out[("35", thr)]
        assert np.array_equal(n3, n35), (thr, n3, n35)
        for i, r in enumerate(reads):
            raw = O.generate_kmers(r, cfg)
            want = O.sort_unique(raw) if len(raw) > thr else raw
            got = h3[int(offs[i]):int(offs[i]) + int(n3[i])]
            assert len(want) == n3[i], (k, s, thr, i, len(r), len(want), int(n3[i]))
            assert np.array_equal(got, want), (k, s, thr, i, len(r))
            assert np.array_equal(h35[int(offs[i]):int(offs[i]) + int(n35[i])], want), (k, s, thr, i)


def test_mixed_block_widths_and_hash_counts(G, oracle_lib, tmp_path):
    """A database whose blocks fall into all three kernel classes (1125-, 38- and 3-byte rows), with 2 and 4 hash
    functions, k = 31 / 64 / 65 (rotations wrap past 64 bits)."""
    import ctypes as C
    O = oracle_lib
    for k, nh, fpr in ((31, 2, 0.05), (64, 4, 0.1), (65, 1, 0.3)):
        genomes = synth.random_genomes(9320, 400, seed=60 + k)
        cfg = O.sketch_cfg(k=k)
        cols = synth.make_columns(genomes, cfg)
        d = tmp_path / f"k{k}"
        root = d / "R001"
        root.mkdir(parents=True)
        # three blocks written directly: 9000, 300 and 20 columns
        arr = (O.Column * len(cols))()
        keep = []
        for i, (name, gsize, ci, nch, h) in enumerate(cols):
            h = np.ascontiguousarray(h, dtype=np.uint64)
            keep.append(h)
            arr[i] = O.Column(name.encode(), gsize, ci, nch, h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h))
        lo = 0
        files = []
        for bi, n in enumerate((9000, 300, 20)):
            sub = (O.Column * n)(*[arr[lo + j] for j in range(n)])
            f = f"_block{bi + 1:03d}.uniki"
            assert O.lib().ko_write_block(str(root / f).encode(), k, 1, nh, fpr, 0, sub, n) == 0
            files.append(f)
            lo += n
        (root / "__db.yml").write_text(
            f"version: 4\nunikiVersion: 4\nalias: t\nk: {k}\nks:\n- {k}\nhashed: true\ncanonical: true\nscaled: false\nscale: 1\n"
            f"minimizer: false\nminimizer-w: 0\nsyncmer: false\nsyncmer-s: 0\nhashes: {nh}\nfpr: {fpr}\nfiles:\n" + "".join(f"- {f}\n" for f in files))
        reads = synth.sample_reads(genomes, 400, 150, sub_rate=0.005, seed=61, frac_random=0.1)
        kw = dict(min_qcov=max(0.55, fpr + 0.2))
        n, _ = _run(G, O, str(root), reads, oracle_kw=kw, gpu_kw=kw)
        assert n > 200


def test_very_long_query_24_plane_counters(G, oracle_lib, tmp_path):
    """A 90-kb query with every k-mer kept (n > 65 535 => 24 counter planes, global-memory sort) and dedup disabled/enabled."""
    O = oracle_lib
    genomes = synth.random_genomes(6, 90000, seed=70)
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=2)
    reads = [genomes[0], genomes[1][:70000] + genomes[2][:20000], genomes[3][5:80005]]
    for kw in (dict(), dict(dedup_threshold=1000000)):
        n, res = _run(G, O, db_dir, reads, oracle_kw=kw, gpu_kw=kw)
        assert n >= 3 and int(res.qkmers[0]) > 65535


def test_in_process_multi_device_handle(G, oracle_lib, tmp_path):
    """kmcpg_open_devices: one process, the blocks partitioned over a device list (here three shards on GPU 0), batches
    fanned out from one host thread per shard and merged; incl. paired-end --try-se on top of the merged result."""
    O = oracle_lib
    genomes = synth.random_genomes(40, 10000, seed=80)
    db_dir = synth.make_db(tmp_path, genomes, k=21, n_chunks=2, overlap=150, threads=8)  # 80 columns -> 8 blocks
    r1 = synth.sample_reads(genomes, 500, 150, seed=81, frac_random=0.05)
    r2 = synth.sample_reads(genomes, 500, 150, seed=82, frac_random=0.5)
    odb = O.OracleDB(db_dir)
    try:
        with G["Database"].open_devices(db_dir, [0, 0, 0]) as db:
            assert db.info.n_blocks_local == db.info.n_blocks >= 3
            res = db.search(r1, params=G["default_params"]())
            assert synth.assert_parity(odb, res, r1) > 300
            res = db.search(r1, r2, params=G["default_params"](try_se=1))
            assert synth.assert_parity(odb, res, r1, r2, O.default_params(try_se=1)) > 100
    finally:
        odb.close()


def test_rccl_exchange_of_the_in_process_handle(G, oracle_lib, tmp_path, monkeypatch):
    """The exchange step of kmcpg_open_devices (exchange.cpp): hit lists gathered on the first GPU with RCCL send/recv and
    copied to the host once.  One GPU here, so the communicator has one rank (KMCPG_RCCL=force: send/recv to self goes through
    the same group calls, buffers and single copy); duplicate ordinals must fall back to the host merge and say why."""
    O = oracle_lib
    genomes = synth.random_genomes(30, 10000, seed=83)
    db_dir = synth.make_db(tmp_path, genomes, k=21, n_chunks=2, overlap=150, threads=8)
    r1 = synth.sample_reads(genomes, 2500, 150, seed=84, frac_random=0.05)
    r2 = synth.sample_reads(genomes, 2500, 150, seed=85, frac_random=0.5)
    odb = O.OracleDB(db_dir)
    try:
        with G["Database"].open_devices(db_dir, [0, 0]) as db:
            assert db.exchange_info().startswith("host merge") and "duplicate device ordinals" in db.exchange_info()
        with G["Database"].open(db_dir, device=0) as db:
            assert db.exchange_info().startswith("single device")
        monkeypatch.setenv("KMCPG_RCCL", "force")
        with G["Database"].open_devices(db_dir, [0]) as db:
            assert db.exchange_info() == "RCCL gather over 1 device(s)", db.exchange_info()
            res = db.search(r1, params=G["default_params"]())
            assert synth.assert_parity(odb, res, r1) > 1500
            res = db.search(r1, r2, params=G["default_params"](try_se=1))  # retries run through the exchange too
            assert synth.assert_parity(odb, res, r1, r2, O.default_params(try_se=1)) > 500
            # several batches in flight, gathered from two waiter threads
            import threading
            seqs, offs = G["lib"].pack_reads(r1)
            ref = db.search_packed(seqs, offs, params=G["default_params"]())
            out = {}

            def pump(t):
                tk = [db.submit(seqs, offs, params=G["default_params"]()) for _ in range(2)]
                out[t] = [db.wait(x) for x in tk]
            th = [threading.Thread(target=pump, args=(t,)) for t in range(2)]
            [x.start() for x in th]
            [x.join() for x in th]
            for t in range(2):
                for r in out[t]:
                    assert np.array_equal(r.offs, ref.offs) and np.array_equal(r.matches, ref.matches)
        monkeypatch.setenv("KMCPG_RCCL", "0")
        with G["Database"].open_devices(db_dir, [0]) as db:
            assert "KMCPG_RCCL=0" in db.exchange_info()
    finally:
        odb.close()


def test_in_process_handle_over_every_visible_gpu(G, oracle_lib, tmp_path):
    """The first time this suite meets a box with two or more GPUs, kmcpg_open_devices runs for real: one shard per device, one
    RCCL communicator per device (ncclCommInitAll), the shards' hit lists gathered on the first GPU with grouped send/recv over
    xGMI (exchange.cpp) — the reference's concatenation of its per-block workers' replies (util-db-search.go:939-964).  It must say
    so, match the oracle (single-end; paired-end with --try-se, whose retries go through the exchange again) and keep four
    batches in flight from two waiter threads.  On a one-GPU box this is skipped, with the reason."""
    import threading
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip(f"{n_dev} GPU visible: the RCCL exchange between devices needs two (its one-rank form is test_rccl_exchange_of_the_in_process_handle)")
    n_dev = min(n_dev, 8)
    O = oracle_lib
    genomes = synth.random_genomes(40, 10000, seed=86)
    db_dir = synth.make_db(tmp_path, genomes, k=21, n_chunks=2, overlap=150, threads=8)  # 80 columns -> 8+ blocks: every device gets one
    r1 = synth.sample_reads(genomes, 2500, 150, seed=87, frac_random=0.05)
    r2 = synth.sample_reads(genomes, 2500, 150, seed=88, frac_random=0.5)
    odb = O.OracleDB(db_dir)
    try:
        with G["Database"].open_devices(db_dir, list(range(n_dev))) as db:
            assert db.exchange_info() == f"RCCL gather over {n_dev} device(s)", db.exchange_info()
            assert db.info.n_blocks_local == db.info.n_blocks >= n_dev
            res = db.search(r1, params=G["default_params"]())
            assert synth.assert_parity(odb, res, r1) > 1500
            res = db.search(r1, r2, params=G["default_params"](try_se=1))
            assert synth.assert_parity(odb, res, r1, r2, O.default_params(try_se=1)) > 500
            seqs, offs = G["lib"].pack_reads(r1)
            ref = db.search_packed(seqs, offs, params=G["default_params"]())
            out, err = {}, []

            def pump(t):
                try:
                    tk = [db.submit(seqs, offs, params=G["default_params"]()) for _ in range(2)]
                    out[t] = [db.wait(x) for x in tk]
                except Exception as e:  # noqa: BLE001 - reported by the main thread
                    err.append(e)
            th = [threading.Thread(target=pump, args=(t,)) for t in range(2)]
            [x.start() for x in th]
            [x.join() for x in th]
            assert not err, err
            for t in range(2):
                for r in out[t]:
                    assert np.array_equal(r.offs, ref.offs) and np.array_equal(r.matches, ref.matches)
            # compact results over the exchange (K3 on the gathering GPU over the concatenated lists)
            pr = db.search_pairs(r1, params=G["default_params"]())
            assert np.array_equal(pr.offs, ref.offs) and np.array_equal(pr.pairs["col"], ref.matches["col"]) and np.array_equal(pr.pairs["count"], ref.matches["mkmers"])
        # the same database on one device: identical records (the merged list is the one-GPU list)
        with G["Database"].open(db_dir, device=0) as db1:
            one = db1.search_packed(seqs, offs, params=G["default_params"]())
            assert np.array_equal(one.offs, ref.offs) and np.array_equal(one.matches, ref.matches)
    finally:
        odb.close()


@pytest.mark.parametrize("split_min", ["50", "3000"])
def test_long_query_split_path(G, oracle_lib, tmp_path, monkeypatch, split_min):
    """The chunked long-query form of the COBS kernel (whole genomes): forced onto ordinary reads with KMCPG_SPLIT_MIN so
    that every kernel class (3-byte, 125-byte, 1125-byte rows; 1 and 3 hashes) and many chunk boundaries are exercised."""
    O = oracle_lib
    monkeypatch.setenv("KMCPG_SPLIT_MIN", split_min)
    # narrow rows + long reads (several chunks of 8192 k-mers when split_min is small)
    genomes = synth.random_genomes(12, 30000, seed=90)
    db_dir = synth.make_db(tmp_path / "a", genomes, k=21, n_chunks=2, overlap=150, threads=4)
    reads = synth.sample_reads(genomes, 200, 150, seed=91) + [genomes[0], genomes[1][:20000], genomes[2][:8192 + 20], genomes[3][:8192 + 21]]
    n, res = _run(G, O, db_dir, reads, oracle_kw=dict(dedup_threshold=1 << 30), gpu_kw=dict(dedup_threshold=1 << 30))
    assert n > 150 and int(res.qkmers[200]) == 29980
    # medium and wide rows in one database, 3 hashes, FracMinHash
    genomes = synth.random_genomes(9100, 500, seed=92)
    db_dir = synth.make_db(tmp_path / "b", genomes, k=25, num_hashes=3, fpr=0.05, block_size=9000)
    reads = synth.sample_reads(genomes, 300, 300, sub_rate=0.01, seed=93)
    n, _ = _run(G, O, db_dir, reads)
    assert n > 150


@pytest.mark.parametrize("k", [11, 21, 33, 64, 65, 66, 90, 130])
def test_k1_all_forms_across_k(G, oracle_lib, k):
    """K1 vs the oracle's generateKmers for every kernel form — one wave per read (prefix-XOR scan for k <= 65, closed form
    above), one workgroup per read (LDS prefix arrays), one workgroup per 65536-position segment — in plain, scaled,
    syncmer and minimizer mode; rotation amounts wrap at 64, tiles end inside k-mers, reads end inside tiles."""
    import torch
    O = oracle_lib
    lib = G["lib"]
    dev = torch.device("cuda:0")
    genomes = synth.random_genomes(2, 150000, seed=200 + k)
    short = synth.sample_reads(genomes, 40, 150, seed=201, n_rate=0.02) + [genomes[0][:n] for n in (k - 1, k, k + 1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 1000, 2048)]
    long_ = [genomes[0][:n] for n in (2049, 3000, 1024 + k - 1, 1024 + k, 2 * 1024 + k - 1, 5000)] + short[:5]
    huge = [genomes[1][:140000], genomes[0][:65536 + k - 1], genomes[0][:65536 + k], genomes[1][:70000]]
    modes = [dict(), dict(scale=5), dict(syncmer_s=max(1, k // 2)), dict(minimizer_w=5)]
    if k == 21:
        modes.append(dict(minimizer_w=600))  # halo too large for the LDS tiles: the scratch-buffer form of the workgroup kernel
    for kw in modes:
        spec = lib.SynthSpec(k=k, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=8, num_sigs=1000, kmers_per_col=10, seed=1,
                             scale=kw.get("scale", 1), syncmer_s=kw.get("syncmer_s", 0), minimizer_w=kw.get("minimizer_w", 0))
        cfg = O.sketch_cfg(k=k, **kw)
        with G["Database"].open_synthetic(spec) as db:
            for reads in (short, long_, huge):
                seqs, offs = lib.pack_reads(reads)
                t_seqs = torch.from_numpy(seqs).to(dev)
                t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
                t_h = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
                t_nk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
                p = G["default_params"](min_qlen=0, min_matched=1, dedup_threshold=1 << 30)
                db.kmers_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), max(len(r) for r in reads),
                                t_h.data_ptr(), t_h.numel(), None, t_nk.data_ptr(), params=p)
                torch.cuda.synchronize()
                h = t_h.cpu().numpy().view(np.uint64)
                nk = t_nk.cpu().numpy()
                for i, r in enumerate(reads):
                    want = O.generate_kmers(r, cfg)
                    got = h[int(offs[i]):int(offs[i]) + int(nk[i])]
                    assert len(want) == nk[i], (kw, i, len(r), len(want), nk[i])
                    assert np.array_equal(got, want), (kw, i, len(r))


@pytest.mark.parametrize("k,flags", [(21, "3"), (31, "3"), (64, "3"), (100, "3"), (128, "3"), (21, "19")])
def test_genome_path_two_bit_kernel_and_its_fallback(G, oracle_lib, monkeypatch, k, flags):
    """Round 5: whole genomes go through k1_seg_roll2 (2-bit codes; A / C / G / T in either case) and every 65 536-position segment that
    holds any other byte through the byte kernel k1_seg_roll behind it.  Queries: clean, soft-masked, an N in ONE segment of several
    (its neighbours stay on the fast kernel), N runs across a segment boundary, U, IUPAC codes, a byte >= 128, lengths that end inside a
    16-base group or right at a segment boundary; plain and FracMinHash k-mers.  KMCPG_K1_FLAGS=19 (the byte kernel alone): same hashes."""
    import torch
    O = oracle_lib
    lib = G["lib"]
    monkeypatch.setenv("KMCPG_K1_FLAGS", flags)
    dev = torch.device("cuda:0")
    g = synth.random_genomes(3, 300000, seed=700 + k)
    rng = np.random.default_rng(k)

    def edit(seq, fn):
        b = bytearray(seq)
        fn(b)
        return bytes(b)

    def soft_mask(b):
        for _ in range(30):
            p = int(rng.integers(0, len(b) - 5000))
            n = int(rng.integers(1, 5000))
            b[p:p + n] = bytes(b[p:p + n]).lower()

    def one_n(b):
        b[65536 + 1000] = ord("N")  # inside the second segment only

    def n_runs(b):
        b[65536 - 300:65536 + 700] = b"N" * 1000  # across the boundary of segments 0 and 1
        b[200000:200003] = b"nNn"

    def rna(b):
        b[:] = bytes(b).replace(b"T", b"U")

    def iupac(b):
        for p in rng.integers(0, len(b), size=60):
            b[int(p)] = int(rng.choice(list(b"RYKMSWBDHV-*.") + [200]))

    def n_tail(b):
        b[-700:] = b"N" * 700  # ... and the next query starts with N: as packed input that is ONE run of foreign bytes across two queries

    def n_head(b):
        b[:300] = b"N" * 300
        b[65536 + k - 2] = ord("N")  # the last base segment 0's k-mers cover

    def n_edge(b):
        b[65536 + k - 1] = ord("n")  # the first base they do not: segment 1 alone goes to the byte kernel

    huge = [g[0][:280000], edit(g[1][:250001], soft_mask), edit(g[2][:262144 + k - 1], one_n), edit(g[0][:299990], n_runs), edit(g[1][:70000], rna),
            edit(g[2][:200017], iupac), g[0][:65536 + k], g[1][:65536 * 2 + k - 1], g[2][:131072 + 5], edit(g[0][100:66000], soft_mask),
            edit(g[1][1000:91003], n_tail), b"", edit(g[2][5:140007], n_head), b"NNNN", g[0][7:k + 6], edit(g[1][3:150002], n_edge), g[2][11:3000]]
    for kw in (dict(), dict(scale=7)):
        spec = lib.SynthSpec(k=k, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=8, num_sigs=1000, kmers_per_col=10, seed=1, scale=kw.get("scale", 1))
        cfg = O.sketch_cfg(k=k, **kw)
        with G["Database"].open_synthetic(spec) as db:
            seqs, offs = lib.pack_reads(huge)
            t_seqs = torch.from_numpy(seqs).to(dev)
            t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
            t_h = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
            t_nk = torch.zeros(len(huge), dtype=torch.int32, device=dev)
            p = G["default_params"](min_qlen=0, min_matched=1, dedup_threshold=1 << 30)
            db.kmers_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(huge), len(seqs), max(len(r) for r in huge), t_h.data_ptr(), t_h.numel(), None,
                            t_nk.data_ptr(), params=p)
            torch.cuda.synchronize()
            h = t_h.cpu().numpy().view(np.uint64)
            nk = t_nk.cpu().numpy()
            wants = [O.generate_kmers(r, cfg) for r in huge]
            for i, r in enumerate(huge):
                got = h[int(offs[i]):int(offs[i]) + int(nk[i])]
                assert len(wants[i]) == nk[i], (kw, i, len(r), len(wants[i]), nk[i])
                assert np.array_equal(got, wants[i]), (kw, i, len(r))
            # Round 6: the same batch as 2-bit codes + runs of foreign bytes (what kmcpg_submit_packed uploads).  k1_seg_roll2 takes the codes
            # from the packed stream itself, the segments a run reaches are listed by k_mark_exc and expanded for the byte kernel — a run
            # across two queries, one that ends on the last base a segment's k-mers cover, one that starts right behind it.  Same hashes.
            codes, exc, total = lib.pack2(huge)
            assert total == len(seqs) and len(exc) > 60
            assert any(int(e["pos"]) < int(o) < int(e["pos"]) + int(e["len"]) for e in exc for o in offs[1:-1]), "no run crosses a query boundary"
            pad = np.zeros((total + 3) // 4 + 16, dtype=np.uint8)
            pad[:(total + 3) // 4] = codes[:(total + 3) // 4]
            pad[(total + 3) // 4:] = 0xA5  # (what lies behind the last base must not matter)
            t_codes = torch.from_numpy(pad).to(dev)
            t_exc = torch.from_numpy(exc.view(np.uint8).copy()).to(dev)
            t_text = torch.full((total + 16,), ord("G"), dtype=torch.uint8, device=dev)
            t_h2 = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
            t_nk2 = torch.zeros(len(huge), dtype=torch.int32, device=dev)
            before = db.k1_codes_batches()
            db.kmers_device_packed(t_codes.data_ptr(), t_exc.data_ptr(), len(exc), t_text.data_ptr(), t_offs.data_ptr(), len(huge), total,
                                   max(len(r) for r in huge), t_h2.data_ptr(), t_h2.numel(), None, t_nk2.data_ptr(), params=p)
            torch.cuda.synchronize()
            after = db.k1_codes_batches()
            assert (after[0] - before[0], after[1] - before[1]) == ((1, 0) if flags == "3" else (0, 1)), (before, after)
            h2 = t_h2.cpu().numpy().view(np.uint64)
            nk2 = t_nk2.cpu().numpy()
            for i, r in enumerate(huge):
                assert len(wants[i]) == nk2[i], ("packed", kw, i, len(r), len(wants[i]), nk2[i])
                assert np.array_equal(h2[int(offs[i]):int(offs[i]) + int(nk2[i])], wants[i]), ("packed", kw, i, len(r))
            if flags == "3":  # text exists only where the byte kernel needed it: the first query (clean) was never expanded
                assert bytes(t_text[:1000].cpu().numpy()) == b"G" * 1000


@pytest.mark.parametrize("flags", ["3", "7", "4"])
def test_window_sketch_kernel_forms_on_long_reads(G, oracle_lib, tmp_path, monkeypatch, flags):
    """The three forms the window sketches of long reads can take — the barrier-free wave form (default; KMCPG_K1_FLAGS=3), the
    1024-thread tile form with two-level arg-min and fused adjacent-repeat filter (7; what windows wider than 60 hashes get), the
    tile form with plain scans and a separate k_adj_unique pass (4) — give the oracle's results: single and paired long reads,
    low-complexity sequence, reads at and around the -u / wave-sort bounds, Closed Syncmer and Minimizer databases, and a window
    too wide for the wave form (minimizer w = 100: tile form whatever the flags)."""
    O = oracle_lib
    monkeypatch.setenv("KMCPG_K1_FLAGS", flags)
    genomes = synth.random_genomes(6, 40000, seed=300)
    rng = np.random.default_rng(301)
    lens = [2049, 2100, 2500, 3000, 4097, 6000, 9000, 12000, 20000, 300, 512 + 30, 150]
    # (round 6: closed syncmers with a window of 20 or 32 s-mers - k 21 / s 11, k 31 / s 15 - go through the rolling kernel k1_windows_roll
    # under flags 3 and 7, with k1_windows_wave behind it for what it leaves: reads with an N, short ones, the -u boundary cases)
    for name, kw in (("syn", dict(syncmer_s=11)), ("min", dict(minimizer_w=7)), ("minwide", dict(minimizer_w=100)), ("synscaled", dict(syncmer_s=13, scale=3)),
                     ("syn31", dict(k=31, syncmer_s=15)), ("syn31scaled", dict(k=31, syncmer_s=15, scale=4))):
        kk = kw.pop("k", 21)
        db_dir = synth.make_db(tmp_path / name, genomes, k=kk, n_chunks=2, overlap=150, threads=2, **kw)
        r1 = [synth.sample_reads(genomes, 1, L, sub_rate=0.005, seed=int(rng.integers(1 << 30)), frac_random=0.0)[0] for L in lens]
        r1 += [b"ACGTTGCAAT" * 400, b"A" * 2600 + genomes[0][:2000], genomes[1][:5000] + b"N" * 300 + genomes[1][5000:9000]]
        n, res = _run(G, O, db_dir, r1)
        assert n >= 10, (name, n)
        r2 = [synth.sample_reads(genomes, 1, int(rng.integers(100, 8000)), sub_rate=0.005, seed=int(rng.integers(1 << 30)), frac_random=0.3)[0] for _ in r1]
        n, res = _run(G, O, db_dir, r1, r2, oracle_kw=dict(fpr_buf_size=499), gpu_kw=dict())
        assert n >= 10, (name, n)


def test_dedup_classes_at_their_boundaries(G, oracle_lib):
    """K1d: sort + unique of queries above -u in every size class — one wave (n <= 512), 256-thread workgroup (<= 4096),
    1024-thread workgroup (LDS <= 16384, global memory above), device-wide sort (> 65536) — at the class boundaries, on reads
    full of repeated k-mers; at or below -u the raw list is kept."""
    import torch
    O = oracle_lib
    lib = G["lib"]
    dev = torch.device("cuda:0")
    k = 21
    rng = np.random.default_rng(77)
    unit = synth.random_genomes(1, 400, seed=78)[0]

    def repetitive(n_kmers):
        length = n_kmers + k - 1
        s = bytearray((unit[:137] * (length // 137 + 1))[:length])  # tandem repeat: most k-mers occur several times
        for p in rng.integers(0, length, max(1, length // 60)):
            s[p] = b"ACGT"[int(rng.integers(0, 4))]
        return bytes(s)

    sizes = [1, 9, 64, 255, 256, 257, 300, 511, 512, 513, 1000, 4095, 4096, 4097, 9000, 16384, 16385, 20000]
    reads = [repetitive(n) for n in sizes] + [synth.random_genomes(1, n + k - 1, seed=79 + n)[0] for n in (260, 512, 513, 4096, 4097, 9000, 16384, 16385)]
    # no spread at all: every k-mer of a homopolymer / short tandem repeat falls into one or two buckets of the distribution sort
    reads += [b"A" * 3000, b"AC" * 1500, b"ACG" * 700, b"ACGTTGCA" * 500 + synth.random_genomes(1, 1500, seed=99)[0], b"AC" * 5000,
              b"ACGTTGCA" * 900 + synth.random_genomes(1, 6000, seed=98)[0]]
    for sk in (0, 11):  # plain k-mers, and Closed Syncmers (runs of equal emissions: the fused adjacent-repeat filter of long reads)
        spec = lib.SynthSpec(k=k, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=8, num_sigs=1000, kmers_per_col=10, seed=1, syncmer_s=sk)
        cfg = O.sketch_cfg(k=k, syncmer_s=sk)
        with G["Database"].open_synthetic(spec) as db:
            for thr in (0, 256, 600):
                for batch in (reads, reads[:9], reads + [repetitive(70000)]):  # max length decides which classes are launched
                    seqs, offs = lib.pack_reads(batch)
                    t_seqs = torch.from_numpy(seqs).to(dev)
                    t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
                    t_h = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
                    t_nk = torch.zeros(len(batch), dtype=torch.int32, device=dev)
                    p = G["default_params"](min_qlen=0, min_matched=1, dedup_threshold=thr)
                    db.kmers_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(batch), len(seqs), max(len(r) for r in batch),
                                    t_h.data_ptr(), t_h.numel(), None, t_nk.data_ptr(), params=p)
                    torch.cuda.synchronize()
                    h = t_h.cpu().numpy().view(np.uint64)
                    nk = t_nk.cpu().numpy()
                    for i, r in enumerate(batch):
                        raw = O.generate_kmers(r, cfg)
                        want = O.sort_unique(raw) if len(raw) > thr else raw
                        got = h[int(offs[i]):int(offs[i]) + int(nk[i])]
                        assert nk[i] == len(want), (sk, thr, i, len(raw), nk[i], len(want))
                        assert np.array_equal(got, want), (sk, thr, i, len(raw))
    # FracMinHash sketches of whole genomes (what the genome search sorts: 4 096 < m <= 16 384 hashes, all below maxHash — the distribution
    # pass of the 1024-thread class shifts the largest possible hash up to bit 63 first), around that class's bounds, with duplicated stretches
    g = synth.random_genomes(3, 300000, seed=801)
    whole = [g[0][:70000] + g[0][:40000], g[1][:200000], g[2][:262000] + g[2][:30000], g[0][:66000], g[1][:131072] + g[1][1000:131072]]
    raws = []
    for scale in (48, 30):
        spec = lib.SynthSpec(k=k, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=8, num_sigs=1000, kmers_per_col=10, seed=1, scale=scale)
        cfg = O.sketch_cfg(k=k, scale=scale)
        with G["Database"].open_synthetic(spec) as db:
            seqs, offs = lib.pack_reads(whole)
            t_seqs = torch.from_numpy(seqs).to(dev)
            t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
            t_h = torch.zeros(len(seqs) + 8, dtype=torch.int64, device=dev)
            t_nk = torch.zeros(len(whole), dtype=torch.int32, device=dev)
            db.kmers_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(whole), len(seqs), max(len(r) for r in whole), t_h.data_ptr(), t_h.numel(), None,
                            t_nk.data_ptr(), params=G["default_params"](min_qlen=0, min_matched=1, dedup_threshold=256))
            torch.cuda.synchronize()
            h = t_h.cpu().numpy().view(np.uint64)
            nk = t_nk.cpu().numpy()
            for i, r in enumerate(whole):
                raw = O.generate_kmers(r, cfg)
                want = O.sort_unique(raw)
                raws.append(len(raw))
                assert len(want) < len(raw) or i in (1, 3)  # (the others hold a stretch twice)
                assert nk[i] == len(want), (scale, i, nk[i], len(want))
                assert np.array_equal(h[int(offs[i]):int(offs[i]) + int(nk[i])], want), (scale, i)
    assert sum(1 for m in raws if 4096 < m <= 16384) >= 4 and max(raws) > 16384 and min(raws) <= 4096, raws


def test_batch_larger_than_one_launch(G):
    """300 k reads x 128 slots = 9.6 M workgroups of k2_cobs: more than one launch may hold (2^32 threads) — the engine splits
    it; the hit list must equal the union of the hit lists of the two half batches."""
    import torch
    lib = G["lib"]
    dev = torch.device("cuda:0")
    spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=128, cols_per_block=2100, num_sigs=40009, kmers_per_col=14270, seed=5)
    B, L = 300000, 150
    with G["Database"].open_synthetic(spec) as db:
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        reads = acgt[torch.randint(0, 4, (B, L), generator=g, device=dev)].contiguous().view(-1)
        cols = torch.randint(0, 128 * 2100, (B,), generator=g, device=dev).to(torch.int32)
        cols[torch.rand(B, generator=g, device=dev) < 0.5] = -1
        offs = (torch.arange(B + 1, device=dev, dtype=torch.int64) * L).contiguous()
        db.plant_reads_device(reads.data_ptr(), offs.data_ptr(), B, B * L, L, cols.data_ptr())

        def query(lo, hi):
            n = hi - lo
            cap = 64 * n
            hits = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            qk = torch.zeros(n, dtype=torch.int32, device=dev)
            ql = torch.zeros(n, dtype=torch.int32, device=dev)
            sub = reads[lo * L:hi * L]
            db.query_device(sub.data_ptr(), offs.data_ptr(), n, n * L, L, hits.data_ptr(), cap, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(),
                            params=G["default_params"]())
            torch.cuda.synchronize()
            c = int(cnt[0].item())
            assert c <= cap
            h = hits[:c].cpu().numpy().astype(np.int64)
            h[:, 0] += lo
            return h[np.lexsort((h[:, 1], h[:, 0]))]

        whole = query(0, B)
        halves = np.concatenate([query(0, B // 2), query(B // 2, B)])
        assert np.array_equal(whole, halves)
        planted = np.nonzero(cols.cpu().numpy() >= 0)[0]
        got = set(zip(whole[:, 0].tolist(), whole[:, 1].tolist()))
        c_host = cols.cpu().numpy()
        assert all((int(r), int(c_host[r])) in got for r in planted[-2000:])  # the reads of the last launch are served too


def test_more_shards_than_blocks(G, oracle_lib, tmp_path):
    """A one-block database opened over three shards (in process) and as shard 2 of 3 (one process per GPU): shards without
    any resident block take part and contribute nothing."""
    import torch
    O = oracle_lib
    genomes = synth.random_genomes(6, 8000, seed=85)
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=1)
    reads = synth.sample_reads(genomes, 300, 150, seed=86, frac_random=0.1)
    odb = O.OracleDB(db_dir)
    try:
        with G["Database"].open_devices(db_dir, [0, 0, 0]) as db:
            assert db.info.n_blocks == 1
            res = db.search(reads, params=G["default_params"]())
            assert synth.assert_parity(odb, res, reads) > 200
    finally:
        odb.close()
    dev = torch.device("cuda:0")
    seqs, offs = G["lib"].pack_reads(reads)
    t_seqs = torch.from_numpy(seqs).to(dev)
    t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    total = 0
    for rank in range(3):
        with G["Database"].open(db_dir, device=0, shard_rank=rank, shard_count=3) as sh:
            hits = torch.zeros((4096, 3), dtype=torch.int32, device=dev)
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            qk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
            ql = torch.zeros(len(reads), dtype=torch.int32, device=dev)
            sh.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), 150, hits.data_ptr(), 4096, cnt.data_ptr(), qk.data_ptr(),
                            ql.data_ptr(), params=G["default_params"]())
            torch.cuda.synchronize()
            c = int(cnt[0].item())
            assert (c > 0) == (sh.info.n_blocks_local == 1)
            assert int(qk.sum().item()) > 0  # k-mers are generated on every shard
            total += c
    assert total > 200


def test_hit_buffer_overflow_is_retried(G, oracle_lib, tmp_path):
    """-t just above the Bloom density: half of all columns pass the threshold by chance, hundreds of hits per read — far more
    than the first hit buffer of kmcpg_search_batch (8 per read); the call must notice the overflow, rerun with room for every
    hit and still agree with the oracle; the same for a caller-sized buffer of kmcpg_query_device (count reported, no write
    past the capacity)."""
    import torch
    O = oracle_lib
    genomes = synth.random_genomes(600, 3000, seed=95)
    db_dir = synth.make_db(tmp_path, genomes, k=21, fpr=0.3, block_size=600)
    reads = synth.sample_reads(genomes, 150, 150, seed=96, frac_random=0.5)
    kw = dict(min_qcov=0.31, max_fpr=1.0, min_matched=1)
    n, res = _run(G, O, db_dir, reads, oracle_kw=kw, gpu_kw=kw)
    total = int(res.offs[-1])
    assert n == total > 50 * len(reads)  # every match compared; way beyond 8 hits per read
    # device-level: capacity 100, guard words behind it stay untouched, the counter reports the real number
    dev = torch.device("cuda:0")
    seqs, offs = G["lib"].pack_reads(reads)
    t_seqs = torch.from_numpy(seqs).to(dev)
    t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    with G["Database"].open(db_dir, device=0) as db:
        hits = torch.full((200, 3), -7, dtype=torch.int32, device=dev)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        qk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
        ql = torch.zeros(len(reads), dtype=torch.int32, device=dev)
        db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), 150, hits.data_ptr(), 100, cnt.data_ptr(), qk.data_ptr(),
                        ql.data_ptr(), params=G["default_params"](**kw))
        torch.cuda.synchronize()
        assert int(cnt[0].item()) >= total  # hits before the float64 filters of the host half
        h = hits.cpu().numpy()
        assert (h[100:] == -7).all() and (h[:100, 2] > 0).all()


def test_degenerate_batches(G, oracle_lib, tmp_path):
    """Zero reads, only empty reads, an empty mate, no hit buffer at all (hit_cap 0: count only)."""
    import torch
    O = oracle_lib
    genomes = synth.random_genomes(8, 4000, seed=97)
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=2)
    dev = torch.device("cuda:0")
    with G["Database"].open(db_dir, device=0) as db:
        res = db.search([], params=G["default_params"]())
        assert len(res) == 0 and len(res.matches) == 0
        res = db.search([b"", b"", b""], params=G["default_params"]())
        assert list(res.qlen) == [0, 0, 0] and len(res.matches) == 0
        r1 = [genomes[0][:150], b"", genomes[1][100:250]]
        r2 = [b"", genomes[2][:150], genomes[1][300:450]]
        odb = O.OracleDB(db_dir)
        try:
            res = db.search(r1, r2, params=G["default_params"](try_se=1))
            assert synth.assert_parity(odb, res, r1, r2, O.default_params(try_se=1, fpr_buf_size=499)) >= 2
        finally:
            odb.close()
        reads = synth.sample_reads(genomes, 64, 150, seed=98, frac_random=0.0)
        seqs, offs = G["lib"].pack_reads(reads)
        t_seqs = torch.from_numpy(seqs).to(dev)
        t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        qk = torch.zeros(64, dtype=torch.int32, device=dev)
        ql = torch.zeros(64, dtype=torch.int32, device=dev)
        db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), 64, len(seqs), 150, None, 0, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(),
                        params=G["default_params"]())
        torch.cuda.synchronize()
        assert int(cnt[0].item()) >= 64 and int(qk.min().item()) == 130


def test_fpr_bound_on_the_device_changes_nothing_but_the_raw_hit_list(G, oracle_lib, tmp_path):
    """-t just above the database's FPR (0.3): counts of 41..51 of 130 k-mers pass the coverage threshold in ~40 % of all
    columns by chance, and every one of them fails -f 0.01 on the host.  The kernel leaves them out (fpr_bound in query.cpp):
    same finalized result as with KMCPG_FPR_BOUND=0 and as the oracle's, a fraction of the raw hits; paired reads (n = 260),
    a stricter and a lax -f too."""
    import os
    import torch
    O = oracle_lib
    genomes = synth.random_genomes(64, 4000, seed=51)  # equal sizes: every column is as dense as the block allows (~0.3)
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=2)
    reads = synth.sample_reads(genomes, 400, 150, sub_rate=0.02, seed=52, frac_random=0.5)
    reads2 = synth.sample_reads(genomes, 400, 150, sub_rate=0.02, seed=53, frac_random=0.5)
    odb = O.OracleDB(db_dir)
    dev = torch.device("cuda:0")
    try:
        with G["Database"].open(db_dir, device=0) as db:
            long_reads = synth.sample_reads(genomes, 60, 700, sub_rate=0.02, seed=54, frac_random=0.5)  # 680 k-mers: the table grows past 512
            short = reads
            for kw, r2, reads in ((dict(min_qcov=0.31), None, short), (dict(min_qcov=0.31, max_fpr=1e-6), None, short),
                                  (dict(min_qcov=0.31, max_fpr=0.9), None, short), (dict(min_qcov=0.31, fpr_buf_size=499), reads2, short),
                                  (dict(min_qcov=0.31), None, long_reads), (dict(min_qcov=0.31), None, short)):
                res, raw = {}, {}
                for bound in ("1", "0"):
                    os.environ["KMCPG_FPR_BOUND"] = bound
                    try:
                        res[bound] = db.search(reads, r2, params=G["default_params"](**kw))
                        if r2 is None:
                            seqs, offs = G["lib"].pack_reads(reads)
                            t_seqs = torch.from_numpy(seqs).to(dev)
                            t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
                            cap = 64 * len(reads)
                            hits = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
                            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
                            qk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
                            ql = torch.zeros(len(reads), dtype=torch.int32, device=dev)
                            db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), max(len(r) for r in reads), hits.data_ptr(), cap, cnt.data_ptr(), qk.data_ptr(),
                                            ql.data_ptr(), params=G["default_params"](**kw))
                            torch.cuda.synchronize()
                            raw[bound] = int(cnt[0].item())
                    finally:
                        os.environ.pop("KMCPG_FPR_BOUND", None)
                assert np.array_equal(res["1"].matches, res["0"].matches) and np.array_equal(res["1"].offs, res["0"].offs), kw
                assert synth.assert_parity(odb, res["1"], reads, r2, O.default_params(**kw)) > len(reads) // 4
                if r2 is None:
                    assert raw["1"] <= raw["0"]
                    if kw.get("max_fpr", 0.01) <= 0.01:
                        assert raw["1"] < 0.5 * raw["0"], (kw, raw)  # most chance columns never leave the GPU
                    else:
                        assert raw["1"] == raw["0"]  # -f 0.9: the coverage threshold is the stricter one again
    finally:
        odb.close()


def test_threshold_at_the_top_of_a_plane_class(G, oracle_lib, tmp_path):
    """Reads of exactly 254 / 255 / 256 and 1022 / 1023 / 1024 k-mers, searched with -t just below 1 (every k-mer has to match) and
    with -t 1 (no count can pass: float64(count) > n * 1.0 never holds, util-db-search.go:7468-7470).  The thresholds are n and
    n + 1: the kernel compares counts with them on NPL bits, so the host puts n = 255 / 1 023 into the next plane class
    (query.cpp); with -t 1 the raw hit list must be empty, not "every column with a count"."""
    import torch
    O = oracle_lib
    lib = G["lib"]
    genomes = synth.random_genomes(12, 6000, seed=71)
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=2)
    reads = []
    for n in (254, 255, 256, 1022, 1023, 1024):
        for gi in (0, 5):
            reads.append(genomes[gi][300:300 + n + 20])       # exact: count == n
            a = bytearray(genomes[gi][900:900 + n + 20])
            a[len(a) // 2] = ord("A") if a[len(a) // 2] != ord("A") else ord("C")  # one substitution: 21 k-mers short of n
            reads.append(bytes(a))
    for t in (0.9995, 1.0):
        kw = dict(min_qcov=t)
        n, res = _run(G, O, db_dir, reads, oracle_kw=kw, gpu_kw=kw)
        assert [int(x) for x in res.qkmers[::2]] == [254, 254, 255, 255, 256, 256, 1022, 1022, 1023, 1023, 1024, 1024]
        assert n == (12 if t < 1 else 0)
    # the raw hit list of the GPU half at -t 1
    dev = torch.device("cuda:0")
    seqs, offs = lib.pack_reads(reads)
    t_seqs = torch.from_numpy(seqs).to(dev)
    t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    nr = len(reads)
    with G["Database"].open(db_dir, device=0) as db:
        hits = torch.zeros((4096, 3), dtype=torch.int32, device=dev)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        qk = torch.zeros(nr, dtype=torch.int32, device=dev)
        ql = torch.zeros(nr, dtype=torch.int32, device=dev)
        db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), nr, int(offs[-1]), max(len(r) for r in reads), hits.data_ptr(), 4096, cnt.data_ptr(),
                        qk.data_ptr(), ql.data_ptr(), params=G["default_params"](min_qcov=1.0))
        torch.cuda.synchronize()
        assert int(cnt[0].item()) == 0


@pytest.mark.parametrize("ncols,nh,split", [(6500, 3, "1"), (6500, 1, "2"), (3000, 1, "0"), (3900, 2, "0"), (5000, 1, None)])
def test_row_remainders_cut_into_power_of_two_tiles(G, oracle_lib, tmp_path, monkeypatch, ncols, nh, split):
    """Round 5: what is left of a row beyond its whole KiB tiles, 257..896 bytes, is cut into 512 / 256 / 128 / 64-byte tiles (32, 16, 8,
    4 lanes per unit) under KMCPG_SPLIT_TILES=1 (multi-hash databases) instead of one 64-lane tile with idle lanes: 6 500 columns = 813-byte
    rows -> 512 + 256 + 64 (an experiment that lost, profiles/r05_split_tiles.txt: the knob is off by default).  KMCPG_SPLIT_TILES=2 does the same
    to single-hash databases, 0 (the default) switches it off - and then 3 000 / 3 900 columns (375- / 488-byte rows) sit on ONE 32-lane
    tile, the default for 257..512-byte remainders since round 5: same results everywhere.  Short reads (8 planes) and long queries (16 planes).
    Round 6: with the knob unset a single-hash database whose remainder is exactly 640 bytes (5 000 columns = 625-byte rows) gets 512 + 128
    by itself (+11-17 %, profiles/r06_lpr_640.txt)."""
    O = oracle_lib
    if split is None:
        monkeypatch.delenv("KMCPG_SPLIT_TILES", raising=False)
    else:
        monkeypatch.setenv("KMCPG_SPLIT_TILES", split)
    genomes = synth.random_genomes(ncols + 21, 420, seed=900 + ncols + nh)
    db_dir = synth.make_db(tmp_path, genomes, k=21, num_hashes=nh, fpr=0.05 if nh > 1 else 0.3, block_size=ncols, threads=4)
    reads = synth.sample_reads(genomes, 500 if split is not None else 200, 150, sub_rate=0.01, seed=7, frac_random=0.1)
    # long queries: several genomes back to back (~1 600 distinct k-mers: the sort + unique path and 16 counter planes)
    rng = np.random.default_rng(11)
    longq = [b"".join(genomes[int(j)] for j in rng.integers(0, len(genomes), size=4)) for _ in range(40 if split is not None else 12)]
    kw = dict(min_qcov=0.2, min_matched=5)
    n, res = _run(G, O, db_dir, reads + longq, oracle_kw=kw, gpu_kw=kw)
    assert n > (400 if split is not None else 150) and int(res.qkmers.max()) > 1024
    if split is None:  # the default rule really cut the row: two slot classes (32-lane + 8-lane), visible as two tiles of hash traffic
        import torch
        with G["Database"].open(db_dir, device=0) as db:
            assert db.block_info(0)["stride"] == 640
            db.set_profiling(2)
            seqs, offs = G["lib"].pack_reads(reads[:64])
            dev = torch.device("cuda:0")
            t_seqs, t_offs = torch.from_numpy(seqs).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
            hits = torch.zeros((4096, 3), dtype=torch.int32, device=dev)
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            qk = torch.zeros(64, dtype=torch.int32, device=dev)
            ql = torch.zeros(64, dtype=torch.int32, device=dev)
            monkeypatch.setenv("KMCPG_PRUNE", "0")
            db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), 64, len(seqs), 150, hits.data_ptr(), 4096, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(),
                            params=G["default_params"]())
            torch.cuda.synchronize()
            assert db.last_hash_bytes() == 8 * int(qk.sum().item()) * 2  # every read's hashes fetched once per slot: 2 slots
