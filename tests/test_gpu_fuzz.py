"""Randomised parity sweep: database shape, k, hash count, sketch mode and search flags drawn from a seeded generator;
every draw must give bit-identical per-read results on the GPU and in the oracle."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _draw(rng):
    k = int(rng.choice([11, 15, 21, 25, 31, 32, 33, 47, 63, 64, 65, 66, 80]))
    mode = rng.choice(["plain", "plain", "scaled", "syncmer", "minimizer"])
    kw = {}
    if mode == "scaled":
        kw["scale"] = int(rng.choice([2, 5, 16]))
    elif mode == "syncmer":
        kw["syncmer_s"] = int(rng.integers(max(1, k - 12), k + 1))
        if rng.random() < 0.5:
            kw["scale"] = int(rng.choice([2, 4]))
    elif mode == "minimizer":
        kw["minimizer_w"] = int(rng.integers(1, 25))
    nh = int(rng.choice([1, 1, 2, 3, 4]))
    fpr = float(rng.choice([0.3, 0.1, 0.01])) if nh == 1 else float(rng.choice([0.05, 0.01]))
    n_genomes = int(rng.choice([5, 40, 300, 1500]))
    glen = int(rng.choice([600, 2500, 9000])) if n_genomes <= 300 else 500
    n_chunks = int(rng.choice([1, 1, 3]))
    threads = int(rng.choice([1, 2, 8, 32]))
    return k, kw, nh, fpr, n_genomes, glen, n_chunks, threads


def _check_packed(db, res, reads, reads2, params):
    """KMCP_FUZZ_PACKED=1 (round 6): the same single-end batch through kmcpg_pack2 + kmcpg_submit_packed — once from ordinary memory, once from
    kmcpg_host_alloc memory (uploaded in place) — must give the records of the text entry bit for bit"""
    if not os.environ.get("KMCP_FUZZ_PACKED") or reads2 is not None:
        return
    from kmcp_amd import lib
    codes, exc, total = lib.pack2(reads)
    _, offs = lib.pack_reads(reads)
    assert total == int(offs[-1])
    a = db.wait(db.submit_packed(codes, offs, exc, params=params))
    with lib.PinnedBytes(len(codes)) as pin:
        pin.a[:len(codes)] = codes
        b = db.wait(db.submit_packed(pin.a, offs, exc, params=params))
    for r in (a, b):
        for f in ("qlen", "qkmers", "ksize", "offs"):
            assert np.array_equal(getattr(res, f), getattr(r, f)), f
        assert r.matches.tobytes() == res.matches.tobytes()


def _check_pairs(db, res, reads, reads2, params):
    """KMCP_FUZZ_PAIRS=1: the compact result of the same search, expanded query by query, must be the records bit for bit"""
    _check_packed(db, res, reads, reads2, params)
    if not os.environ.get("KMCP_FUZZ_PAIRS"):
        return
    pr = db.search_pairs(reads, reads2, params=params)
    for f in ("qlen", "qkmers", "ksize", "offs"):
        assert np.array_equal(getattr(res, f), getattr(pr, f)), f
    parts = [db.expand_pairs(int(pr.qkmers[i]), pr.read(i)) for i in range(len(pr)) if pr.offs[i + 1] > pr.offs[i]]
    got = np.concatenate(parts).tobytes() if parts else b""
    assert got == res.matches.tobytes()


# KMCP_FUZZ_SEEDS=N / KMCP_FUZZ_LONG_SEEDS=N widen the sweeps (32 / 8 by default; the final round-1 build passed a soak run of
# 20 000 + 6 000 seeds on MI355X: `pytest tests/test_gpu_fuzz.py -m gpu -n 14`, 3 minutes)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("KMCP_FUZZ_SEEDS", "32")))))
def test_random_configuration(oracle_lib, tmp_path, seed):
    from kmcp_amd import Database, default_params
    O = oracle_lib
    rng = np.random.default_rng(1000 + seed)
    k, kw, nh, fpr, n_genomes, glen, n_chunks, threads = _draw(rng)
    genomes = synth.random_genomes(n_genomes, glen, seed=2000 + seed)
    db_dir = synth.make_db(tmp_path, genomes, k=k, n_chunks=n_chunks, overlap=min(150, glen // 8), num_hashes=nh, fpr=fpr, threads=threads, **kw)
    # ragged reads: 20 .. 400 bases, some N, some shorter than k / than -m
    reads = []
    for i in range(250):
        L = int(rng.integers(20, 400))
        reads += synth.sample_reads(genomes, 1, min(L, glen - 1), sub_rate=float(rng.choice([0, 0.01, 0.05])), seed=int(rng.integers(1 << 30)),
                                    frac_random=0.15, n_rate=float(rng.choice([0, 0, 0.01])))
    reads += [b"", b"A" * 25, genomes[0][:k], genomes[0][:k - 1]]
    paired = rng.random() < 0.3
    reads2 = None
    if paired:
        reads2 = [synth.sample_reads(genomes, 1, max(1, min(len(r), glen - 1)), seed=int(rng.integers(1 << 30)), frac_random=0.4)[0] if len(r) else b""
                  for r in reads]
    t = float(rng.choice([0.55, 0.4, 0.7, 0.9]))
    t = max(t, fpr + 0.05)
    flags = dict(min_qcov=t, min_matched=int(rng.choice([1, 3, 10, 30])), min_qlen=int(rng.choice([0, 30, 70])), max_fpr=float(rng.choice([0.01, 0.05, 1e-6])),
                 min_tcov=float(rng.choice([0, 0, 0.01])), dedup_threshold=int(rng.choice([256, 256, 64, 100000])), sort_by=int(rng.integers(0, 3)),
                 top_n_scores=int(rng.choice([0, 0, 1, 2])), try_se=int(paired and rng.random() < 0.5))
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as db:
            res = db.search(reads, reads2, params=default_params(**flags))
            _check_pairs(db, res, reads, reads2, default_params(**flags))
        synth.assert_parity(odb, res, reads, reads2, O.default_params(**flags))
    finally:
        odb.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("KMCP_FUZZ_LONG_SEEDS", "8")))))
def test_random_long_queries(oracle_lib, tmp_path, seed):
    """Long queries (1 kb .. 200 kb: HiFi reads, contigs, small genomes) mixed with short ones in one batch: the workgroup
    and segment forms of K1, every dedup class, 16/24-plane counters and the SPLIT form of the COBS kernel, in all sketch
    modes, with -u at its default, tiny or off."""
    from kmcp_amd import Database, default_params
    O = oracle_lib
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.choice([15, 21, 31, 33, 64, 66]))
    mode = rng.choice(["plain", "scaled", "syncmer", "minimizer"])
    kw = {}
    if mode == "scaled":
        kw["scale"] = int(rng.choice([3, 20, 200]))
    elif mode == "syncmer":
        kw["syncmer_s"] = int(rng.integers(max(1, k - 14), k + 1))
    elif mode == "minimizer":
        kw["minimizer_w"] = int(rng.integers(1, 40))
    nh = int(rng.choice([1, 1, 3]))
    fpr = 0.3 if nh == 1 else 0.01
    n_genomes = int(rng.choice([6, 30]))
    glen = int(rng.choice([30000, 220000])) if n_genomes == 6 else 30000
    genomes = synth.random_genomes(n_genomes, glen, seed=6000 + seed)
    db_dir = synth.make_db(tmp_path, genomes, k=k, n_chunks=int(rng.choice([1, 4])), overlap=150, num_hashes=nh, fpr=fpr, threads=int(rng.choice([1, 4])), **kw)
    reads = []
    for _ in range(int(rng.integers(3, 14))):
        L = int(min(glen - 1, rng.choice([1000, 2047, 2048, 2049, 5000, 12000, 29999, 70000, 200000])))
        reads += synth.sample_reads(genomes, 1, L, sub_rate=float(rng.choice([0, 0.01, 0.1])), seed=int(rng.integers(1 << 30)), frac_random=0.1,
                                    n_rate=float(rng.choice([0, 0.001])))
    reads += synth.sample_reads(genomes, 20, 150, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.3:
        reads.append(genomes[0] + genomes[1][: glen // 2])  # longer than any reference
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    t = max(float(rng.choice([0.55, 0.3, 0.8])), fpr + 0.05)
    flags = dict(min_qcov=t, min_matched=int(rng.choice([1, 10])), dedup_threshold=int(rng.choice([256, 0, 1 << 30])), sort_by=int(rng.integers(0, 3)),
                 max_fpr=float(rng.choice([0.01, 1.0])))
    # every fourth draw is paired: long (or short, or empty) mates go through the second-mate code of the workgroup kernels
    reads2 = None
    if seed % 4 == 3:
        reads2 = []
        for r in reads:
            L2 = int(min(glen - 1, rng.choice([0, 150, 3000, 9000])))
            reads2.append(synth.sample_reads(genomes, 1, L2, sub_rate=0.01, seed=int(rng.integers(1 << 30)), frac_random=0.3)[0] if L2 else b"")
        flags["try_se"] = int(rng.random() < 0.5)
    oflags = dict(flags, fpr_buf_size=499) if reads2 is not None else flags
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as db:
            res = db.search(reads, reads2, params=default_params(**flags))
            _check_pairs(db, res, reads, reads2, default_params(**flags))
        # (a few draws — large scale, high -t — legitimately have no matching read on either side)
        synth.assert_parity(odb, res, reads, reads2, O.default_params(**oflags))
    finally:
        odb.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("KMCP_FUZZ_WIDE_SEEDS", "2")))))
def test_random_mid_width_rows(oracle_lib, tmp_path, seed):
    """Rows of 257..512 bytes (2 049 .. 4 096 columns per block: the 32-lane form of k2_cobs, round 5) with random widths, hash counts,
    k and thresholds, a second ragged block beside the wide one; short reads and long queries against the oracle."""
    from kmcp_amd import Database, default_params
    O = oracle_lib
    rng = np.random.default_rng(9000 + seed)
    ncols = int(rng.integers(2049, 4097))
    extra = int(rng.choice([0, 5, 300, 2600]))
    nh = int(rng.choice([1, 1, 2, 3]))
    k = int(rng.choice([21, 31, 64]))
    fpr = 0.3 if nh == 1 else float(rng.choice([0.05, 0.01]))
    # (the ragged second block holds longer genomes: another NumSigs, so the two are not laid side by side into one wider row)
    genomes = synth.random_genomes(ncols, 380, seed=9500 + seed) + synth.random_genomes(extra, 520, seed=9700 + seed)
    db_dir = synth.make_db(tmp_path, genomes, k=k, num_hashes=nh, fpr=fpr, block_size=ncols, threads=4)
    reads = synth.sample_reads(genomes, 300, 150, sub_rate=float(rng.choice([0, 0.01, 0.03])), seed=int(rng.integers(1 << 30)), frac_random=0.15)
    longq = [b"".join(genomes[int(j)] for j in rng.integers(0, len(genomes), size=int(rng.integers(2, 9)))) for _ in range(25)]
    t = max(float(rng.choice([0.55, 0.3, 0.15])), fpr + 0.05) if nh > 1 else float(rng.choice([0.55, 0.4]))
    flags = dict(min_qcov=t if nh == 1 else min(t, 0.55), min_matched=int(rng.choice([1, 5, 10])), sort_by=int(rng.integers(0, 3)), top_n_scores=int(rng.choice([0, 0, 2])))
    if nh > 1:
        flags["min_qcov"] = float(rng.choice([0.12, 0.3, 0.55]))
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as db:
            assert any(256 < db.block_info(b)["stride"] <= 512 for b in range(db.info.n_blocks))
            res = db.search(reads + longq, params=default_params(**flags))
            _check_pairs(db, res, reads + longq, None, default_params(**flags))
        synth.assert_parity(odb, res, reads + longq, None, O.default_params(**flags))
    finally:
        odb.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("KMCP_FUZZ_TAIL_SEEDS", "3")))))
def test_random_wide_rows_long_queries(oracle_lib, tmp_path, seed, monkeypatch):
    """Rows of 513 .. 1 150 bytes (a 64-lane tile, with or without a remainder beside it) and queries of 1 000 .. 6 000 k-mers that are made
    of a few whole references: most sectors of a tile die, the ones that hold the query's parts do not — the tail mode of the 16-plane
    kernels (round 6), with 1 .. 4 (or more) live sectors, 1 .. 3 hash functions, default and forced settings — against the oracle."""
    from kmcp_amd import Database, default_params
    O = oracle_lib
    rng = np.random.default_rng(12000 + seed)
    ncols = int(rng.integers(4100, 9200))
    extra = int(rng.choice([0, 40, 700]))
    nh = int(rng.choice([1, 1, 3]))
    k = int(rng.choice([21, 31]))
    fpr = 0.3 if nh == 1 else float(rng.choice([0.05, 0.01]))
    glen = int(rng.choice([420, 700]))
    genomes = synth.random_genomes(ncols, glen, seed=12500 + seed) + synth.random_genomes(extra, glen + 160, seed=12700 + seed)
    db_dir = synth.make_db(tmp_path, genomes, k=k, num_hashes=nh, fpr=fpr, block_size=ncols, threads=4)
    longq = []
    for _ in range(40):
        m = int(rng.choice([2, 2, 3, 4, 6, 9]))
        first = int(rng.integers(0, len(genomes)))
        # parts from one neighbourhood of columns (one or two sectors) or from anywhere
        near = rng.random() < 0.5
        ids = [(first + int(rng.integers(0, 900))) % len(genomes) if near else int(rng.integers(0, len(genomes))) for _ in range(m)]
        longq.append(b"".join(genomes[j] for j in ids))
    reads = synth.sample_reads(genomes, 60, 150, sub_rate=0.01, seed=int(rng.integers(1 << 30)), frac_random=0.15)
    # one hash function at fpr 0.3: a part of an m-part query reaches ~1/m + 0.3 (1 - 1/m) of the k-mers, an unrelated column ~0.3
    t = float(rng.choice([0.34, 0.45])) if nh == 1 else float(rng.choice([0.08, 0.2, 0.3, 0.45]))
    flags = dict(min_qcov=t, min_matched=int(rng.choice([1, 10])), sort_by=int(rng.integers(0, 3)))
    if rng.random() < 0.5:
        monkeypatch.setenv("KMCPG_SPLIT_MIN", "0")      # every query on the plain kernel, whatever its length
    knob = rng.choice(["default", "4", "1"])
    if knob != "default":
        monkeypatch.setenv("KMCPG_TAIL_SECTORS", str(knob))
        monkeypatch.setenv("KMCPG_TAIL_MIN", "1")
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as db:
            assert any(db.block_info(b)["stride"] > 512 for b in range(db.info.n_blocks))
            if os.environ.get("KMCP_FUZZ_TAIL_REPORT"):  # how many waves of this draw finished in tail mode (soak runs: the mode must not sit idle)
                db.set_profiling(2)
            res = db.search(longq + reads, params=default_params(**flags))
            if os.environ.get("KMCP_FUZZ_TAIL_REPORT"):
                with open(os.environ["KMCP_FUZZ_TAIL_REPORT"], "a") as f:
                    f.write("%d %s nh=%d t=%.2f tail_waves=%d\n" % (seed, knob, nh, t, db.last_tail_waves()))
                db.set_profiling(0)
            _check_pairs(db, res, longq + reads, None, default_params(**flags))
        synth.assert_parity(odb, res, longq + reads, None, O.default_params(**flags))
    finally:
        odb.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("KMCP_FUZZ_ROLL_SEEDS", "4")))))
def test_random_long_syncmer_reads(oracle_lib, tmp_path, seed):
    """Closed-Syncmer databases whose window is 12 / 16 / 20 / 24 / 32 s-mers (k - s = 6 .. 16: what k1_windows_roll, round 6, takes) searched with long reads
    of random lengths from just above its smallest window count to 30 kb: the lanes' run length, the number of idle lanes, the position of the
    block boundaries and of the read's end relative to them all vary; a fraction of the reads is soft-masked, carries an N (those go to
    k1_windows_wave), repeats itself (runs of equal emissions across lanes) or is shorter than the kernel takes; FracMinHash on top in half
    of the draws; -u at its default, tiny or off.  Everything against the oracle."""
    from kmcp_amd import Database, default_params
    O = oracle_lib
    rng = np.random.default_rng(12000 + seed)
    d = int(rng.choice([6, 8, 10, 12, 16, 10, 16]))
    k = int(rng.choice([21, 25, 31, 32, 40]))
    if k - d < 1:
        k = d + 11
    kw = dict(syncmer_s=k - d)
    if rng.random() < 0.5:
        kw["scale"] = int(rng.choice([2, 7]))
    genomes = synth.random_genomes(6, 32000, seed=13000 + seed)
    db_dir = synth.make_db(tmp_path, genomes, k=k, n_chunks=int(rng.choice([1, 3])), overlap=150, threads=2, **kw)
    reads = []
    Lw = 2 * k - (k - d) - 1
    for _ in range(int(rng.integers(6, 20))):
        L = int(rng.choice([Lw + 1023, Lw + 1024, int(rng.integers(1100, 4000)), int(rng.integers(4000, 12000)), int(rng.integers(12000, 30000))]))
        r = bytearray(synth.sample_reads(genomes, 1, min(L, 31999), sub_rate=float(rng.choice([0, 0.001, 0.02])), seed=int(rng.integers(1 << 30)), frac_random=0.1)[0])
        what = rng.random()
        if what < 0.15:
            r = bytearray(bytes(r).lower())
        elif what < 0.3:
            r[int(rng.integers(0, len(r)))] = ord("N")
        elif what < 0.4:
            unit = bytes(r[:int(rng.integers(40, 900))])
            r = bytearray((unit * (len(r) // len(unit) + 1))[:len(r)])
        reads.append(bytes(r))
    reads += synth.sample_reads(genomes, 10, 150, seed=int(rng.integers(1 << 30)))
    flags = dict(min_qcov=float(rng.choice([0.55, 0.35])), min_matched=int(rng.choice([1, 10])), dedup_threshold=int(rng.choice([256, 256, 0, 1 << 30])),
                 sort_by=int(rng.integers(0, 3)))
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir, device=0) as db:
            res = db.search(reads, params=default_params(**flags))
            _check_pairs(db, res, reads, None, default_params(**flags))
        synth.assert_parity(odb, res, reads, None, O.default_params(**flags))
    finally:
        odb.close()
