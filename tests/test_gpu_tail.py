"""Tail mode of the long-query COBS kernel (k2_cobs.hip, round 6): once all but <= 4 of the 8 sectors of a 1-KiB row tile are
dead (two by default), the idle lane octets take shares of the remaining rows and their partial counts are added to the owner's at the end.

The hits must not change: the same batch is searched with the mode on (default) and off (KMCPG_TAIL_SECTORS=0) and the two hit
lists compared entry by entry; a few queries are also counted from the rows resident in HBM with the oracle's arithmetic
(`h % NumSigs`, AND of the hash functions' rows, per-column sums, the integer threshold of util-db-search.go:7468-7470).  The
batches are built so that a (query, tile) keeps 0, 1, 2, 3, 4 or more sectors alive — full copies, 70 % copies (pass, with partial
counts) and 30 % copies (fail, but die late) of a query planted into neighbouring sectors of one block, into other blocks, or
nowhere — and `kmcpg_last_tail_waves` shows that the mode ran."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(db, torch, reads_t, offs_t, n, total, maxlen, params, cap):
    dev = reads_t.device
    hits = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    qk = torch.zeros(n, dtype=torch.int32, device=dev)
    ql = torch.zeros(n, dtype=torch.int32, device=dev)
    db.query_device(reads_t.data_ptr(), offs_t.data_ptr(), n, total, maxlen, hits.data_ptr(), cap, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(), params=params)
    torch.cuda.synchronize()
    m = int(cnt[0].item())
    assert m <= cap
    h = hits[:m].cpu().numpy().astype(np.int64)
    return h[np.lexsort((h[:, 1], h[:, 0]))], qk.cpu().numpy(), ql.cpu().numpy()


def _queries(torch, dev, lens, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    offs = torch.zeros(len(lens) + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(torch.as_tensor(lens, dtype=torch.int64, device=dev), 0)
    total = int(offs[-1].item())
    return acgt[torch.randint(0, 4, (total,), generator=g, device=dev)].contiguous(), offs.contiguous(), total


def _plant_prefixes(db, torch, reads, offs, lens, frac, cols):
    """plants the first `frac` of every query i into column cols[i] (-1: nowhere)"""
    dev = reads.device
    n = len(lens)
    plens = [max(64, int(l * frac)) for l in lens]
    poffs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    poffs[1:] = torch.cumsum(torch.as_tensor(plens, dtype=torch.int64, device=dev), 0)
    parts = [reads[int(offs[i]):int(offs[i]) + plens[i]] for i in range(n)]
    frag = torch.cat(parts).contiguous()
    tgt = torch.as_tensor(cols, dtype=torch.int32, device=dev).contiguous()
    db.plant_reads_device(frag.data_ptr(), poffs.contiguous().data_ptr(), n, int(poffs[-1].item()), max(plens), tgt.data_ptr())
    torch.cuda.synchronize()


def _oracle_counts(db, km, n_blocks, cols_per_block, num_hashes):
    out = []
    for b in range(n_blocks):
        ns = np.uint64(db.block_info(b)["num_sigs"])
        acc = None
        for t in range(num_hashes):
            hv = km if num_hashes == 1 else ((km >> np.uint64(32)).astype(np.uint32) + km.astype(np.uint32) * np.uint32(t)).astype(np.uint64)
            bits = db.read_rows(b, hv % ns)
            acc = bits if acc is None else (acc & bits)
        out.append(np.unpackbits(acc, axis=1)[:, :cols_per_block].sum(axis=0, dtype=np.int64))
    return out


def _plan(n, n_blocks, cols, rng):
    """per query: (fraction, column) plants.  Case c = i % 8 decides how many sectors of which tiles stay alive."""
    row_sectors = (cols + 1023) // 1024
    plants = []
    for i in range(n):
        b = int(rng.integers(0, n_blocks))
        tile = int(rng.integers(0, (row_sectors + 7) // 8))          # a 1-KiB tile of that block's rows ...
        secs = [s for s in rng.permutation(8) if (tile * 8 + s) * 1024 < cols]  # ... and its sectors in a random order

        def col(j):
            lo = (tile * 8 + int(secs[j % len(secs)])) * 1024
            return b * cols + lo + int(rng.integers(0, min(1024, cols - lo)))

        c = i % 8
        if c == 0:
            p = []                                                    # unrelated: every sector dies, no tail
        elif c == 1:
            p = [(1.0, col(0))]                                       # one live sector
        elif c == 2:
            p = [(1.0, col(0)), (0.7, col(1))]                        # two
        elif c == 3:
            p = [(1.0, col(0)), (0.7, col(1)), (0.3, col(2))]         # three, one of them failing late
        elif c == 4:
            p = [(0.7, col(0)), (0.7, col(1)), (0.7, col(2)), (1.0, col(3))]  # four
        elif c == 5:
            p = [(1.0, col(j)) for j in range(5)]                     # five: stays in the main loop
        elif c == 6:
            p = [(1.0, col(0)), (0.7, ((b + 1) % n_blocks) * cols + int(rng.integers(0, cols)))]  # two blocks, one sector each
        else:
            p = [(0.3, col(0))]                                       # a lone near miss: its sector dies late, inside the tail
        plants.append(p)
    return plants


def _check(db, torch, reads, offs, lens, n, total, params, n_blocks, cols, nh, cfg, O, oracle_ids, cap=1 << 18):
    db.set_profiling(2)
    with _Env(KMCPG_SPLIT_MIN=0, KMCPG_TAIL_SECTORS=None):
        h_on, qk, ql = _run(db, torch, reads, offs, n, total, max(lens), params, cap)
        tail_waves = db.last_tail_waves()
        bytes_on = db.last_gathered_bytes()
    with _Env(KMCPG_SPLIT_MIN=0, KMCPG_TAIL_SECTORS=0):
        h_off, qk2, ql2 = _run(db, torch, reads, offs, n, total, max(lens), params, cap)
        assert db.last_tail_waves() == 0
        bytes_off = db.last_gathered_bytes()
    with _Env(KMCPG_SPLIT_MIN=0, KMCPG_TAIL_SECTORS=1, KMCPG_TAIL_MIN=1):
        h_one, _, _ = _run(db, torch, reads, offs, n, total, max(lens), params, cap)
    with _Env(KMCPG_SPLIT_MIN=0, KMCPG_TAIL_SECTORS=4, KMCPG_TAIL_MIN=1):  # 3 and 4 live sectors: one helper octet each
        h_four, _, _ = _run(db, torch, reads, offs, n, total, max(lens), params, cap)
        assert db.last_tail_waves() >= tail_waves
    with _Env(KMCPG_SPLIT_MIN=0, KMCPG_PAIR=0):  # the two lane forms of a database as two launches (default: one grid, k2_cobs_pair)
        h_seq, _, _ = _run(db, torch, reads, offs, n, total, max(lens), params, cap)
    db.set_profiling(0)
    assert np.array_equal(qk, qk2) and np.array_equal(ql, ql2)
    assert h_on.shape == h_off.shape and np.array_equal(h_on, h_off)
    assert np.array_equal(h_one, h_off) and np.array_equal(h_four, h_off) and np.array_equal(h_seq, h_off)
    assert tail_waves > 0
    # no pruning inside the tail: a little more traffic than the plain loop is the price, not a multiple
    assert bytes_on <= 1.35 * bytes_off, (bytes_on, bytes_off)
    seq_h, offs_h = reads.cpu().numpy(), offs.cpu().numpy()
    got = {}
    for r, c, k in h_on:
        got.setdefault(int(r), []).append((int(c), int(k)))
    for i in oracle_ids:
        km = O.sort_unique(O.generate_kmers(seq_h[offs_h[i]:offs_h[i + 1]].tobytes(), cfg))
        assert len(km) == qk[i]
        cmin = max(params.min_matched, int(np.floor(len(km) * params.min_qcov)) + 1)
        want = []
        for b, cnt in enumerate(_oracle_counts(db, km, n_blocks, cols, nh)):
            want += [(b * cols + int(c), int(cnt[c])) for c in np.nonzero(cnt >= cmin)[0]]
        assert got.get(i, []) == want, i
    return tail_waves, bytes_on, bytes_off, got, qk


def test_tail_mode_three_hashes_genome_sketches(oracle_lib):
    """the genome-search shape: 3 hash functions, 782-byte rows (one 64-lane tile of 7 sectors), ~8 000 sketch k-mers, -t 0.4"""
    import torch
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    dev = torch.device("cuda:0")
    n_blocks, cols, nh, scale = 4, 6256, 3, 1000
    spec = lib.SynthSpec(k=21, num_hashes=nh, fpr=0.001, n_blocks=n_blocks, cols_per_block=cols, num_sigs=431000, kmers_per_col=10000, seed=5, scale=scale, sigs_step=13)
    n = 24
    lens = [4_000_000] * n
    reads, offs, total = _queries(torch, dev, lens, seed=31)
    rng = np.random.default_rng(7)
    plants = _plan(n, n_blocks, cols, rng)
    params = default_params(min_qcov=0.4, sort_by=2)
    with Database.open_synthetic(spec) as db:
        for rnd in range(5):
            for frac in (1.0, 0.7, 0.3):
                sel = [p[rnd][1] if (len(p) > rnd and p[rnd][0] == frac) else -1 for p in plants]
                if any(c >= 0 for c in sel):
                    _plant_prefixes(db, torch, reads, offs, lens, frac, sel)
        tw, b_on, b_off, got, qk_all = _check(db, torch, reads, offs, lens, n, total, params, n_blocks, cols, nh, O.sketch_cfg(k=21, scale=scale), O, (1, 2, 3, 4, 6, 7))
        # full copies come back with every sketch k-mer
        for i, p in enumerate(plants):
            for frac, c in p:
                if frac == 1.0:
                    assert (c, int(qk_all[i])) in got.get(i, []), (i, c)


def test_tail_mode_single_hash_hifi_syncmers(oracle_lib):
    """the HiFi shape with one NumSigs: 1 248-byte grouped rows = a 1-KiB tile (tail mode) + a 224-byte remainder (16-lane form)"""
    import torch
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    dev = torch.device("cuda:0")
    n_blocks, cols, s = 32, 312, 11
    spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=n_blocks, cols_per_block=cols, num_sigs=300000, kmers_per_col=100000, seed=2, syncmer_s=s, sigs_step=0)
    n = 1024
    rng = np.random.default_rng(3)
    lens = np.clip((rng.standard_normal(n) * 2000 + 10000).astype(np.int64), 2000, 20000).tolist()
    reads, offs, total = _queries(torch, dev, lens, seed=4)
    ncols = n_blocks * cols
    params = default_params()
    with Database.open_synthetic(spec) as db:
        full = [int(rng.integers(0, ncols)) if i % 4 != 0 else -1 for i in range(n)]
        part = [int(rng.integers(0, ncols)) if i % 3 == 1 else -1 for i in range(n)]
        near = [int(rng.integers(0, ncols)) if i % 5 == 2 else -1 for i in range(n)]
        _plant_prefixes(db, torch, reads, offs, lens, 1.0, full)
        _plant_prefixes(db, torch, reads, offs, lens, 0.8, part)
        _plant_prefixes(db, torch, reads, offs, lens, 0.45, near)
        tw, b_on, b_off, got, qk_all = _check(db, torch, reads, offs, lens, n, total, params, n_blocks, cols, 1, O.sketch_cfg(k=21, syncmer_s=s), O, (0, 1, 2, 5, 7, n - 1))
        assert tw >= n // 4
        for i in range(n):
            if full[i] >= 0:
                assert any(c == full[i] for c, _ in got.get(i, [])), i


def test_tail_mode_24_planes_plain_kmers():
    """queries of more than 65 534 k-mers count on 24 planes: one block of 16 384 columns (two 1-KiB tiles), all k-mers of 80-kb queries"""
    import torch
    from kmcp_amd import Database, default_params, lib
    dev = torch.device("cuda:0")
    spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=2, cols_per_block=16384, num_sigs=400000, kmers_per_col=100000, seed=9, sigs_step=3)
    n = 12
    lens = [80_000 + 1000 * i for i in range(n)]
    reads, offs, total = _queries(torch, dev, lens, seed=8)
    rng = np.random.default_rng(11)
    params = default_params(dedup_threshold=1 << 30)  # keep every k-mer (no sort + unique): ~80 000 per query
    with Database.open_synthetic(spec) as db:
        full = [int(rng.integers(0, 2 * 16384)) if i % 3 != 2 else -1 for i in range(n)]
        part = [int(rng.integers(0, 2 * 16384)) if i % 2 == 0 else -1 for i in range(n)]
        _plant_prefixes(db, torch, reads, offs, lens, 1.0, full)
        _plant_prefixes(db, torch, reads, offs, lens, 0.75, part)
        db.set_profiling(2)
        with _Env(KMCPG_SPLIT_MIN=0, KMCPG_TAIL_SECTORS=None):
            h_on, qk, _ = _run(db, torch, reads, offs, n, total, max(lens), params, 1 << 16)
            tw = db.last_tail_waves()
        with _Env(KMCPG_SPLIT_MIN=0, KMCPG_TAIL_SECTORS=0):
            h_off, qk2, _ = _run(db, torch, reads, offs, n, total, max(lens), params, 1 << 16)
        with _Env(KMCPG_SPLIT_MIN=0, KMCPG_TAIL_SECTORS=4, KMCPG_TAIL_MIN=1):
            h_four, _, _ = _run(db, torch, reads, offs, n, total, max(lens), params, 1 << 16)
        db.set_profiling(0)
        assert (qk > 65534).all() and np.array_equal(qk, qk2)
        assert np.array_equal(h_on, h_off) and np.array_equal(h_four, h_off) and tw > 0
        got = {(int(r), int(c)): int(k) for r, c, k in h_on}
        for i in range(n):
            if full[i] >= 0:
                assert got.get((i, full[i])) == int(qk[i]), i
            if part[i] >= 0 and part[i] != full[i]:
                assert (i, part[i]) in got and got[(i, part[i])] < int(qk[i]), i
