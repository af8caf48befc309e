"""Parity at BASELINE.json's full size (GTDB-scale index: 32 blocks x 14 976 columns x 968 700 rows = 58 GB in HBM)
through size-independent properties, plus an oracle check on rows read back from HBM:

  * planting: a fragment whose k-mers were planted in column c comes back with (c, count == n): no false negatives;
  * shard invariance: the union of the hit lists of the two halves of the index == the hit list of the whole index
    (this is also the multi-GPU merge, exercised here with two handles on one GPU);
  * oracle on a sample: for a few reads the 130 x 32 rows they touch are copied back and counted by the oracle.

Two layouts of the same 58 GB: every block with the same NumSigs (the 32 blocks are laid side by side and served as ONE 59 904-byte
row: `setup`), and every block with a NumSigs of its own (`sigs_step=64`: 32 groups of one block, 1872-byte rows, what a real
database has and what bench.py's `gtdb` workload measures: `setup_distinct`) - a different slot / tile decomposition of the same
kernel, each with its own oracle check.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_BLOCKS, COLS, NUM_SIGS, KMERS = 32, 14976, 968700, 345510
B, L = 8192, 150


SIGS_STEP = 64  # bench.py WORKLOADS["gtdb"]["sigs_step"]


def _build(sigs_step, with_halves, need_bytes):
    import torch
    from kmcp_amd import Database, default_params, lib
    free_b, _ = torch.cuda.mem_get_info(0)
    if free_b < need_bytes:
        pytest.skip(f"needs {need_bytes / 1e9:.0f} GB of free HBM")
    dev = torch.device("cuda:0")
    spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=N_BLOCKS, cols_per_block=COLS, num_sigs=NUM_SIGS, kmers_per_col=KMERS, seed=7, sigs_step=sigs_step)
    whole = Database.open_synthetic(spec, device=0)
    halves = [Database.open_synthetic(spec, device=0, shard_rank=r, shard_count=2) for r in range(2)] if with_halves else []
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    reads = acgt[torch.randint(0, 4, (B, L), generator=g, device=dev)].contiguous().view(-1)
    cols = torch.randint(0, N_BLOCKS * COLS, (B,), generator=g, device=dev).to(torch.int32)
    cols[B // 2:] = -1  # second half: random reads, nothing planted
    offs = (torch.arange(B + 1, device=dev, dtype=torch.int64) * L).contiguous()
    for db in [whole] + halves:  # the same generator + the same plants => identical bit matrices
        db.plant_reads_device(reads.data_ptr(), offs.data_ptr(), B, B * L, L, cols.data_ptr())
    return dict(torch=torch, dev=dev, whole=whole, halves=halves, reads=reads, offs=offs, cols=cols.cpu().numpy(), params=default_params(), lib=lib,
                sigs_step=sigs_step)


@pytest.fixture(scope="module")
def setup():
    s = _build(0, True, 130e9)
    yield s
    for db in [s["whole"]] + s["halves"]:
        db.close()


@pytest.fixture(scope="module")
def setup_distinct():
    """the layout bench.py measures: block b has NUM_SIGS + 64 b rows, so no two blocks can share a gather"""
    s = _build(SIGS_STEP, False, 70e9)
    yield s
    s["whole"].close()


def _query(s, db):
    torch, dev = s["torch"], s["dev"]
    cap = 8 * B
    hits = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    qk = torch.zeros(B, dtype=torch.int32, device=dev)
    ql = torch.zeros(B, dtype=torch.int32, device=dev)
    db.query_device(s["reads"].data_ptr(), s["offs"].data_ptr(), B, B * L, L, hits.data_ptr(), cap, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(),
                    params=s["params"])
    torch.cuda.synchronize()
    n = int(cnt[0].item())
    assert n <= cap
    h = hits[:n].cpu().numpy().astype(np.int64)
    return h[np.lexsort((h[:, 1], h[:, 0]))], qk.cpu().numpy(), ql.cpu().numpy()


def test_info_is_gtdb_scale(setup):
    i = setup["whole"].info
    assert i.n_blocks == 32 and i.n_cols == 479232 and i.matrix_bytes == 32 * 968700 * 1872 == 58029004800
    assert i.row_bytes_sum_local == 32 * 1872
    assert setup["halves"][0].info.n_blocks_local == 16 and setup["halves"][1].info.n_blocks_local == 16


def _check_planted(setup):
    h, qk, ql = _query(setup, setup["whole"])
    assert (qk == 130).all() and (ql == 150).all()
    got = {(int(r), int(c)): int(n) for r, c, n in h}
    cols = setup["cols"]
    for r in range(B // 2):
        assert got.get((r, int(cols[r]))) == 130, r  # every planted k-mer is found: count == n
    # chance hits: P(Bin(130, 0.3) >= 72) ~ 1e-9 per (read, column) x 3.9e9 pairs => a handful, all barely above 72
    chance = [n for (r, c), n in got.items() if r >= B // 2 or c != int(cols[r])]
    assert len(chance) < 60 and all(72 <= n < 85 for n in chance)


def test_planted_fragments_have_no_false_negatives(setup):
    _check_planted(setup)


def test_planted_fragments_have_no_false_negatives_distinct_numsigs(setup_distinct):
    i = setup_distinct["whole"].info
    assert i.n_blocks == 32 and i.matrix_bytes == sum((NUM_SIGS + SIGS_STEP * b) * 1872 for b in range(N_BLOCKS))
    assert [setup_distinct["whole"].block_info(b)["num_sigs"] for b in (0, 1, 31)] == [NUM_SIGS, NUM_SIGS + SIGS_STEP, NUM_SIGS + 31 * SIGS_STEP]
    _check_planted(setup_distinct)


def test_shard_union_equals_whole(setup):
    hw, qk, _ = _query(setup, setup["whole"])
    parts = [_query(setup, db) for db in setup["halves"]]
    merged = np.concatenate([p[0] for p in parts])
    merged = merged[np.lexsort((merged[:, 1], merged[:, 0]))]
    assert np.array_equal(hw, merged)
    assert len(parts[0][0]) > 0 and len(parts[1][0]) > 0
    for p in parts:
        assert np.array_equal(p[1], qk)
    # finalize over the concatenated shard lists == finalize over the whole index's list
    lib = setup["lib"]
    ql = np.full(B, 150, dtype=np.int32)
    a = setup["whole"].finalize(np.ascontiguousarray(hw.astype(np.uint32)).view(lib.HIT_DTYPE).reshape(-1), qk, ql)
    b = setup["halves"][1].finalize(np.ascontiguousarray(np.concatenate([p[0] for p in parts])[::-1].astype(np.uint32)).view(lib.HIT_DTYPE).reshape(-1), qk, ql)
    assert np.array_equal(a.offs, b.offs) and np.array_equal(a.matches, b.matches)


def test_oracle_on_rows_read_back(setup, oracle_lib):
    """Counts of 6 reads recomputed by the oracle from the very rows resident in HBM (130 x 32 rows each)."""
    _check_rows_read_back(setup, oracle_lib)


def test_oracle_on_rows_read_back_distinct_numsigs(setup_distinct, oracle_lib):
    """... and on the benchmarked layout: 32 different NumSigs, each block's rows addressed with that block's own modulus."""
    _check_rows_read_back(setup_distinct, oracle_lib)


def _check_rows_read_back(setup, oracle_lib):
    O = oracle_lib
    db = setup["whole"]
    h, qk, _ = _query(setup, db)
    reads = setup["reads"].cpu().numpy()
    cfg = O.sketch_cfg(k=21)
    for r in (0, 1, 2, B // 2 - 1, B // 2, B - 1):
        km = O.generate_kmers(reads[r * L:(r + 1) * L].tobytes(), cfg)
        assert len(km) == qk[r] == 130
        want = []
        for b in range(N_BLOCKS):
            rows = db.read_rows(b, km % np.uint64(NUM_SIGS + setup["sigs_step"] * b))  # [130, 1872] bytes; loc = h % NumSigs of THIS block (:6811-6816)
            bits = np.unpackbits(rows, axis=1)[:, :COLS]               # bit 7 of byte c/8 = column c
            cnt = bits.sum(axis=0)
            for c in np.nonzero(cnt >= 72)[0]:                         # float64(c) > 130*0.55 = 71.5
                want.append((r, b * COLS + int(c), int(cnt[c])))
        got = [tuple(int(x) for x in row) for row in h[h[:, 0] == r]]
        assert got == sorted(want), r


def test_pruning_keeps_boundary_columns(oracle_lib):
    """Sector pruning is exact at its boundary: a column whose matches all sit in the LAST cmin k-mers of the read has
    count + remaining == cmin at every step until the end and must be reported; one k-mer fewer must not.  All three kernel
    classes (3-, 125- and 1872-byte rows), low Bloom density so that every other sector dies early."""
    import torch
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    reads = [bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)) for _ in range(64)]
    cfg = O.sketch_cfg(k=21)
    kms = [O.generate_kmers(r, cfg) for r in reads]
    for cols_per_block in (24, 1000, 14976):
        spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=3, cols_per_block=cols_per_block, num_sigs=200003, kmers_per_col=4000, seed=3)
        with Database.open_synthetic(spec) as db:  # density 1-exp(-4000/200003) = 2 %
            want = set()
            ncols = 3 * cols_per_block
            for i, km in enumerate(kms):
                n = len(km)
                cmin = max(10, int(np.floor(n * 0.55)) + 1)
                c_hit, c_miss, c_first = (7 * i + 1) % ncols, (7 * i + 3) % ncols, (7 * i + 5) % ncols
                db.plant(c_hit, km[n - cmin:])        # exactly cmin matches, all at the end
                db.plant(c_miss, km[n - cmin + 1:])   # cmin - 1
                db.plant(c_first, km[:cmin])          # exactly cmin matches, all at the start
            # expected counts from the rows actually resident (plants of different reads may touch the same column)
            seqs, offs = lib.pack_reads(reads)
            t_seqs = torch.from_numpy(seqs).to(dev)
            t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
            cap = 4096
            hits = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            qk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
            ql = torch.zeros(len(reads), dtype=torch.int32, device=dev)
            for i, km in enumerate(kms):
                n = len(km)
                cmin = max(10, int(np.floor(n * 0.55)) + 1)
                for b in range(3):
                    rows = db.read_rows(b, km % np.uint64(200003))
                    c = np.unpackbits(rows, axis=1)[:, :cols_per_block].sum(axis=0)
                    for col in np.nonzero(c >= cmin)[0]:
                        want.add((i, b * cols_per_block + int(col), int(c[col])))
            got, gathered, hashed = {}, {}, {}
            db.set_profiling(2)  # the kernel counts the 16-byte row loads it issues
            for prune in ("1", "0"):
                os.environ["KMCPG_PRUNE"] = prune
                try:
                    db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), 150, hits.data_ptr(), cap, cnt.data_ptr(),
                                    qk.data_ptr(), ql.data_ptr(), params=default_params())
                    torch.cuda.synchronize()
                finally:
                    os.environ.pop("KMCPG_PRUNE", None)
                h = hits[:int(cnt[0].item())].cpu().numpy()
                got[prune] = {(int(r), int(c), int(k)) for r, c, k in h}
                gathered[prune] = db.last_gathered_bytes()
                hashed[prune] = db.last_hash_bytes()
            assert got["1"] == got["0"] == want
            # without pruning every (k-mer, group) costs one padded row (k-mers rounded up to whole groups of 8 rows: the tail
            # reads the all-zero row); the three equal-NumSigs blocks form one group.  With pruning the kernel asks for less.
            stride = db.block_info(0)["stride"]
            assert gathered["0"] == sum((len(km) + 7) // 8 * 8 for km in kms) * stride
            assert gathered["1"] <= gathered["0"]
            # ... and a read's hashes are fetched once per slot (every whole KiB of the row is a tile, the rest one more)
            nslots = stride // 1024 + (1 if stride % 1024 else 0)
            assert hashed["0"] == 8 * sum(len(km) for km in kms) * nslots
            assert 0 < hashed["1"] <= hashed["0"]
            if cols_per_block == 14976:
                assert gathered["1"] < 0.8 * gathered["0"]  # 2 % density: almost every sector dies early (narrow rows share one sector with the planted columns)
            assert len(want) >= 2 * len(reads)  # the end-loaded and the start-loaded column of every read
