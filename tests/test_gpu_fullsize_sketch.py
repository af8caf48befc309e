"""BASELINE.json configs[2] and configs[4] at their full shapes, through size-independent properties plus an oracle check on
rows read back from HBM (the scaled-down oracle diffs of these modes live in test_gpu_parity.py / test_gpu_fuzz.py):

  * configs[2] — genome search: 4-Mbp assemblies against a FracMinHash (scale 1000) index of 50 048 references, 3 hash
    functions, fpr 0.001 (8 blocks x 6 256 columns); queries of ~8 000 sketch k-mers take the sort+unique path and the
    SPLIT form of the COBS kernel (device-wide count arrays + threshold pass);
  * configs[4] — HiFi reads (~10 kb, N(10 k, 2 k)) against a Closed-Syncmer (k=21, s=11) index of 10 k chunks: window
    sketching on the workgroup form of K1, adjacent-repeat removal, LDS sort+unique, 16-plane counters.

Properties: a query whose sketch was planted into a column comes back with that column and count == qKmers (no false
negatives, every planted k-mer found, exact unique counts); unplanted queries match nothing (at fpr 0.001^... / 0.3 a chance
hit is out of reach at these thresholds); qKmers equals the oracle's sorted-unique sketch size; for a few queries the rows
they touch are copied back and counted with the oracle's arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _random_queries(torch, dev, lens, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    offs = torch.zeros(len(lens) + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(torch.as_tensor(lens, dtype=torch.int64, device=dev), 0)
    total = int(offs[-1].item())
    return acgt[torch.randint(0, 4, (total,), generator=g, device=dev)].contiguous(), offs.contiguous(), total


def _oracle_counts(O, db, km, n_blocks, cols_per_block, num_hashes):
    """per-block column counts of the (sorted-unique) k-mers km from the rows resident in HBM, by the oracle's arithmetic"""
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


def test_config2_genome_search_fracminhash_three_hashes(oracle_lib):
    import torch
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    dev = torch.device("cuda:0")
    n_blocks, cols, nh, scale = 8, 6256, 3, 1000
    spec = lib.SynthSpec(k=21, num_hashes=nh, fpr=0.001, n_blocks=n_blocks, cols_per_block=cols, num_sigs=431000, kmers_per_col=10000, seed=3, scale=scale,
                         sigs_step=13)
    n = 48
    lens = [4_000_000] * n
    reads, offs, total = _random_queries(torch, dev, lens, seed=21)
    target = torch.full((n,), -1, dtype=torch.int32, device=dev)
    planted = {i: (i * 1049 + 17) % (n_blocks * cols) for i in range(0, n, 2)}  # every other genome is "in the database"
    for i, c in planted.items():
        target[i] = c
    params = default_params(min_qcov=0.4, sort_by=2)  # -g --sort-by jacc -t 0.4 as SURVEY.md §8d config 2
    with Database.open_synthetic(spec) as db:
        assert db.info.scaled and db.info.scale == scale and db.info.num_hashes == nh and int(db.info.n_cols) == 50048
        db.plant_reads_device(reads.data_ptr(), offs.data_ptr(), n, total, 4_000_000, target.data_ptr())
        h, qk, ql = _run(db, torch, reads, offs, n, total, 4_000_000, params, cap=1 << 16)
        assert (ql == 4_000_000).all()
        assert ((qk > 7000) & (qk < 9000)).all()  # ~ 2 x 4e6 / 1000 canonical sketch hashes
        got = {}
        for r, c, k in h:
            got.setdefault(int(r), []).append((int(c), int(k)))
        for i in range(n):
            if i in planted:
                assert got.get(i) == [(planted[i], int(qk[i]))], i  # every sketch k-mer found, nothing else at fpr 0.001
            else:
                assert i not in got
        # qKmers = the oracle's sorted-unique FracMinHash sketch; counts from the rows actually resident
        seq_h = reads.cpu().numpy()
        offs_h = offs.cpu().numpy()
        cfg = O.sketch_cfg(k=21, scale=scale)
        for i in (0, 1, n - 1):
            km = O.sort_unique(O.generate_kmers(seq_h[offs_h[i]:offs_h[i + 1]].tobytes(), cfg))
            assert len(km) == qk[i]
            cmin = max(params.min_matched, int(np.floor(len(km) * params.min_qcov)) + 1)
            want = []
            for b, cnt in enumerate(_oracle_counts(O, db, km, n_blocks, cols, nh)):
                want += [(b * cols + int(c), int(cnt[c])) for c in np.nonzero(cnt >= cmin)[0]]
            assert got.get(i, []) == want, i


def test_config4_hifi_reads_closed_syncmer(oracle_lib):
    import torch
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    dev = torch.device("cuda:0")
    n_blocks, cols, s = 32, 312, 11
    spec = lib.SynthSpec(k=21, num_hashes=1, fpr=0.3, n_blocks=n_blocks, cols_per_block=cols, num_sigs=300000, kmers_per_col=100000, seed=2, syncmer_s=s,
                         sigs_step=5)
    n = 4096
    rng = np.random.default_rng(3)
    lens = np.clip((rng.standard_normal(n) * 2000 + 10000).astype(np.int64), 2000, 20000).tolist()
    reads, offs, total = _random_queries(torch, dev, lens, seed=4)
    target = torch.full((n,), -1, dtype=torch.int32, device=dev)
    planted = {i: (i * 613 + 5) % (n_blocks * cols) for i in range(0, n, 2)}
    for i, c in planted.items():
        target[i] = c
    params = default_params()
    with Database.open_synthetic(spec) as db:
        assert db.info.syncmer and db.info.syncmer_s == s and int(db.info.n_cols) == 9984
        db.plant_reads_device(reads.data_ptr(), offs.data_ptr(), n, total, max(lens), target.data_ptr())
        h, qk, ql = _run(db, torch, reads, offs, n, total, max(lens), params, cap=1 << 18)
        assert np.array_equal(ql, np.array(lens))
        got = {}
        for r, c, k in h:
            got.setdefault(int(r), []).append((int(c), int(k)))
        for i in range(n):
            if i in planted:
                assert (planted[i], int(qk[i])) in got.get(i, []), i  # all of the read's syncmers are in its column
            # (other columns may match by chance: several reads are planted into one column of a 100 k-k-mer filter)
        seq_h = reads.cpu().numpy()
        offs_h = offs.cpu().numpy()
        cfg = O.sketch_cfg(k=21, syncmer_s=s)
        for i in (0, 1, 2, n // 2 + 1, n - 1):
            km = O.sort_unique(O.generate_kmers(seq_h[offs_h[i]:offs_h[i + 1]].tobytes(), cfg))
            assert len(km) == qk[i] and 200 < len(km) < 4000
            cmin = max(params.min_matched, int(np.floor(len(km) * params.min_qcov)) + 1)
            want = []
            for b, cnt in enumerate(_oracle_counts(O, db, km, n_blocks, cols, 1)):
                want += [(b * cols + int(c), int(cnt[c])) for c in np.nonzero(cnt >= cmin)[0]]
            assert got.get(i, []) == want, i
