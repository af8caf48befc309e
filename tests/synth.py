"""Synthetic genomes, databases and reads for the parity tests (scaled down).

Databases are built with the oracle's restatement of `kmcp compute` + `kmcp index`
(kmcp/cmd/compute.go:746-824, index.go:657-682,1023,1107-1309): test infrastructure, not product.
"""
import numpy as np

from oracle import oracle as O

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_genomes(n, length, seed):
    rng = np.random.default_rng(seed)
    return [_ACGT[rng.integers(0, 4, size=length)].tobytes() for _ in range(n)]


def split_chunks(seq: bytes, n_chunks: int, overlap: int):
    """compute.go:686-690: splitSize = (len + (n-1)*overlap + n-1)/n, step = splitSize - overlap (linear)."""
    if n_chunks <= 1:
        return [seq]
    L = len(seq)
    size = (L + (n_chunks - 1) * overlap + n_chunks - 1) // n_chunks
    step = size - overlap
    out = []
    i = 0
    while i < L and len(out) < n_chunks:
        out.append(seq[i:i + size])
        i += step
    return out


def make_columns(genomes, cfg, n_chunks=1, overlap=150, names=None):
    cols = []
    for gi, g in enumerate(genomes):
        name = names[gi] if names else f"g{gi:05d}"
        chunks = split_chunks(g, n_chunks, overlap)
        for ci, c in enumerate(chunks):
            h = O.sort_unique(O.generate_kmers(c, cfg))
            cols.append((name, len(g), ci, len(chunks), h))
    return cols


def make_db(tmp, genomes, k=21, n_chunks=1, overlap=150, num_hashes=1, fpr=0.3, threads=32, block_size=0, scale=1,
            syncmer_s=0, minimizer_w=0, names=None):
    cfg = O.sketch_cfg(k=k, scale=scale, syncmer_s=syncmer_s, minimizer_w=minimizer_w)
    cols = make_columns(genomes, cfg, n_chunks, overlap, names)
    return O.build_db(str(tmp), cfg, cols, num_hashes=num_hashes, fpr=fpr, threads=threads, block_size=block_size)


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def sample_reads(genomes, n, length, sub_rate=0.01, seed=7, frac_random=0.1, both_strands=True, n_rate=0.0):
    """Reads sampled from the genomes with substitutions, plus a fraction of uniform random reads."""
    rng = np.random.default_rng(seed)
    reads = []
    for _ in range(n):
        if rng.random() < frac_random:
            r = _ACGT[rng.integers(0, 4, size=length)].copy()
        else:
            g = genomes[int(rng.integers(0, len(genomes)))]
            p = int(rng.integers(0, max(1, len(g) - length)))
            r = np.frombuffer(g[p:p + length], dtype=np.uint8).copy()
            m = rng.random(len(r)) < sub_rate
            r[m] = _ACGT[rng.integers(0, 4, size=int(m.sum()))]
        if n_rate > 0:
            m = rng.random(len(r)) < n_rate
            r[m] = ord("N")
        b = r.tobytes()
        if both_strands and rng.random() < 0.5:
            b = b.translate(_COMP)[::-1]
        reads.append(b)
    return reads


def oracle_tuples(odb, read, read2=None, params=None):
    """Per-read parity record from the oracle: (qlen, qkmers, sorted [(col, mkmers, qcov, tcov, jacc)], {col: fpr})."""
    o = odb.search(read, read2, params=params)
    ms = o["matches"] or []
    return (o["qlen"], o["qkmers"], sorted((m["col_global"], m["mkmers"], m["qcov"], m["tcov"], m["jacc"]) for m in ms),
            {m["col_global"]: m["fpr"] for m in ms}, [m["col_global"] for m in ms])


def gpu_tuples(res, i):
    ms = res.read(i)
    return (int(res.qlen[i]), int(res.qkmers[i]),
            sorted((int(m["col"]), int(m["mkmers"]), float(m["qcov"]), float(m["tcov"]), float(m["jacc"])) for m in ms),
            {int(m["col"]): float(m["fpr"]) for m in ms}, [int(m["col"]) for m in ms])


def assert_parity(odb, res, reads, reads2=None, oparams=None, check_order=True):
    """Bit-exact per-read sets of (target column, mKmers, qCov, tCov, jacc), and the FPR doubles bit for bit (product and oracle
    restate the same Go arithmetic, util-fpr.go:32-71; both reproduce the reference's printed values, tests/test_fpr_golden.py)."""
    n_hits = 0
    for i, r in enumerate(reads):
        want = oracle_tuples(odb, r, reads2[i] if reads2 is not None else None, oparams)
        got = gpu_tuples(res, i)
        assert got[0] == want[0], f"read {i}: qLen {got[0]} != {want[0]}"
        assert got[1] == want[1], f"read {i}: qKmers {got[1]} != {want[1]}"
        assert got[2] == want[2], f"read {i}: matches differ\n gpu={got[2]}\n ora={want[2]}"
        for c, f in want[3].items():
            g = got[3][c]
            assert g == f, f"read {i} col {c}: FPR {g!r} vs {f!r}"
        if check_order:
            # both sides break exact score ties by column, so even the order agrees
            assert got[4] == want[4], f"read {i}: order differs {got[4]} vs {want[4]}"
        n_hits += len(want[2])
    return n_hits
