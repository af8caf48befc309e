"""Pins the CPU oracle to the reference's own golden values: the two demo-searching tables
(/root/reference/demo-searching/README.md:61-68 and :102-109), hash known-answer values and index arithmetic.
The reference has no unit tests; these documentation goldens are the only pins it offers."""
import gzip
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLD, "demo_searching_tables.json")))


@pytest.fixture(scope="module")
def sketches():
    return np.load(os.path.join(GOLD, "demo_searching_sketches.npz"))


def read_query():
    with gzip.open(os.path.join(GOLD, "NC_018658.1.fasta.gz"), "rt") as fh:
        return "".join(line.strip() for line in fh if line[0] != ">").encode()


def build_demo_db(O, tmp, meta, sketches, mode):
    t = meta["tables"][mode]
    cfg = O.sketch_cfg(k=t["k"], scale=t["scale"], syncmer_s=t["syncmer_s"])
    cols = [(acc, g["gsize"], 0, 1, sketches[f"{mode}:{acc}"]) for acc, g in sorted(meta["genomes"].items())]
    # `kmcp index -n 3 -f 0.01` with >= 2 threads: sBlock = 8 => blocks {8 smallest} + {largest} (index.go:671-682)
    return O.build_db(str(tmp), cfg, cols, num_hashes=3, fpr=0.01, threads=8)


def test_hash_kats(oracle_lib, meta):
    O = oracle_lib
    L = O.lib()
    for key in ("nthash_k21_ACGTx", "nthash_k21_A21", "nthash_k31_GATTACA"):
        seq, want = meta["kats"][key]
        assert L.ko_nthash_kmer(seq.encode(), len(seq), 1) == int(want, 16)
        # reverse complement has the same canonical hash
        rc = seq.encode().translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
        assert L.ko_nthash_kmer(rc, len(rc), 1) == int(want, 16)
    for scale, want in meta["kats"]["max_hash"].items():
        assert L.ko_max_hash(int(scale)) == want
    for n, h, f, want in meta["kats"]["calc_signature_size"]:
        assert L.ko_calc_signature_size(n, h, f) == want
    q = read_query()
    assert hex(int(np.bitwise_xor.reduce(O.nthash_all(q[:150], 21)))) == meta["kats"]["query_first150_k21_xor"]


def test_rolling_equals_closed_form(oracle_lib):
    O = oracle_lib
    rng = np.random.default_rng(0)
    seq = np.frombuffer(b"ACGTNacgtRY", dtype=np.uint8)[rng.integers(0, 11, 500)].tobytes()
    for k in (1, 5, 21, 31, 64, 65, 100):
        roll = O.nthash_all(seq, k)
        closed = np.array([O.lib().ko_nthash_kmer(seq[i:i + k], k, 1) for i in range(len(seq) - k + 1)], dtype=np.uint64)
        assert np.array_equal(roll, closed), k


@pytest.mark.parametrize("mode", ["minhash", "syncmer"])
def test_demo_searching_table(oracle_lib, meta, sketches, tmp_path, mode):
    """All 18 published values of the table, from the committed sketches + the query genome."""
    O = oracle_lib
    db = O.OracleDB(build_demo_db(O, tmp_path, meta, sketches, mode))
    assert db.nblocks == 2 and [db.block_info(b)[1] for b in range(2)] == [8, 1]
    r = db.search(read_query(), params=O.default_params(min_qcov=0.5, sort_by=2))
    db.close()
    got = [[m["target"], "%.4f" % m["qcov"], "%.4f" % m["tcov"], "%.4f" % m["jacc"]] for m in r["matches"]]
    assert got == meta["tables"][mode]["rows"]
    assert r["qkmers"] == meta["genomes"]["NC_018658.1"][f"{mode}_kmers"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/demo-searching/refs"), reason="reference tree not present")
def test_fixture_sketches_match_reference_genomes(oracle_lib, meta, sketches):
    """The committed sketches are what the oracle's `compute` restatement yields on the reference's genomes."""
    O = oracle_lib
    from tests.golden.make_golden import read_fasta_gz
    for acc in ("NC_010655.1", "NC_018658.1"):
        recs = read_fasta_gz(f"/root/reference/demo-searching/refs/{acc}.fasta.gz")
        for mode, t in meta["tables"].items():
            cfg = O.sketch_cfg(k=t["k"], scale=t["scale"], syncmer_s=t["syncmer_s"])
            h = O.sort_unique(np.concatenate([O.generate_kmers(s, cfg) for n, s in recs if "plasmid" not in n]))
            assert np.array_equal(h, sketches[f"{mode}:{acc}"])


def test_syncmer_discriminates(oracle_lib, meta):
    """Sensitivity noted in SURVEY.md §8c: the sketch sizes are what the goldens need (14 482 / 10 071 for the query)."""
    assert meta["genomes"]["NC_018658.1"]["syncmer_kmers"] == 14482
    assert meta["genomes"]["NC_018658.1"]["minhash_kmers"] == 10071
    assert meta["genomes"]["NZ_CP028116.1"]["minhash_kmers"] == 10439


def test_query_fpr_properties(oracle_lib):
    L = oracle_lib.lib()
    # Theorem 2: 1 - sum_{i<=k} C(n,i) p^i (1-p)^(n-i); monotone in k; matches scipy's survival function
    from scipy.stats import binom
    for n, p in ((130, 0.3), (70, 0.3), (249, 0.25), (20, 0.01)):
        prev = 1.0
        for k in range(0, n + 1, max(1, n // 17)):
            v = L.ko_query_fpr(n, k, p)
            assert 0.0 <= v <= prev + 1e-15
            prev = v
            assert abs(v - binom.sf(k, n, p)) < 1e-9
    assert L.ko_binomial_coeff(130, 65) == pytest.approx(9.5067625827960698e37, rel=1e-12)
    assert L.ko_go_pow(0.3, 72.0) == pytest.approx(0.3 ** 72, rel=1e-13)
    assert L.ko_go_pow(2.0, 10.0) == 1024.0


def test_uniki_roundtrip_and_layout(oracle_lib, tmp_path):
    """Header fields, row addressing and the bit order (bit 7 of byte c/8 = column c) of .uniki files."""
    O = oracle_lib
    from tests import synth
    genomes = synth.random_genomes(11, 3000, seed=3)
    cfg = O.sketch_cfg(k=21)
    cols = synth.make_columns(genomes, cfg, n_chunks=2, overlap=100)
    db_dir = O.build_db(str(tmp_path), cfg, cols, num_hashes=2, fpr=0.1, threads=2)
    raw = open(os.path.join(db_dir, "_block001.uniki"), "rb").read()
    assert raw[:8] == b".kmcpidx" and raw[8] == 4 and raw[9] == 21 and raw[10] & 1 and raw[11] == 2
    db = O.OracleDB(db_dir)
    assert db.ncols == 22 and db.num_hashes == 2 and abs(db.fpr - 0.1) < 1e-15
    # every k-mer of a column is found in that column: count == size
    tot = 0
    for b in range(db.nblocks):
        ns, nc, rb = db.block_info(b)
        assert rb == (nc + 7) // 8
        for c in range(nc):
            name, tidx, gsize, size = db.col_info(tot + c)
            col = next(x for x in cols if x[0] == name and x[2] == (tidx & 0xFFFF))
            cnt = db.block_counts(b, col[4])
            assert cnt[c] == size == len(col[4])
            assert tidx >> 16 == 2 and gsize == 3000
            # numpy restatement of the addressing: row = uint32(hi + lo*i) % NumSigs, bit 7-c%8 of byte c//8
            rows = db.block_rows(b)
            h = col[4][:50]
            for i in range(2):
                loc = ((h >> np.uint64(32)).astype(np.uint32) + h.astype(np.uint32) * np.uint32(i)).astype(np.uint64) % np.uint64(ns)
                assert ((rows[loc.astype(np.int64), c // 8] >> (7 - c % 8)) & 1).all()
        tot += nc
    db.close()


def test_handle_query_gates(oracle_lib, tmp_path):
    O = oracle_lib
    from tests import synth
    genomes = synth.random_genomes(5, 4000, seed=9)
    db = O.OracleDB(synth.make_db(tmp_path, genomes, k=21, threads=2))
    g = genomes[0]
    assert db.search(g[:29])["qkmers"] == 0 and db.search(g[:29])["matches"] is None      # < -m 30
    r = db.search(g[:30])                                                                 # 10 k-mers == -c
    assert r["qkmers"] == 10 and r["matches"] and r["matches"][0]["mkmers"] == 10
    assert db.search(g[:30], params=O.default_params(min_matched=11))["matches"] is None
    r = db.search(g[100:250])
    assert r["qlen"] == 150 and r["qkmers"] == 130 and r["matches"][0]["qcov"] == 1.0
    r = db.search(g[100:250], g[400:550])
    assert r["qlen"] == 300 and r["qkmers"] == 260  # > 256 => sort+unique, no duplicates here
    # repeated k-mers are counted twice below the dedup threshold (search.go:55-57)
    r = db.search(g[100:160] + g[100:160])
    assert r["qkmers"] == 100
    # line formatting as search.go:517-575
    import ctypes as C
    res = O.Result()
    p = O.default_params()
    O.lib().ko_search(db.h, g[100:250], 150, None, 0, C.byref(p), C.byref(res))
    buf = C.create_string_buffer(1024)
    O.lib().ko_format_match(buf, 1024, b"read1", C.byref(res), C.byref(res.matches[0]), 7)
    f = buf.value.decode().rstrip("\n").split("\t")
    assert len(f) == 15 and f[0] == "read1" and f[1] == "150" and f[2] == "130" and f[11] == "1.0000" and f[14] == "7"
    assert "e-" in f[3] or f[3] == "0.0000e+00"
    O.lib().ko_result_free(C.byref(res))
    db.close()


def test_batch_search_equals_single(oracle_lib, tmp_path):
    """ko_search_batch (the cpu_baseline leg) returns exactly what ko_search returns."""
    O = oracle_lib
    from tests import synth
    genomes = synth.random_genomes(40, 6000, seed=21)
    db = O.OracleDB(synth.make_db(tmp_path, genomes, k=21, n_chunks=2, threads=4))
    reads = synth.sample_reads(genomes, 300, 150, seed=22, frac_random=0.2, n_rate=0.005) + [b"", b"ACGT", genomes[1][:35]]
    from kmcp_amd.lib import pack_reads
    seqs, offs = pack_reads(reads)
    qk, hits = db.search_batch(seqs, offs, threads=4)
    want = []
    for i, r in enumerate(reads):
        o = db.search(r)
        assert qk[i] == o["qkmers"]
        want += [(i, m["col_global"], m["mkmers"]) for m in (o["matches"] or [])]
    assert sorted(want) == [tuple(int(x) for x in h) for h in hits]
    assert len(want) > 100
    db.close()
