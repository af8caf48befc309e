"""On-disk format handling of libkmcpgpu without a GPU (metadata-only handles, device = -1): what `.uniki` headers and
`__db.yml` say must come back through kmcpg_db_info / kmcpg_col_info / kmcpg_block_info, and broken databases must be refused
with the reference's error texts (index/serialization.go:41-57, util-db-info.go:98-129, util-db-search.go:654-695)."""
import os
import shutil

import numpy as np
import pytest

from tests import synth


@pytest.fixture(scope="module")
def small_db(oracle_lib, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("fmt")
    genomes = synth.random_genomes(21, 3000, seed=5)
    names = [f"GCF_{i:09d}.1" for i in range(21)]
    db_dir = synth.make_db(tmp, genomes, k=31, n_chunks=3, overlap=100, num_hashes=2, fpr=0.05, threads=4, scale=3, names=names)
    return db_dir, names


def test_metadata_matches_oracle_reader(oracle_lib, small_db):
    from kmcp_amd.lib import Database
    O = oracle_lib
    db_dir, names = small_db
    odb = O.OracleDB(db_dir)
    with Database.open(db_dir, device=-1) as db:
        i = db.info
        assert (i.k, i.canonical, i.num_hashes, i.scaled, i.scale, i.minimizer, i.syncmer) == (31, 1, 2, 1, 3, 0, 0)
        assert abs(i.fpr - 0.05) < 1e-15 and i.n_blocks == odb.nblocks and i.n_cols == odb.ncols == 63
        tot = 0
        for b in range(i.n_blocks):
            ns, nc, rb = odb.block_info(b)
            bi = db.block_info(b)
            assert (bi["num_sigs"], bi["n_cols"], bi["row_bytes"], bi["col_base"]) == (ns, nc, rb, tot)
            assert bi["stride"] >= rb and bi["stride"] % 16 == 0
            tot += nc
        for c in range(63):
            assert db.col_info(c) == odb.col_info(c)
        assert {db.col_info(c)[0] for c in range(63)} == set(names)
        assert i.matrix_bytes == sum(odb.block_info(b)[0] * odb.block_info(b)[2] for b in range(odb.nblocks))
    odb.close()


def _copy(db_dir, tmp_path):
    d = tmp_path / "R001"
    shutil.copytree(db_dir, d)
    return str(d)


def _open_fails(path, needle):
    from kmcp_amd.lib import Database, KmcpGpuError
    with pytest.raises(KmcpGpuError) as e:
        Database.open(path, device=-1)
    assert needle in str(e.value), str(e.value)
    return e.value.code


def test_broken_databases_are_refused(small_db, tmp_path):
    db_dir, _ = small_db
    # no __db.yml
    assert _open_fails(str(tmp_path), "fail to open kmcp database info file") == -2
    # wrong magic
    d = _copy(db_dir, tmp_path / "magic")
    with open(os.path.join(d, "_block001.uniki"), "r+b") as f:
        f.write(b".kmcpXXX")
    assert _open_fails(d, "invalid index format") == -3
    # index version
    d = _copy(db_dir, tmp_path / "ver")
    with open(os.path.join(d, "_block001.uniki"), "r+b") as f:
        f.seek(8)
        f.write(bytes([3]))
    _open_fails(d, "version mismatch")
    # truncated matrix
    d = _copy(db_dir, tmp_path / "trunc")
    p = os.path.join(d, "_block002.uniki")
    os.truncate(p, os.path.getsize(p) - 10)
    _open_fails(d, "truncated index file")
    # block listed in __db.yml but missing
    d = _copy(db_dir, tmp_path / "missing")
    os.remove(os.path.join(d, "_block001.uniki"))
    assert _open_fails(d, "kmcp index file missing") == -2
    # k in __db.yml differs from the blocks' k; database version
    d = _copy(db_dir, tmp_path / "k")
    y = open(os.path.join(d, "__db.yml")).read()
    open(os.path.join(d, "__db.yml"), "w").write(y.replace("k: 31", "k: 21").replace("- 31", "- 21"))
    _open_fails(d, "index files not compatible")
    open(os.path.join(d, "__db.yml"), "w").write(y.replace("version: 4\n", "version: 3\n", 1))
    _open_fails(d, "version mismatch")
    open(os.path.join(d, "__db.yml"), "w").write(y.split("files:")[0] + "files: []\n")
    _open_fails(d, "no index files")


def test_yaml_variants(small_db, tmp_path):
    """yaml.v2 output and hand-edited variants (flow lists, quotes, comments) parse the same."""
    from kmcp_amd.lib import Database
    db_dir, _ = small_db
    d = _copy(db_dir, tmp_path / "y")
    y = open(os.path.join(d, "__db.yml")).read()
    files = [ln[2:].strip() for ln in y.split("files:")[1].splitlines() if ln.startswith("- ")]
    head = y.split("ks:")[0]
    rest = y.split("ks:")[1].split("\n", 2)[2].split("files:")[0]
    y2 = "# edited by hand\n" + head + "ks: [31]\n" + rest + "files: [" + ", ".join(f'"{f}"' for f in files) + "]\n"
    open(os.path.join(d, "__db.yml"), "w").write(y2.replace("alias: oracle-db", "alias: 'my db'"))
    with Database.open(d, device=-1) as db:
        assert db.info.k == 31 and db.info.n_blocks == len(files) and db.info.scale == 3


def test_finalize_on_cpu_matches_oracle(oracle_lib, small_db):
    """kmcpg_finalize (float64 thresholds, FPR, Match values, sorting) fed with the oracle's raw counts: the host half of the
    product is exercised without a GPU."""
    from kmcp_amd.lib import HIT_DTYPE, Database, default_params
    O = oracle_lib
    db_dir, _ = small_db
    odb = O.OracleDB(db_dir)
    genomes = synth.random_genomes(21, 3000, seed=5)
    reads = synth.sample_reads(genomes, 120, 400, sub_rate=0.01, seed=6, frac_random=0.1)
    for flags in (dict(), dict(sort_by=2, top_n_scores=1, min_qcov=0.3), dict(sort_by=1, min_tcov=0.02, max_fpr=1e-4, min_matched=3)):
        p, op = default_params(**flags), O.default_params(**flags)
        hits, qk, ql = [], [], []
        for i, r in enumerate(reads):
            km = O.generate_kmers(r, odb.cfg)
            n = len(km) if len(km) >= p.min_matched else 0
            if n > p.dedup_threshold:
                km = O.sort_unique(km)
                n = len(km)
            qk.append(n)
            ql.append(len(r))
            if n == 0:
                continue
            tot = 0
            for b in range(odb.nblocks):
                cnt = odb.block_counts(b, km)
                for c in np.nonzero(cnt)[0]:
                    hits.append((i, tot + int(c), int(cnt[c])))  # every non-zero count: finalize must apply all thresholds itself
                tot += len(cnt)
        h = np.array(hits, dtype=np.uint32).view(HIT_DTYPE).reshape(-1) if hits else np.zeros(0, dtype=HIT_DTYPE)
        with Database.open(db_dir, device=-1) as db:
            res = db.finalize(h, np.array(qk, dtype=np.int32), np.array(ql, dtype=np.int32), params=p)
        assert synth.assert_parity(odb, res, reads, None, op) > 30
    odb.close()


def test_finalize_worker_threads_agree(small_db):
    """A hit list large enough for kmcpg_finalize to fan out over worker threads gives exactly what finalizing the same reads
    in small single-threaded pieces gives."""
    from kmcp_amd.lib import HIT_DTYPE, Database, default_params
    db_dir, _ = small_db
    rng = np.random.default_rng(3)
    n_reads, n_hits = 120000, 400000
    h = np.zeros(n_hits, dtype=HIT_DTYPE)
    h["read"] = rng.integers(0, n_reads, n_hits)
    h["col"] = rng.integers(0, 63, n_hits)
    h["count"] = rng.integers(40, 131, n_hits)
    # one (read, col) pair at most once, as the GPU emits them
    _, keep = np.unique(h["read"].astype(np.int64) * 64 + h["col"], return_index=True)
    h = h[np.sort(keep)]
    qk = rng.choice([130, 120, 97], n_reads).astype(np.int32)
    ql = np.full(n_reads, 150, dtype=np.int32)
    p = default_params(min_qcov=0.35, sort_by=2, top_n_scores=2)
    with Database.open(db_dir, device=-1) as db:
        whole = db.finalize(h, qk, ql, params=p)
        pieces, step = [], 6000
        for lo in range(0, n_reads, step):
            sel = h[(h["read"] >= lo) & (h["read"] < lo + step)].copy()
            sel["read"] -= lo
            pieces.append(db.finalize(sel, qk[lo:lo + step], ql[lo:lo + step], params=p))
        # odd worker counts, more workers than the library would pick, a single one: the ranges of reads move, the result does not
        others = []
        for t in ("1", "3", "13", "64"):
            os.environ["KMCPG_FINALIZE_THREADS"] = t
            try:
                others.append(db.finalize(h, qk, ql, params=p))
            finally:
                os.environ.pop("KMCPG_FINALIZE_THREADS", None)
    assert len(whole.matches) > 100000
    assert np.array_equal(whole.matches, np.concatenate([x.matches for x in pieces]))
    assert np.array_equal(np.diff(whole.offs), np.concatenate([np.diff(x.offs) for x in pieces]))
    for o in others:
        assert np.array_equal(o.matches, whole.matches) and np.array_equal(o.offs, whole.offs)


@pytest.mark.parametrize("spread", [False, True])
def test_finalize_orders_many_matches_per_read(small_db, spread):
    """Reads with dozens of matches (a database of close relatives) take kmcpg_finalize's key-sort path: the order must be
    Matches.Less / SortByTCov / SortByJacc (util-db-search.go:105-145) — score, then tie score, both descending — with the
    column as the last resort, for every sort mode; -S/--do-not-sort lists by column; --keep-top-scores keeps a prefix."""
    from kmcp_amd.lib import HIT_DTYPE, Database, default_params
    db_dir, _ = small_db
    rng = np.random.default_rng(17)
    n_reads, per = 3000, 40
    cols = np.argsort(rng.random((n_reads, 63)), axis=1)[:, :per]
    h = np.zeros(n_reads * per, dtype=HIT_DTYPE)
    h["read"] = np.repeat(np.arange(n_reads, dtype=np.uint32), per)
    h["col"] = cols.reshape(-1)
    h["count"] = rng.integers(20, 26, n_reads * per)  # few distinct counts: ties in qcov everywhere (-s qcov: counting sort)
    qk = rng.choice([130, 97], n_reads).astype(np.int32)
    if spread:  # long queries, counts all over the place: more distinct counts than matches (-s qcov: the general index sort)
        h["count"] = rng.integers(800, 6000, n_reads * per)
        qk = rng.choice([6000, 7001], n_reads).astype(np.int32)
    h = h[rng.permutation(len(h))]
    ql = np.full(n_reads, 150, dtype=np.int32)
    with Database.open(db_dir, device=-1) as db:
        for flags in (dict(sort_by=0), dict(sort_by=1), dict(sort_by=2), dict(do_not_sort=1)):
            p = default_params(min_qcov=0.1, max_fpr=1.0, **flags)
            res = db.finalize(h, qk, ql, params=p)
            m, offs = res.matches, res.offs
            assert len(m) == len(h)  # nothing filtered: every read keeps its 40
            for r in range(0, n_reads, 37):
                mm = m[offs[r]:offs[r + 1]]
                if flags.get("do_not_sort"):
                    key = [(int(x["col"]),) for x in mm]
                elif flags["sort_by"] == 0:
                    key = [(-x["qcov"], -x["tcov"], int(x["col"])) for x in mm]
                else:
                    key = [(-x["tcov" if flags["sort_by"] == 1 else "jacc"], -int(x["mkmers"]), int(x["col"])) for x in mm]
                assert key == sorted(key), (flags, r)
                assert sorted(int(x["col"]) for x in mm) == sorted(int(c) for c in cols[r])
            if not flags.get("do_not_sort"):
                top = db.finalize(h, qk, ql, params=default_params(min_qcov=0.1, max_fpr=1.0, top_n_scores=2, **flags))
                assert 0 < len(top.matches) < len(m)
                for r in range(0, n_reads, 37):
                    t = top.matches[top.offs[r]:top.offs[r + 1]]
                    assert 1 <= len(t) < per and np.array_equal(t, m[offs[r]:offs[r] + len(t)])


def test_dist_search_fastx_reader(tmp_path):
    """The record reader of kmcp_amd.dist_search: multi-line FASTA, FASTQ with '@' leading a quality line, gzip, empty records."""
    import gzip
    from kmcp_amd.dist_search import read_fastx
    fa = tmp_path / "a.fa"
    fa.write_text(">s1 desc\nACGT\nAC\n\n>s2\n>s3\tx\nGG\n")
    assert list(read_fastx(str(fa))) == [(b"s1", b"ACGTAC"), (b"s2", b""), (b"s3", b"GG")]
    fq = tmp_path / "a.fq.gz"
    with gzip.open(fq, "wt") as fh:
        fh.write("@q1 d\nACGT\n+\n@III\n@q2\n\n+\n\n@q3\nAC\nGT\n+q3\n>I\n@@\r\n")
    assert list(read_fastx(str(fq))) == [(b"q1", b"ACGT"), (b"q2", b""), (b"q3", b"ACGT")]


@pytest.mark.timeout(120)
def test_mutated_headers_never_hang_or_crash(small_db, tmp_path):
    """Random corruption of a block header (flipped bytes, truncation, all-ones and random 32/64-bit counts) and of __db.yml:
    every count read from the file is checked against the bytes that are left, so a bad database is refused (or, if the
    damage is harmless, opened) at once — never a multi-gigabyte allocation or a loop over 2^32 phantom entries."""
    from kmcp_amd import lib
    rng = np.random.default_rng(3)
    db_dir, _ = small_db
    good_blk = open(os.path.join(db_dir, "_block001.uniki"), "rb").read()
    good_yml = open(os.path.join(db_dir, "__db.yml"), "rb").read()
    with lib.Database.open(db_dir, device=-1) as h:
        bi = h.block_info(0)
    hdr = len(good_blk) - bi["num_sigs"] * bi["row_bytes"]  # header bytes in front of the matrix
    d = tmp_path / "db"
    shutil.copytree(db_dir, d)
    refused = opened = 0
    for i in range(400):
        b, y = bytearray(good_blk), bytearray(good_yml)
        mode = int(rng.integers(0, 6))
        if mode == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, hdr))] = int(rng.integers(0, 256))
        elif mode == 1:
            b = b[:int(rng.integers(0, hdr))]
        elif mode == 2:
            p = int(rng.integers(8, hdr - 8))
            b[p:p + 4] = b"\xff\xff\xff\xff"
        elif mode == 3:
            p = int(rng.integers(8, hdr - 8))
            b[p:p + 8] = int(rng.integers(0, 2**63)).to_bytes(8, "big")
        elif mode == 4:
            for _ in range(int(rng.integers(1, 6))):
                y[int(rng.integers(0, len(y)))] = int(rng.integers(0, 256))
        else:
            y = y[:int(rng.integers(0, len(y)))]
        (d / "_block001.uniki").write_bytes(bytes(b))
        (d / "__db.yml").write_bytes(bytes(y))
        try:
            lib.Database.open(str(d), device=-1).close()
            opened += 1
        except lib.KmcpGpuError:
            refused += 1
    assert refused > 200 and refused + opened == 400


def _reader_checksum(recs):
    """kmcp-search --parse-only: sum over records i of fnv1a("id\\tseq\\n") * (2 i + 1) mod 2^64."""
    total = 0
    for idx, (i, s) in enumerate(recs):
        h = 1469598103934665603
        for b in i + b"\t" + s + b"\n":
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        total = (total + h * (2 * idx + 1)) & 0xFFFFFFFFFFFFFFFF
    return total


def test_cli_reader_against_python_reader(tmp_path):
    """`kmcp-search --parse-only` (the CLI's block-buffered FASTA/Q reader alone: no database, no GPU) vs the independent Python
    reader of kmcp_amd.dist_search on random inputs — FASTA/FASTQ, wrapped or not, CRLF, blank lines, empty records, quality
    lines starting with '@' or '+', missing final newline, gzip — with block buffers of 16 bytes .. 1 MB so that refills land
    everywhere inside records."""
    import gzip
    import subprocess
    from kmcp_amd.dist_search import read_fastx
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmcp_amd", "kmcp-search")
    if not os.path.exists(cli):
        import __graft_entry__ as g
        g.build()
    rng = np.random.default_rng(21)

    def bgzf(data, block):
        import struct
        import zlib
        blocks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]  # + the EOF marker block
        o = bytearray()
        for b in blocks:
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            d = c.compress(b) + c.flush()
            o += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(b), len(b))
        return bytes(o)

    fnv = _reader_checksum

    for it in range(160):
        fastq = rng.random() < 0.6
        nl = b"\r\n" if rng.random() < 0.3 else b"\n"
        out = bytearray()
        for r in range(int(rng.integers(0, 30))):
            L = int(rng.choice([0, 1, 5, 30, 150, 151, 400, 3000]))
            s = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), L))
            name = b"r%d" % r + (b" some text" if rng.random() < 0.5 else b"")
            wrap = int(rng.choice([0, 0, 7, 60]))

            def lines(x):
                if wrap == 0 or len(x) == 0:
                    return x + nl
                return b"".join(x[p:p + wrap] + nl for p in range(0, len(x), wrap))
            if fastq:
                q = bytes(rng.choice(np.frombuffer(b"@+I#5>", dtype=np.uint8), L))
                out += b"@" + name + nl + lines(s) + b"+" + (name if rng.random() < 0.3 else b"") + nl + lines(q)
            else:
                out += b">" + name + nl + lines(s)
            if rng.random() < 0.1:
                out += nl
        if out and rng.random() < 0.3:
            while out and out[-1:] in (b"\n", b"\r"):
                out = out[:-1]
        p = str(tmp_path / ("f%d.%s" % (it, "fq" if fastq else "fa")))
        u = rng.random()
        if u < 0.2:
            p += ".gz"
            with gzip.open(p, "wb") as fh:
                fh.write(bytes(out))
        elif u < 0.4:  # BGZF (bgzip/htslib): inflated by several threads in kmcp-search, read as multi-member gzip by Python
            p += ".gz"
            open(p, "wb").write(bgzf(bytes(out), int(rng.choice([1, 37, 1000, 65280]))))
        else:
            open(p, "wb").write(bytes(out))
        want = list(read_fastx(p))
        env = dict(os.environ, KMCP_READER_BUF=str(int(rng.choice([16, 17, 31, 64, 100, 257, 4096, 1 << 20]))), KMCP_BGZF_THREADS="3")
        r = subprocess.run([cli, "--parse-only", p], capture_output=True, text=True, env=env, timeout=60)
        assert r.returncode == 0, (p, r.stderr)
        got = dict(x.split("=") for x in r.stdout.strip().split("\t")[1:])
        assert got == dict(records=str(len(want)), bases=str(sum(len(s) for _, s in want)), id_bytes=str(sum(len(i) for i, _ in want)),
                           fnv1a="%016x" % fnv(want)), (p, env["KMCP_READER_BUF"])


def test_parallel_fastq_reader(tmp_path):
    """Plain four-line FASTQ is cut at record boundaries and parsed by several threads (cli/fastx_reader.hpp ParallelFastq).
    Chunks of 64 bytes .. 64 KB put the cuts everywhere; quality lines start with '@' and '+' (they must never pass for a
    header), CRLF, blank lines between records, no final newline, empty reads; files that stop being four-line half-way
    (wrapped sequences, a FASTA record) are handed to the general reader at the chunk where that shows.  Always the
    records of the independent Python reader, in order."""
    import subprocess
    from kmcp_amd.dist_search import read_fastx
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmcp_amd", "kmcp-search")
    if not os.path.exists(cli):
        import __graft_entry__ as g
        g.build()
    rng = np.random.default_rng(77)
    qual = np.frombuffer(b"@+I#5>", dtype=np.uint8)
    for it in range(120):
        nl = b"\r\n" if rng.random() < 0.25 else b"\n"
        out = bytearray()
        nrec = int(rng.integers(1, 400))
        break_at = int(rng.integers(0, nrec)) if rng.random() < 0.35 else -1
        for r in range(nrec):
            L = int(rng.choice([0, 1, 30, 150, 150, 150, 151, 300]))
            s = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), L))
            q = bytes(rng.choice(qual, L))
            name = b"r%d" % r + (b" some text" if rng.random() < 0.5 else b"") + (b"\tmore" if rng.random() < 0.1 else b"")
            if r == break_at:
                kind = int(rng.integers(0, 2))
                if kind == 0 and L > 10:  # a wrapped record
                    out += b"@" + name + nl + s[:7] + nl + s[7:] + nl + b"+" + nl + q[:5] + nl + q[5:] + nl
                    continue
                if kind == 1:  # a FASTA record in the middle
                    out += b">" + name + nl + s + nl
                    continue
            out += b"@" + name + nl + s + nl + b"+" + (name if rng.random() < 0.3 else b"") + nl + q + nl
            if rng.random() < 0.05:
                out += nl
        if rng.random() < 0.3:
            while out and out[-1:] in (b"\n", b"\r"):
                out = out[:-1]
        p = str(tmp_path / ("p%d.fq" % it))
        open(p, "wb").write(bytes(out))
        want = list(read_fastx(p))
        env = dict(os.environ, KMCP_PARALLEL_MIN_BYTES="1", KMCP_READER_CHUNK=str(int(rng.choice([64, 100, 333, 1000, 4096, 65536]))),
                   KMCP_READER_THREADS=str(int(rng.integers(1, 5))), KMCP_READER_BUF=str(int(rng.choice([64, 4096, 1 << 20]))))
        r = subprocess.run([cli, "--parse-only", "-q", "--gpu-batch", str(int(rng.choice([1, 7, 100, 100000]))), p], capture_output=True, text=True, env=env,
                           timeout=60)
        assert r.returncode == 0, (p, r.stderr)
        got = dict(x.split("=") for x in r.stdout.strip().split("\t")[1:])
        assert got == dict(records=str(len(want)), bases=str(sum(len(s) for _, s in want)), id_bytes=str(sum(len(i) for i, _ in want)),
                           fnv1a="%016x" % _reader_checksum(want)), (p, env["KMCP_READER_CHUNK"], break_at)


def test_paired_reader_zips_mates_in_order(tmp_path):
    """`kmcp-search --parse-only -1 a -2 b`: the two mate files are parsed independently (several threads each for plain
    FASTQ, the general reader for gzip) and zipped back record by record.  Reads of different lengths put the chunk cuts of
    the two files at different records, so the mates are re-cut; the pairs end with the shorter file (search.go:807-826).
    Checked against the independent Python reader: checksum over "id1\tseq1\tseq2\n" in order."""
    import gzip
    import subprocess
    from kmcp_amd.dist_search import read_fastx
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmcp_amd", "kmcp-search")
    if not os.path.exists(cli):
        import __graft_entry__ as g
        g.build()
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for it in range(60):
        n1 = int(rng.integers(0, 600))
        n2 = n1 if rng.random() < 0.6 else int(rng.integers(0, 600))
        fixed = rng.random() < 0.4  # equal lengths: both files are cut at the same records (the no-copy path)
        paths = []
        for mate, n in ((1, n1), (2, n2)):
            out = bytearray()
            for r in range(n):
                L = 150 if fixed else int(rng.choice([0, 1, 30, 100, 150, 151, 250]))
                s = bytes(rng.choice(acgt, L))
                out += b"@r%d/%d\n" % (r, mate) + s + b"\n+\n" + b"I" * L + b"\n"
            p = str(tmp_path / ("m%d_%d.fq" % (it, mate)))
            if rng.random() < 0.25:
                p += ".gz"
                with gzip.open(p, "wb") as fh:
                    fh.write(bytes(out))
            else:
                open(p, "wb").write(bytes(out))
            paths.append(p)
        a, b = list(read_fastx(paths[0])), list(read_fastx(paths[1]))
        m = min(len(a), len(b))
        want = [(a[i][0], a[i][1] + b"\t" + b[i][1]) for i in range(m)]
        env = dict(os.environ, KMCP_PARALLEL_MIN_BYTES="1", KMCP_READER_CHUNK=str(int(rng.choice([64, 333, 1000, 4096, 65536]))),
                   KMCP_READER_THREADS=str(int(rng.integers(1, 5))))
        r = subprocess.run([cli, "--parse-only", "-q", "--gpu-batch", str(int(rng.choice([1, 7, 100, 100000]))), "-1", paths[0], "-2", paths[1]],
                           capture_output=True, text=True, env=env, timeout=60)
        assert r.returncode == 0, (paths, r.stderr)
        got = dict(x.split("=") for x in r.stdout.strip().split("\t")[1:])
        assert got == dict(records=str(m), bases=str(sum(len(x[1]) + len(y[1]) for x, y in zip(a[:m], b[:m]))), id_bytes=str(sum(len(x[0]) for x in a[:m])),
                           fnv1a="%016x" % _reader_checksum(want)), (paths, env["KMCP_READER_CHUNK"], n1, n2)


def test_truncated_gzip_is_an_error(tmp_path):
    """A .fastq.gz cut short (or with a damaged tail) must not produce a well-formed result for the part that could be read:
    the reference's gzip reader aborts with 'unexpected EOF'."""
    import gzip
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmcp_amd", "kmcp-search")
    rng = np.random.default_rng(5)
    recs = b"".join(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150)) + b"\n+\n" + b"I" * 150 + b"\n" for i in range(3000))
    good = str(tmp_path / "good.fq.gz")
    with gzip.open(good, "wb") as fh:
        fh.write(recs)
    data = open(good, "rb").read()
    r = subprocess.run([cli, "--parse-only", "-q", good], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "records=3000" in r.stdout
    for name, blob in (("cut", data[:len(data) // 2]), ("crc", data[:-8] + bytes(8)), ("tail", data[:-3])):
        p = str(tmp_path / (name + ".fq.gz"))
        open(p, "wb").write(blob)
        r = subprocess.run([cli, "--parse-only", "-q", p], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "records=" not in r.stdout, (name, r.stdout, r.stderr)


def test_xz_and_bzip2_input_through_the_system_decompressor(tmp_path):
    """The reference's xopen reads xz, zstd and bzip2 besides gzip (util-io.go:68-97): kmcp-search runs the system's decompressor
    for those and parses its output; a damaged file is an error, a missing tool a message that names it."""
    import shutil
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmcp_amd", "kmcp-search")
    rng = np.random.default_rng(9)
    recs = [(b"r%d" % i, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150))) for i in range(2000)]
    plain = b"".join(b"@" + i + b" x\n" + s + b"\n+\n" + b"I" * 150 + b"\n" for i, s in recs)
    want = dict(records="2000", bases=str(150 * 2000), id_bytes=str(sum(len(i) for i, _ in recs)), fnv1a="%016x" % _reader_checksum(recs))
    done = 0
    for tool, ext in (("xz", "xz"), ("bzip2", "bz2"), ("zstd", "zst")):
        if not shutil.which(tool):
            continue
        p = str(tmp_path / ("reads.fq." + ext))
        open(p, "wb").write(subprocess.run([tool, "-c"], input=plain, capture_output=True, check=True).stdout)
        r = subprocess.run([cli, "--parse-only", "-q", p], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        assert dict(x.split("=") for x in r.stdout.strip().split("\t")[1:]) == want
        blob = open(p, "rb").read()
        open(p, "wb").write(blob[:len(blob) // 2])
        r = subprocess.run([cli, "--parse-only", "-q", p], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "failed" in r.stderr
        # the tool missing from PATH
        open(p, "wb").write(blob)
        r = subprocess.run([cli, "--parse-only", "-q", p], capture_output=True, text=True, timeout=60, env=dict(os.environ, PATH="/nonexistent"))
        assert r.returncode != 0 and "not on PATH" in r.stderr
        done += 1
    if not done:
        pytest.skip("no xz/bzip2/zstd on this box")


def test_multi_k_database_metadata(oracle_lib, tmp_path):
    """`ks: [21, 31]` in __db.yml (block or flow style): the handle reports the largest k (the .uniki headers carry it,
    util-db-search.go:690) and refuses a database whose headers disagree with it."""
    import re
    from kmcp_amd import Database, lib
    O = oracle_lib
    genomes = synth.random_genomes(4, 2000, seed=8)
    cols = [(f"g{i}", len(g), 0, 1, O.sort_unique(np.concatenate([O.generate_kmers(g, O.sketch_cfg(k=k)) for k in (21, 31)]))) for i, g in enumerate(genomes)]
    db_dir = O.build_db(str(tmp_path / "a"), O.sketch_cfg(k=31), cols, num_hashes=1, fpr=0.3, threads=1)
    yml = open(db_dir + "/__db.yml").read()
    for style in ("ks:\n- 21\n- 31\n", "ks: [31, 21]\n"):
        open(db_dir + "/__db.yml", "w").write(re.sub(r"ks:\n- 31\n", style, yml))
        with Database.open(db_dir, device=-1) as db:
            assert db.info.k == 31 and int(db.info.n_cols) == 4
        odb = O.OracleDB(db_dir)
        assert odb.cfg.k == 31
        odb.close()
    open(db_dir + "/__db.yml", "w").write(re.sub(r"ks:\n- 31\n", "ks:\n- 31\n- 33\n", yml))  # headers say 31, the list's maximum is 33
    with pytest.raises(lib.KmcpGpuError):
        Database.open(db_dir, device=-1)
