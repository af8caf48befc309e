"""kmcp-search (C++ host above the C ABI) vs the oracle's TSV lines: the 15 columns of kmcp/cmd/search.go:517-575,
the header (:436-438) and the 3-line trailer (:1022-1025) that `kmcp profile` parses."""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "kmcp_amd", "kmcp-search")
HEADER = "#query\tqLen\tqKmers\tFPR\thits\ttarget\tchunkIdx\tchunks\ttLen\tkSize\tmKmers\tqCov\ttCov\tjacc\tqueryIdx"


def oracle_tsv(O, odb, ids, reads, reads2=None, params=None, keep_unmatched=False, name_map=None):
    """What `kmcp search` would print."""
    L = O.lib()
    p = params or O.default_params()
    lines, matched = [], 0
    buf = C.create_string_buffer(4096)
    for i, r in enumerate(reads):
        res = O.Result()
        r2 = reads2[i] if reads2 is not None else None
        L.ko_search(odb.h, r, len(r), r2, len(r2) if r2 is not None else 0, C.byref(p), C.byref(res))
        if res.nmatches < 0:
            if keep_unmatched:
                lines.append(f"{ids[i]}\t{res.qlen}\t{res.qkmers}\t0\t0\t\t-1\t0\t0\t{res.k}\t0\t0\t0\t0\t{i}")
        else:
            matched += 1
            for j in range(res.nmatches):
                L.ko_format_match(buf, 4096, ids[i].encode(), C.byref(res), C.byref(res.matches[j]), i)
                f = buf.value.decode().rstrip("\n").split("\t")
                if name_map and f[5] in name_map:
                    f[5] = name_map[f[5]]
                lines.append("\t".join(f))
        L.ko_result_free(C.byref(res))
    n = len(reads)
    trailer = [f"# input queries: {n}", f"# matched queries: {matched}", "# matched percentage: %.4f%%" % (matched / n * 100)]
    return lines, trailer


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_cli(args, out_path, env=None):
    r = subprocess.run([CLI] + args + ["-o", out_path, "-q"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    opener = gzip.open if out_path.endswith(".gz") else open
    with opener(out_path, "rt") as fh:
        return fh.read().split("\n")


def compare(got_lines, want_rows, want_trailer, header=True):
    assert got_lines[-1] == ""
    got = got_lines[:-1]
    if header:
        assert got[0] == HEADER
        got = got[1:]
    assert got[-3:] == want_trailer
    rows = got[:-3]
    assert len(rows) == len(want_rows)
    for g, w in zip(rows, want_rows):
        assert g == w, (g, w)  # every field as a string, the FPR's %.4e digits included (pinned: tests/test_fpr_golden.py)


def write_fastq(path, ids, reads, gz=False):
    opener = gzip.open if gz else open
    with opener(path, "wt") as fh:
        for i, r in zip(ids, reads):
            fh.write(f"@{i} some description\n{r.decode()}\n+\n{'I' * len(r)}\n")


def write_fasta(path, recs, width=70):
    with open(path, "w") as fh:
        for name, s in recs:
            fh.write(f">{name} desc\n")
            s = s.decode()
            for i in range(0, len(s), width):
                fh.write(s[i:i + width] + "\n")


def test_cli_single_end_gz_and_flags(oracle_lib, tmp_path):
    O = oracle_lib
    genomes = synth.random_genomes(20, 12000, seed=40)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=3, overlap=150, threads=4)
    db_root = os.path.dirname(db_dir)
    reads = synth.sample_reads(genomes, 700, 150, sub_rate=0.01, seed=41, frac_random=0.15) + [b"ACGT", genomes[0][:29]]
    ids = [f"read{i}/1" for i in range(len(reads))]
    fq = str(tmp_path / "reads.fq.gz")
    write_fastq(fq, ids, reads, gz=True)
    odb = O.OracleDB(db_dir)
    # default flags, gz in, gz out, small GPU batches so that several batches are stitched in order
    want, trailer = oracle_tsv(O, odb, ids, reads)
    compare(run_cli(["-d", db_root, fq, "--gpu-batch", "100"], str(tmp_path / "o1.tsv.gz")), want, trailer)
    # one process driving several shards (here two on the same GPU)
    compare(run_cli(["-d", db_root, fq, "--gpu-ids", "0,0"], str(tmp_path / "o1b.tsv")), want, trailer)
    # ... and the multi-GPU handle's RCCL exchange (one rank here: KMCPG_RCCL=force), announced in the log
    r = subprocess.run([CLI, "-d", db_root, fq, "--gpu-ids", "0", "-o", str(tmp_path / "o1c.tsv")], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, KMCPG_RCCL="force"))
    assert r.returncode == 0, r.stderr
    assert "exchange of the hit lists: RCCL gather over 1 device(s)" in r.stderr, r.stderr
    compare(open(tmp_path / "o1c.tsv").read().split("\n"), want, trailer)
    # the log carries an order-independent checksum of the (queryIdx, column, mKmers) tuples: the same however the index is spread
    # over GPUs and however the input is cut into batches — what a first run on real multi-GPU hardware is compared by
    import re
    sums = {re.search(r"matches: (\d+), checksum ([0-9a-f]{16})", r.stderr).groups()}
    for extra in (["--gpu-ids", "0,0"], ["--gpu-batch", "64"], []):
        r2 = subprocess.run([CLI, "-d", db_root, fq, "-o", str(tmp_path / "o1d.tsv")] + extra, capture_output=True, text=True, timeout=300)
        assert r2.returncode == 0, r2.stderr
        sums.add(re.search(r"matches: (\d+), checksum ([0-9a-f]{16})", r2.stderr).groups())
    assert len(sums) == 1 and int(next(iter(sums))[0]) == len(want), sums
    # -K keeps unmatched rows; -H drops the header; thresholds + sort by jacc + top score
    p = O.default_params(min_qcov=0.4, min_matched=5, sort_by=2, top_n_scores=1)
    want, trailer = oracle_tsv(O, odb, ids, reads, params=p, keep_unmatched=True)
    compare(run_cli(["-d", db_root, fq, "-K", "-H", "-t", "0.4", "-c", "5", "-s", "jacc", "-n", "1"], str(tmp_path / "o2.tsv")), want, trailer,
            header=False)
    # name map
    nm = {f"g{i:05d}": f"Genome number {i}" for i in range(0, 20, 2)}
    with open(tmp_path / "name.map", "w") as fh:
        for k, v in nm.items():
            fh.write(f"{k}\t{v}\n")
    want, trailer = oracle_tsv(O, odb, ids, reads, name_map=nm)
    compare(run_cli(["-d", db_root, fq, "-N", str(tmp_path / "name.map")], str(tmp_path / "o3.tsv")), want, trailer)
    odb.close()


def test_cli_over_every_visible_gpu(oracle_lib, tmp_path):
    """`kmcp-search --gpus N` when the box shows N >= 2 GPUs: the index blocks partitioned over the devices inside one process, the
    hit lists gathered over RCCL (announced in the log), the TSV byte-identical to the oracle's and the log's checksum equal to the
    one-GPU run's.  Skipped, with the reason, on a one-GPU box (there `--gpu-ids 0,0` and KMCPG_RCCL=force cover the code paths)."""
    import re
    import torch
    n_dev = min(torch.cuda.device_count(), 8)
    if n_dev < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: --gpus N needs N devices")
    O = oracle_lib
    genomes = synth.random_genomes(40, 12000, seed=45)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=2, overlap=150, threads=8)
    db_root = os.path.dirname(db_dir)
    reads = synth.sample_reads(genomes, 3000, 150, sub_rate=0.01, seed=46, frac_random=0.15)
    ids = [f"read{i}" for i in range(len(reads))]
    fq = str(tmp_path / "reads.fq")
    write_fastq(fq, ids, reads)
    odb = O.OracleDB(db_dir)
    want, trailer = oracle_tsv(O, odb, ids, reads)
    odb.close()
    sums = set()
    for extra in (["--gpus", str(n_dev)], ["--gpus", str(n_dev), "--gpu-batch", "500"], ["--gpus", "2"], []):
        out = str(tmp_path / "o.tsv")
        r = subprocess.run([CLI, "-d", db_root, fq, "-o", out] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        if extra:
            n = int(extra[1])
            assert f"exchange of the hit lists: RCCL gather over {n} device(s)" in r.stderr, r.stderr
        compare(open(out).read().split("\n"), want, trailer)
        sums.add(re.search(r"matches: (\d+), checksum ([0-9a-f]{16})", r.stderr).groups())
    assert len(sums) == 1 and int(next(iter(sums))[0]) == len(want), sums


def test_cli_paired_end_and_whole_file(oracle_lib, tmp_path):
    O = oracle_lib
    genomes = synth.random_genomes(10, 20000, seed=42)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=2, overlap=150, threads=2)
    db_root = os.path.dirname(db_dir)
    odb = O.OracleDB(db_dir)
    r1 = synth.sample_reads(genomes, 300, 150, seed=43, frac_random=0.0)
    r2 = synth.sample_reads(genomes, 300, 150, seed=44, frac_random=0.5)
    ids = [f"pair{i}" for i in range(300)]
    write_fastq(str(tmp_path / "r1.fq"), ids, r1)
    write_fastq(str(tmp_path / "r2.fq"), ids, r2)
    p = O.default_params(try_se=1, fpr_buf_size=499)
    want, trailer = oracle_tsv(O, odb, ids, r1, r2, params=p)
    compare(run_cli(["-d", db_root, "-1", str(tmp_path / "r1.fq"), "-2", str(tmp_path / "r2.fq"), "--try-se"], str(tmp_path / "pe.tsv")), want, trailer)
    # -g: records 2..m are each followed by k-1 N's (search.go:899-914); query id = first record / file name
    recs = [("ctgA", genomes[0][:3000]), ("ctgB", genomes[0][5000:7000]), ("ctgC", genomes[1][100:900])]
    fa = str(tmp_path / "asm.fasta")
    write_fasta(fa, recs)
    whole = recs[0][1] + recs[1][1] + b"N" * 20 + recs[2][1] + b"N" * 20
    p = O.default_params(min_qcov=0.35)
    want, trailer = oracle_tsv(O, odb, ["ctgA"], [whole], params=p)
    compare(run_cli(["-d", db_root, "-g", "-t", "0.35", fa], str(tmp_path / "g.tsv")), want, trailer)
    want, trailer = oracle_tsv(O, odb, ["asm"], [whole], params=p)
    compare(run_cli(["-d", db_root, "-g", "-G", "-t", "0.35", fa], str(tmp_path / "g2.tsv")), want, trailer)
    # several files = several queries in one batch, packed 4 bases to a byte as they are read (kmcpg_pack2 / kmcpg_submit_packed): records
    # of every length modulo 4, soft-masked stretches, N runs, IUPAC codes, U — and an assembly that matches nothing
    rng = np.random.default_rng(5)
    files, wholes, qids = [], [], []
    for fi in range(7):
        recs = []
        for ri in range(1 + fi % 4):
            g = genomes[(fi + ri) % len(genomes)]
            a = int(rng.integers(0, len(g) - 6000))
            s = bytearray(g[a:a + 1500 + int(rng.integers(0, 3000)) + ri])
            if ri % 2:
                s[100:400] = bytes(s[100:400]).lower()
            if fi % 3 == 1:
                s[700:700 + 37 + ri] = b"N" * (37 + ri)
                for pp in rng.integers(0, len(s), size=6):
                    s[int(pp)] = int(rng.choice(list(b"RYKMSWn")))
            if fi == 5:
                s = bytearray(bytes(s).replace(b"T", b"U"))
            if fi == 6:
                s = bytearray(rng.choice(list(b"ACGT"), len(s)).astype(np.uint8).tobytes())
            recs.append((f"f{fi}c{ri}", bytes(s)))
        fa_i = str(tmp_path / f"asm{fi}.fasta")
        write_fasta(fa_i, recs)
        files.append(fa_i)
        qids.append(recs[0][0])
        wholes.append(recs[0][1] + b"".join(r[1] + b"N" * 20 for r in recs[1:]))
    want, trailer = oracle_tsv(O, odb, qids, wholes, params=p)
    assert len(want) >= 6
    compare(run_cli(["-d", db_root, "-g", "-t", "0.35"] + files, str(tmp_path / "g3.tsv")), want, trailer)
    compare(run_cli(["-d", db_root, "-g", "-t", "0.35", "--gpu-batch", "3"] + files, str(tmp_path / "g4.tsv")), want, trailer)
    odb.close()


def test_dist_search_module_matches_oracle(oracle_lib, tmp_path):
    """python -m kmcp_amd.dist_search (the one-process-per-GPU driver; one rank here) prints the same TSV."""
    import sys
    O = oracle_lib
    genomes = synth.random_genomes(12, 9000, seed=46)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=2, overlap=150, threads=3)
    reads = synth.sample_reads(genomes, 400, 150, sub_rate=0.01, seed=47, frac_random=0.2) + [b"", b"ACGTACGT"]
    ids = [f"r{i}" for i in range(len(reads))]
    fq = str(tmp_path / "reads.fq.gz")
    write_fastq(fq, ids, reads, gz=True)
    odb = O.OracleDB(db_dir)
    p = O.default_params(min_qcov=0.45, sort_by=1)
    want, trailer = oracle_tsv(O, odb, ids, reads, params=p, keep_unmatched=True)
    odb.close()
    out = str(tmp_path / "d.tsv")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "kmcp_amd.dist_search", "-d", os.path.dirname(db_dir), fq, "-o", out, "-t", "0.45", "-s", "tcov", "-K",
                        "--gpu-batch", "128"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    compare(open(out).read().split("\n"), want, trailer)
    # three processes, each holding a shard of the blocks (all on GPU 0 here, exchange over gloo instead of RCCL)
    out3 = str(tmp_path / "d3.tsv")
    env["KMCP_DIST_SAME_GPU"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "-m", "kmcp_amd.dist_search", "-d", os.path.dirname(db_dir), fq, "-o", out3, "-t", "0.45", "-s", "tcov",
                        "-K", "--gpu-batch", "128"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out3).read() == open(out).read()


def test_cli_errors(tmp_path, oracle_lib):
    O = oracle_lib
    genomes = synth.random_genomes(3, 3000, seed=45)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, threads=2)
    db_root = os.path.dirname(db_dir)
    fq = str(tmp_path / "r.fq")
    write_fastq(fq, ["a"], [genomes[0][:150]])
    # search refuses min-query-cov <= db fpr (search.go:405-409); exit status of checkError is 255
    r = subprocess.run([CLI, "-d", db_root, "-t", "0.3", fq], capture_output=True, text=True)
    assert r.returncode == 255 and "should not be smaller than FPR" in r.stderr
    r = subprocess.run([CLI, fq], capture_output=True, text=True)
    assert r.returncode == 255 and "flag -d/--db-dir needed" in r.stderr
    r = subprocess.run([CLI, "-d", str(tmp_path), fq], capture_output=True, text=True)
    assert r.returncode == 255 and "invalid kmcp database" in r.stderr
    _ = O


def test_cli_reader_torture(oracle_lib, tmp_path):
    """The block-buffered reader: wrapped FASTA with CRLF line ends, a 9-Mbp record on one line (longer than the 8-MB block
    buffer), an empty record, no newline at the end of the file; and a FASTQ whose records straddle block refills, with wrapped
    sequence/quality lines and '@'/'+' as first quality characters."""
    O = oracle_lib
    genomes = synth.random_genomes(6, 30000, seed=90)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, threads=2)
    db_root = os.path.dirname(db_dir)
    odb = O.OracleDB(db_dir)
    rng = np.random.default_rng(91)
    big = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 9_000_000))
    big = big[:4_000_000] + genomes[2][1000:21000] + big[4_000_000:]
    recs = [("w0", genomes[0][:3000]), ("big", big), ("w2", genomes[1][500:1000]), ("s3", genomes[3][:777]), ("e4", b""), ("last", genomes[4][100:400])]
    fa = str(tmp_path / "t.fa")
    with open(fa, "wb") as fh:
        for i, (name, s) in enumerate(recs):
            nl = b"\r\n" if i % 2 == 0 else b"\n"
            fh.write(b">" + name.encode() + b" desc" + nl)
            if name in ("big", "s3"):
                fh.write(s + nl)
            else:
                for p in range(0, len(s), 60):
                    fh.write(s[p:p + 60] + nl)
        fh.seek(fh.tell() - 1)
        fh.truncate()  # no newline at the end of the file
    ids = [n for n, _ in recs]
    seqs = [s for _, s in recs]
    p = O.default_params(min_qcov=0.35)
    want, trailer = oracle_tsv(O, odb, ids, seqs, params=p, keep_unmatched=True)
    compare(run_cli(["-d", db_root, "-t", "0.35", "-K", fa], str(tmp_path / "fa.tsv")), want, trailer)
    # FASTQ: 60k records (~13 MB: several block refills), every 7th wrapped over three lines, quality lines starting with @ or +
    reads = synth.sample_reads(genomes, 60000, 100, seed=92, frac_random=0.2)
    fq = str(tmp_path / "t.fq")
    with open(fq, "wb") as fh:
        for i, r in enumerate(reads):
            q = (b"@" if i % 3 == 0 else b"+" if i % 3 == 1 else b"I") + b"I" * (len(r) - 1)
            if i % 7 == 0:
                fh.write(b"@q%d x\n" % i + r[:40] + b"\n" + r[40:80] + b"\n" + r[80:] + b"\n+q%d\n" % i + q[:50] + b"\n" + q[50:] + b"\n")
            else:
                fh.write(b"@q%d\n" % i + r + b"\n+\n" + q + b"\n")
    ids = [f"q{i}" for i in range(len(reads))]
    want, trailer = oracle_tsv(O, odb, ids, reads)
    compare(run_cli(["-d", db_root, fq], str(tmp_path / "fq.tsv")), want, trailer)
    odb.close()


def test_cli_empty_and_ragged_inputs(oracle_lib, tmp_path):
    """An empty file, a file without any record, paired files of different lengths (the shorter one ends the run, search.go:818-826),
    -g on an empty file: no crash, header + trailer as the reference prints them (0 queries: 'NaN%')."""
    O = oracle_lib
    genomes = synth.random_genomes(4, 3000, seed=99)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, threads=2)
    db_root = os.path.dirname(db_dir)
    empty = str(tmp_path / "empty.fq")
    open(empty, "w").close()
    blank = str(tmp_path / "blank.fa")
    open(blank, "w").write("\n\n\n")
    for args in ([empty], [blank], ["-g", empty], [empty, blank]):
        got = run_cli(["-d", db_root] + args, str(tmp_path / "e.tsv"))
        assert got == [HEADER, "# input queries: 0", "# matched queries: 0", "# matched percentage: NaN%", ""], (args, got)
    r1 = synth.sample_reads(genomes, 30, 150, seed=100, frac_random=0.0)
    r2 = synth.sample_reads(genomes, 30, 150, seed=101, frac_random=0.0)
    ids = [f"p{i}" for i in range(30)]
    write_fastq(str(tmp_path / "a.fq"), ids, r1)
    write_fastq(str(tmp_path / "b.fq"), ids[:17], r2[:17])
    odb = O.OracleDB(db_dir)
    want, trailer = oracle_tsv(O, odb, ids[:17], r1[:17], r2[:17], params=O.default_params(fpr_buf_size=499))
    odb.close()
    compare(run_cli(["-d", db_root, "-1", str(tmp_path / "a.fq"), "-2", str(tmp_path / "b.fq")], str(tmp_path / "pe.tsv")), want, trailer)
    compare(run_cli(["-d", db_root, "-1", str(tmp_path / "b.fq"), "-2", str(tmp_path / "a.fq")], str(tmp_path / "pe2.tsv")),
            *oracle_tsv_swapped(O, db_dir, ids[:17], r2[:17], r1[:17]))


def test_cli_paired_ragged_reads_through_the_parallel_reader(oracle_lib, tmp_path):
    """Paired files whose reads have different lengths: the two parser pools cut their files at different records and the
    mates are re-cut at read 1's batch boundaries (cli/kmcp_search.cpp read_paired).  Small chunks and batches put the cuts
    everywhere; the TSV must be the oracle's for the pairs in file order."""
    O = oracle_lib
    genomes = synth.random_genomes(5, 4000, seed=31)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, threads=2)
    db_root = os.path.dirname(db_dir)
    rng = np.random.default_rng(8)
    n = 700
    r1 = [synth.sample_reads(genomes, 1, int(rng.choice([60, 100, 150, 151, 250])), seed=1000 + i)[0] for i in range(n)]
    r2 = [synth.sample_reads(genomes, 1, int(rng.choice([60, 100, 150, 151, 250])), seed=5000 + i)[0] for i in range(n)]
    ids = [f"pair{i}" for i in range(n)]
    write_fastq(str(tmp_path / "a.fq"), ids, r1)
    write_fastq(str(tmp_path / "b.fq"), ids, r2)
    odb = O.OracleDB(db_dir)
    want, trailer = oracle_tsv(O, odb, ids, r1, r2, params=O.default_params(fpr_buf_size=499))
    odb.close()
    for chunk, batch in ((1000, 64), (7777, 100), (1 << 20, 100000)):
        env = dict(os.environ, KMCP_PARALLEL_MIN_BYTES="1", KMCP_READER_CHUNK=str(chunk), KMCP_READER_THREADS="3")
        got = run_cli(["-d", db_root, "-1", str(tmp_path / "a.fq"), "-2", str(tmp_path / "b.fq"), "--gpu-batch", str(batch)], str(tmp_path / "pe.tsv"), env=env)
        compare(got, want, trailer)


def test_cli_long_reads_through_the_parallel_reader(oracle_lib, tmp_path):
    """A plain FASTQ file of ~10-kb reads (12 MB: above the parallel reader's threshold without any test knob): several parser
    threads, chunks capped in bytes, the syncmer index's dedup path.  Same TSV as through the serial reader and with tiny
    chunks; the oracle's lines for the whole file."""
    O = oracle_lib
    genomes = synth.random_genomes(6, 60000, seed=71)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, threads=2, syncmer_s=11)
    db_root = os.path.dirname(db_dir)
    rng = np.random.default_rng(72)
    reads = []
    for i in range(600):
        L = int(np.clip(rng.normal(10000, 2000), 2000, 20000))
        reads.append(synth.sample_reads(genomes, 1, L, sub_rate=0.002, seed=7000 + i, frac_random=0.1)[0])
    ids = [f"hifi{i}" for i in range(len(reads))]
    fq = str(tmp_path / "long.fq")
    write_fastq(fq, ids, reads)
    assert os.path.getsize(fq) > (8 << 20)
    odb = O.OracleDB(db_dir)
    want, trailer = oracle_tsv(O, odb, ids, reads)
    odb.close()
    assert len(want) > 400
    base = run_cli(["-d", db_root, fq], str(tmp_path / "a.tsv"))
    compare(base, want, trailer)
    serial = run_cli(["-d", db_root, fq], str(tmp_path / "b.tsv"), env=dict(os.environ, KMCP_SERIAL_READER="1"))
    small = run_cli(["-d", db_root, fq, "--gpu-batch", "50"], str(tmp_path / "c.tsv"), env=dict(os.environ, KMCP_READER_CHUNK="300000", KMCP_READER_THREADS="3"))
    assert base == serial == small


def oracle_tsv_swapped(O, db_dir, ids, a, b):
    odb = O.OracleDB(db_dir)
    try:
        return oracle_tsv(O, odb, ids, a, b, params=O.default_params(fpr_buf_size=499))
    finally:
        odb.close()


def test_cli_fpr_column_reproduces_the_reference_tutorial(oracle_lib, tmp_path):
    """The FPR digits end to end: reads whose best match has 83 / 84 / 86 / 89 / ... / 130 of 130 k-mers must print the very
    strings of the reference's demo result (docs/tutorial/profiling/index.md:203-211, tests/golden/tutorial_profiling_fpr.json:
    default index FPR 0.3, 150-bp reads, k = 21) in column 4 of kmcp-search's TSV."""
    import json
    O = oracle_lib
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tutorial_profiling_fpr.json")))
    want_fpr = {r["mKmers"]: r["FPR"] for r in gold["rows"]}
    genomes = synth.random_genomes(8, 20000, seed=70)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=2, overlap=150, fpr=gold["db_fpr"], threads=2)
    odb = O.OracleDB(db_dir)
    # reads with 2-3 substitutions lose 40-50 k-mers; keep one read per golden mKmers value (the oracle says which they are)
    cand = synth.sample_reads(genomes, 4000, 150, sub_rate=0.017, seed=71, frac_random=0.0)
    pick = {}
    for r in cand:
        ms = odb.search(r)["matches"] or []
        for m in ms:
            if m["mkmers"] in want_fpr and m["mkmers"] not in pick:
                pick[m["mkmers"]] = r
    assert set(pick) == set(want_fpr), sorted(set(want_fpr) - set(pick))
    reads = [pick[m] for m in sorted(pick)]
    ids = [f"m{m}" for m in sorted(pick)]
    fq = str(tmp_path / "r.fq")
    write_fastq(fq, ids, reads)
    got = run_cli(["-d", os.path.dirname(db_dir), fq], str(tmp_path / "o.tsv"))
    want, trailer = oracle_tsv(O, odb, ids, reads)
    compare(got, want, trailer)
    seen = set()
    for line in got[1:-4]:
        f = line.split("\t")
        assert f[2] == "130"
        if int(f[10]) in want_fpr:
            assert f[3] == want_fpr[int(f[10])], line
            seen.add(int(f[10]))
    assert seen == set(want_fpr)
    odb.close()


def test_kmcp_search_spelling_and_profile_contract(oracle_lib, tmp_path):
    """`kmcp search ...` (dispatcher, cli/kmcp_dispatch.cpp) and `kmcp-search search ...` print what `kmcp-search ...` prints, and
    that result read by the rules `kmcp profile` applies to its input (tests/profile_contract.py: util-profile.go:94-182,
    profile.go:1939-1962) yields the oracle's values."""
    from tests import profile_contract as PC
    O = oracle_lib
    genomes = synth.random_genomes(12, 15000, seed=80)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=3, overlap=150, threads=4, names=[f"GCF_{i:09d}.1" for i in range(12)])
    db_root = os.path.dirname(db_dir)
    reads = synth.sample_reads(genomes, 1500, 150, sub_rate=0.02, seed=81, frac_random=0.2)
    ids = [f"r{i}/1" for i in range(len(reads))]
    fq = str(tmp_path / "r.fq.gz")
    write_fastq(fq, ids, reads, gz=True)
    outs = []
    for argv in ([CLI], [CLI, "search"], [os.path.join(ROOT, "kmcp_amd", "kmcp"), "search"], [os.path.join(ROOT, "kmcp_amd", "kmcp"), "-q", "-j", "8", "search"]):
        out = str(tmp_path / f"o{len(outs)}.tsv.gz")
        r = subprocess.run(argv + ["-d", db_root, fq, "-o", out, "-q"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (argv, r.stderr)
        outs.append(gzip.open(out, "rt").read())
    assert outs[0] == outs[1] == outs[2] == outs[3]
    odb = O.OracleDB(db_dir)
    want, trailer = oracle_tsv(O, odb, ids, reads)
    compare(outs[0].split("\n"), want, trailer)
    ms, total, stats = PC.read_search_result(outs[0], max_fpr=0.05, min_qcov=0.55)
    assert total == len(reads) and stats["matched queries"] == trailer[1].split(": ")[1]
    # what profile would work with: one record per (query, target chunk) that passes its two filters, values as the oracle has them
    expect = []
    for i, r in enumerate(reads):
        o = odb.search(r)
        for m in (o["matches"] or []):
            if float("%.4f" % m["qcov"]) >= 0.55 and float("%.4e" % m["fpr"]) <= 0.05:
                name, tidx, gsize, _ = odb.col_info(m["col_global"])
                expect.append((ids[i], o["qlen"], o["qkmers"], len(o["matches"]), name, tidx & 0xffff, tidx >> 16, gsize, 21, m["mkmers"]))
    got = [(m["query"], m["qlen"], m["qkmers"], m["hits"], m["target"], m["chunk_idx"], m["chunks"], m["gsize"], m["k"], m["mkmers"]) for m in ms]
    assert got == expect and len(got) > 500
    odb.close()


def test_shim_fixture_through_kmcp_search(tmp_path):
    """shim/testdata (the fixture of the Go-side test of the cgo binding, shim/kmcp_gpu_test.go): kmcp-search on the GPU prints
    exactly the committed TSV — the file the Go test diffs its rows against (tests/test_shim_fixture_cpu.py ties it to the oracle)."""
    fix = os.path.join(ROOT, "shim", "testdata")
    got = run_cli(["-d", os.path.join(fix, "db"), os.path.join(fix, "reads.fq")], str(tmp_path / "o.tsv"))
    assert got == open(os.path.join(fix, "expected.tsv")).read().split("\n")
