"""kmcp-merge (host-side fan-in of search results from several databases / nodes) vs a line-by-line Python restatement of
kmcp/cmd/merge.go:190-262,283-341,386-388: k-way merge on queryIdx, per-query re-sort by the parsed score, `hits` rewritten,
trailer recomputed.  Inputs are the oracle's `kmcp search` TSVs of the same reads against two disjoint databases."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests import synth
from tests.test_gpu_cli import HEADER, oracle_tsv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MERGE = os.path.join(ROOT, "kmcp_amd", "kmcp-merge")


def write_tsv(path, rows, trailer, header=True):
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wt") as fh:
        if header:
            fh.write(HEADER + "\n")
        for r in rows:
            fh.write(r + "\n")
        for t in trailer:
            fh.write(t + "\n")


def merge_restated(files_rows, totals, sort_by="qcov", f_qidx=15, f_hits=5):
    """files_rows: per input, the list of data rows.  Equal scores keep (input order, row order) — the tool's documented choice
    where the reference's order is unspecified."""
    sf = f_qidx - {"qcov": 3, "tcov": 2, "jacc": 1}[sort_by] - 1
    by_q = {}
    for fi, rows in enumerate(files_rows):
        for ri, r in enumerate(rows):
            f = r.split("\t", f_qidx - 1)
            by_q.setdefault(int(f[f_qidx - 1]), []).append((-float(f[sf]), fi, ri, f))
    out = []
    for q in sorted(by_q):
        grp = sorted(by_q[q], key=lambda t: (t[0], t[1], t[2]))
        for _, _, _, f in grp:
            f[f_hits - 1] = str(len(grp))
            out.append("\t".join(f))
    matched = max(1, len(by_q))  # merge.go:243: the closing `matched++` also runs when there was no row at all
    total = totals[0]
    for n in totals[1:]:
        if total == 0 or n != 0:
            total = n
    return out, [f"# input queries: {total}", f"# matched queries: {matched}", "# matched percentage: %.4f%%" % (matched / total * 100)]


def run(args, expect=0):
    if not os.path.exists(MERGE):
        import __graft_entry__ as g
        g.build()
    r = subprocess.run([MERGE] + args, capture_output=True, text=True, timeout=120)
    assert r.returncode == expect, (r.returncode, r.stderr)
    return r


@pytest.fixture(scope="module")
def two_results(oracle_lib, tmp_path_factory):
    O = oracle_lib
    tmp = tmp_path_factory.mktemp("merge")
    genomes = synth.random_genomes(16, 8000, seed=60)
    # related genomes in both halves so that one read hits both databases
    for i in range(8):
        g = bytearray(genomes[i])
        for p in range(0, len(g), 97):
            g[p] = ord("A") if g[p] != ord("A") else ord("C")
        genomes[8 + i] = bytes(g)
    reads = synth.sample_reads(genomes, 500, 150, sub_rate=0.01, seed=61, frac_random=0.2)
    ids = [f"q{i}" for i in range(len(reads))]
    res = []
    for h, (lo, hi) in enumerate([(0, 8), (8, 16)]):
        db_dir = synth.make_db(tmp / f"db{h}", genomes[lo:hi], k=21, n_chunks=2, overlap=150, threads=2, names=[f"g{i:05d}" for i in range(lo, hi)])
        odb = O.OracleDB(db_dir)
        rows, trailer = oracle_tsv(O, odb, ids, reads, params=O.default_params(min_qcov=0.4), keep_unmatched=(h == 1))
        odb.close()
        res.append((rows, trailer))
    return tmp, res, len(reads)


def test_merge_two_databases(two_results):
    tmp, res, n = two_results
    a, b = str(tmp / "a.tsv"), str(tmp / "b.tsv.gz")
    write_tsv(a, *res[0])
    write_tsv(b, *res[1])
    both = sum(1 for q in {r.split("\t")[14] for r in res[0][0]} & {r.split("\t")[14] for r in res[1][0] if r.split("\t")[5]})
    assert both > 50  # the case of interest: rows of one query from both inputs
    for sort_by in ("qcov", "tcov", "jacc"):
        out = str(tmp / f"m_{sort_by}.tsv.gz")
        run(["-o", out, "-s", sort_by, a, b])
        got = gzip.open(out, "rt").read().split("\n")
        want, trailer = merge_restated([res[0][0], res[1][0]], [n, n], sort_by)
        assert got[0] == HEADER and got[-1] == "" and got[-4:-1] == trailer
        assert got[1:-4] == want
    # -H, stdout, an input list file
    lst = tmp / "list.txt"
    lst.write_text(f"{a}\n{b}\n")
    r = run(["-H", "-i", str(lst)])
    want, trailer = merge_restated([res[0][0], res[1][0]], [n, n])
    assert r.stdout.split("\n") == want + trailer + [""]
    # every merged row group is sorted and carries its own size in `hits`
    groups = {}
    for line in want:
        f = line.split("\t")
        groups.setdefault(f[14], []).append(f)
    for g in groups.values():
        assert all(f[4] == str(len(g)) for f in g)
        sc = [float(f[11]) for f in g]
        assert sc == sorted(sc, reverse=True)


def test_merge_copy_through_and_errors(two_results):
    tmp, res, n = two_results
    a = str(tmp / "c_a.tsv")
    write_tsv(a, *res[0])
    out = str(tmp / "copy.tsv")
    r = run(["-o", out, a])
    assert "only one file given" in r.stderr and open(out).read() == open(a).read()
    assert "duplicated file" in run([a, a], expect=255).stderr
    assert "should not be the same" in run(["-o", a, a, str(tmp / "b.tsv.gz")], expect=255).stderr
    assert "invalid value for flag -s/--sort-by" in run(["-s", "x", a, str(tmp / "b.tsv.gz")], expect=255).stderr
    # different numbers of input queries
    c = str(tmp / "c_c.tsv")
    write_tsv(c, res[1][0], [f"# input queries: {n + 1}", "# matched queries: 1", "# matched percentage: 1%"])
    assert "different numbers of queries" in run([a, c], expect=255).stderr
    # a trailer-less input (old kmcp versions) is accepted, the other one's count is used
    d = str(tmp / "c_d.tsv")
    write_tsv(d, res[1][0], [], header=False)
    r = run([a, d])
    assert f"# input queries: {n}" in r.stdout.split("\n")
    # query IDs of one queryIdx must agree
    rows = list(res[1][0])
    f = rows[0].split("\t")
    shared = next(x for x in res[0][0] if x.split("\t")[14] == f[14]) if any(x.split("\t")[14] == f[14] for x in res[0][0]) else None
    if shared is None:
        f[14] = res[0][0][0].split("\t")[14]
    f[0] = "someone_else"
    e = str(tmp / "c_e.tsv")
    write_tsv(e, ["\t".join(f)], res[1][1])
    assert "unmatched sequence Ids detected" in run([a, e], expect=255).stderr
    # too few fields / bad numbers
    g = str(tmp / "c_g.tsv")
    write_tsv(g, ["a\tb\tc"], [])
    assert "number of fields (3) < query index field (15)" in run([a, g], expect=255).stderr
    # no rows anywhere: the reference's closing `matched++` still runs
    z1, z2 = str(tmp / "z1.tsv"), str(tmp / "z2.tsv")
    write_tsv(z1, [], ["# input queries: 7", "# matched queries: 0", "# matched percentage: 0.0000%"])
    write_tsv(z2, [], ["# input queries: 7", "# matched queries: 0", "# matched percentage: 0.0000%"])
    assert run([z1, z2]).stdout.split("\n")[1:] == ["# input queries: 7", "# matched queries: 1", "# matched percentage: 14.2857%", ""]


def test_merge_three_inputs_with_ties(tmp_path):
    """Hand-made rows: three inputs, interleaved queryIdx, equal printed scores -> input order, then row order."""
    def row(q, idx, target, qcov, hits=9):
        return f"{q}\t150\t130\t1.0000e-05\t{hits}\t{target}\t0\t1\t1000\t21\t80\t{qcov}\t0.1000\t0.0900\t{idx}"
    f1 = [row("r0", 0, "A", "0.9000"), row("r0", 0, "B", "0.7000"), row("r5", 5, "A", "0.6000")]
    f2 = [row("r0", 0, "C", "0.7000"), row("r2", 2, "C", "0.8000")]
    f3 = [row("r2", 2, "D", "0.8000"), row("r2", 2, "E", "0.9500"), row("r5", 5, "F", "0.6000"), row("r9", 9, "F", "1.0000")]
    paths = []
    for i, rows in enumerate((f1, f2, f3)):
        p = str(tmp_path / f"f{i}.tsv")
        write_tsv(p, rows, ["# input queries: 10", "# matched queries: 2", "# matched percentage: 20.0000%"])
        paths.append(p)
    got = run(paths).stdout.split("\n")
    want = [row("r0", 0, "A", "0.9000", 3), row("r0", 0, "B", "0.7000", 3), row("r0", 0, "C", "0.7000", 3),
            row("r2", 2, "E", "0.9500", 3), row("r2", 2, "C", "0.8000", 3), row("r2", 2, "D", "0.8000", 3),
            row("r5", 5, "A", "0.6000", 2), row("r5", 5, "F", "0.6000", 2), row("r9", 9, "F", "1.0000", 1)]
    assert got == [HEADER] + want + ["# input queries: 10", "# matched queries: 4", "# matched percentage: 40.0000%", ""]
    assert want == merge_restated([f1, f2, f3], [10, 10, 10])[0]


def test_merge_random_inputs_any_block_size_gz_or_not(tmp_path):
    """Random inputs (runs of 1-40 rows per query, ties, CRLF, a last line without newline) against the restatement above, read in
    blocks of 16 bytes ... 4 MB (lines and runs of one queryIdx across block boundaries: the inputs are parsed in place, block by
    block, on threads of their own), plain and gzip in, plain and multi-member gzip out."""
    import gzip
    rng = np.random.default_rng(3)

    def row(q, idx, target, score, hits=9):
        return f"{q}\t150\t130\t1.0000e-05\t{hits}\t{target}\t0\t1\t1000\t21\t80\t{score}\t0.1000\t0.0900\t{idx}"
    for it in range(12):
        n_in = int(rng.integers(2, 5))
        inputs, paths = [], []
        for k in range(n_in):
            rows = []
            for idx in sorted(rng.choice(400, int(rng.integers(0, 120)), replace=False)):
                for j in range(int(rng.integers(1, 41)) if rng.random() < 0.1 else int(rng.integers(1, 4))):
                    rows.append(row(f"read_{idx}/1", int(idx), f"T{k}_{j}", "%.4f" % (rng.integers(0, 11) / 10.0)))
            inputs.append(rows)
            p = str(tmp_path / f"i{it}_{k}.tsv")
            text = "\n".join([HEADER] + rows + ["# input queries: 400", "# matched queries: 1", "# matched percentage: 0.2500%"])
            if rng.random() < 0.5:
                text += "\n"
            if rng.random() < 0.3:
                text = text.replace("\n", "\r\n")
            if rng.random() < 0.5:
                p += ".gz"
                with gzip.open(p, "wb") as fh:
                    fh.write(text.encode())
            else:
                open(p, "wb").write(text.encode())
            paths.append(p)
        want_rows, trailer = merge_restated(inputs, [400] * n_in)
        out = str(tmp_path / f"o{it}.tsv") + (".gz" if it % 2 else "")
        env = dict(os.environ, KMCP_MERGE_BLOCK=str(int(rng.choice([16, 33, 100, 1000, 65536, 4 << 20]))))
        r = subprocess.run([MERGE, "-o", out] + paths, capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr
        got = (gzip.open(out, "rt") if out.endswith(".gz") else open(out)).read().split("\n")
        assert got[0] == HEADER and got[1:1 + len(want_rows)] == want_rows, (it, env["KMCP_MERGE_BLOCK"])
        assert got[1 + len(want_rows):] == trailer + [""]


def test_merge_equals_search_of_the_joint_database(oracle_lib, tmp_path):
    """An expectation that does not come from reading merge.go: a database whose index files are the files of database A
    followed by the files of database B answers every query with A's rows and B's rows together (blocks are independent:
    util-db-search.go:939-964 concatenates their replies).  So `kmcp search` on the joint database == `kmcp merge` of the
    searches on A and on B: the same set of rows per query with `hits` = their number, rows in non-increasing printed score,
    the same `# input queries` / `# matched queries` trailer.  (The order among rows whose scores print alike may differ:
    search sorts on the exact value, merge on the four printed decimals — merge.go re-parses the column.)"""
    import shutil
    O = oracle_lib
    genomes = synth.random_genomes(18, 9000, seed=160)
    for i in range(9):  # relatives in the other half: queries with rows from both databases
        g = bytearray(genomes[i])
        for p in range(0, len(g), 83):
            g[p] = ord("A") if g[p] != ord("A") else ord("C")
        genomes[9 + i] = bytes(g)
    reads = synth.sample_reads(genomes, 700, 150, sub_rate=0.01, seed=161, frac_random=0.15)
    ids = [f"q{i}" for i in range(len(reads))]
    dirs = [synth.make_db(tmp_path / f"db{h}", genomes[lo:hi], k=21, n_chunks=2, overlap=150, threads=3, names=[f"g{i:05d}" for i in range(lo, hi)])
            for h, (lo, hi) in enumerate([(0, 9), (9, 18)])]
    joint = tmp_path / "joint" / "R001"
    os.makedirs(joint)
    files = []
    for d in dirs:
        for f in sorted(x for x in os.listdir(d) if x.endswith(".uniki")):
            name = "_block%03d.uniki" % (len(files) + 1)
            shutil.copy(os.path.join(d, f), joint / name)
            files.append(name)
    yml = open(os.path.join(dirs[0], "__db.yml")).read()
    yml = yml[:yml.index("files:")] + "files:\n" + "".join(f"- {f}\n" for f in files)
    (joint / "__db.yml").write_text(yml)
    p = O.default_params(min_qcov=0.4)
    tsvs = []
    for d in dirs + [str(joint)]:
        odb = O.OracleDB(d)
        tsvs.append(oracle_tsv(O, odb, ids, reads, params=p))
        odb.close()
    assert len(tsvs[2][0]) == len(tsvs[0][0]) + len(tsvs[1][0]) > 900
    a, b = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    write_tsv(a, *tsvs[0])
    write_tsv(b, *tsvs[1])
    for sort_by, col, key in (("qcov", 11, 0), ("tcov", 12, 1), ("jacc", 13, 2)):
        ps = O.default_params(min_qcov=0.4, sort_by=key)
        odb = O.OracleDB(str(joint))
        want_rows, want_trailer = oracle_tsv(O, odb, ids, reads, params=ps)
        odb.close()
        got = run(["-s", sort_by, a, b]).stdout.split("\n")
        assert got[0] == HEADER and got[-1] == "" and got[-4:-1] == want_trailer

        def groups(rows):
            g = {}
            for r in rows:
                g.setdefault(r.split("\t")[14], []).append(r)
            return g
        gg, gw = groups(got[1:-4]), groups(want_rows)
        assert list(gg) == list(gw)  # queries in input order
        both = 0
        for q in gw:
            assert sorted(gg[q]) == sorted(gw[q]), q  # the same rows, `hits` rewritten to the joint count
            sc = [float(r.split("\t")[col]) for r in gg[q]]
            assert sc == sorted(sc, reverse=True), q
            both += len({r.split("\t")[5][:6] < "g00009" for r in gw[q]}) == 2
        assert both > 50
