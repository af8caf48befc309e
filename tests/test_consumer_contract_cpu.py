"""The drop-in spelling and the consumer contract without a GPU: `kmcp merge ...` through the dispatcher (cli/kmcp_dispatch.cpp)
and kmcp-merge's output read by the rules `kmcp profile` applies to its input (tests/profile_contract.py restates
util-profile.go:94-182, profile.go:1939-1962).  The search side of the same contract needs the GPU: tests/test_gpu_cli.py."""
import os
import subprocess

import pytest

from tests import profile_contract as PC
from tests.test_gpu_cli import HEADER
from tests.test_merge_cpu import two_results, write_tsv  # noqa: F401  (fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMCP = os.path.join(ROOT, "kmcp_amd", "kmcp")


def _built():
    if not os.path.exists(KMCP):
        import __graft_entry__ as g
        g.build()


def test_contract_reader_rules():
    row = "q1\t150\t130\t7.4626e-15\t1\tGCF_1\t9\t10\t4857450\t21\t90\t0.6923\t0.0002\t0.0002\t1"
    m = PC.parse_match_result(row)
    assert m == {"query": "q1", "qlen": 150, "qkmers": 130, "fpr": 7.4626e-15, "hits": 1, "target": "GCF_1", "chunk_idx": 9, "chunks": 10,
                 "gsize": 4857450, "k": 21, "mkmers": 90, "qcov": 0.6923}
    assert PC.parse_match_result(row, min_qcov=0.7) is None and PC.parse_match_result(row, max_fpr=1e-20) is None
    with pytest.raises(PC.ContractError):
        PC.parse_match_result("\t".join(row.split("\t")[:12]))  # 12 fields: "invalid kmcp search result format"
    with pytest.raises(PC.ContractError):
        PC.parse_match_result(row.replace("\t150\t", "\t150.0\t"))  # Atoi takes no decimal point
    ms, total, stats = PC.read_search_result(HEADER + "\n" + row + "\n\n# input queries: 7\n# matched queries: 1\n# matched percentage: 14.2857%\n")
    assert len(ms) == 1 and total == 7 and stats["matched queries"] == "1"


def test_kmcp_merge_spelling_and_profile_contract(two_results, tmp_path):  # noqa: F811
    _built()
    tmp, res, n = two_results
    a, b = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv.gz")
    write_tsv(a, *res[0])
    write_tsv(b, *res[1])
    out = {}
    for name, argv in (("dispatch", [KMCP, "merge"]), ("dispatch_root_flags", [KMCP, "-q", "-j", "4", "merge"]),
                       ("subcommand_word", [os.path.join(ROOT, "kmcp_amd", "kmcp-merge"), "merge"]), ("plain", [os.path.join(ROOT, "kmcp_amd", "kmcp-merge")])):
        r = subprocess.run(argv + ["-s", "qcov", a, b], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.stderr)
        out[name] = r.stdout
    assert out["dispatch"] == out["plain"] == out["subcommand_word"] == out["dispatch_root_flags"]
    # the merged result as `kmcp profile` reads it (defaults of profile: --max-fpr 0.05, --min-query-cov 0.55)
    ms, total, stats = PC.read_search_result(out["plain"])
    assert total == n and set(stats) == {"input queries", "matched queries", "matched percentage"}
    rows = [ln for ln in out["plain"].split("\n") if ln and ln[0] != "#"]
    kept = [ln for ln in rows if ln.split("\t")[5] and float(ln.split("\t")[11]) >= 0.55 and float(ln.split("\t")[3]) <= 0.05]
    assert len(ms) == len(kept) > 100
    for m, ln in zip(ms, kept):
        f = ln.split("\t")
        assert (m["query"], m["target"], str(m["qlen"]), str(m["qkmers"]), str(m["hits"]), str(m["chunk_idx"]), str(m["chunks"]), str(m["gsize"]), str(m["k"]),
                str(m["mkmers"])) == (f[0], f[5], f[1], f[2], f[4], f[6], f[7], f[8], f[9], f[10])
        assert "%.4f" % m["qcov"] == f[11] and "%.4e" % m["fpr"] == f[3]
    # rows kept with -K (unmatched queries: empty target, qCov 0) never pass profile's qCov filter
    assert any(ln.split("\t")[5] == "" for ln in rows) and all(m["target"] for m in ms)
    # other commands go to the reference binary, or fail like checkError when there is none
    env = dict(os.environ, PATH="/nonexistent", KMCP_REFERENCE_BIN="")
    env.pop("KMCP_REFERENCE_BIN")
    r = subprocess.run([KMCP, "profile", "x.tsv"], capture_output=True, text=True, env=env)
    assert r.returncode == 255 and "not part of this build" in r.stderr
    fake = tmp_path / "refbin"
    fake.write_text("#!/bin/sh\necho reference-kmcp \"$@\"\n")
    fake.chmod(0o755)
    r = subprocess.run([KMCP, "-j", "8", "profile", "-X", "taxdump", "x.tsv"], capture_output=True, text=True, env=dict(env, KMCP_REFERENCE_BIN=str(fake)))
    assert r.returncode == 0 and r.stdout.strip() == "reference-kmcp -j 8 profile -X taxdump x.tsv"
