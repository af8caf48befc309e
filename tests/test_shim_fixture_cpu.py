"""shim/testdata — the fixture of the Go-side test of the cgo binding (shim/kmcp_gpu_test.go, run by shim/build.sh where a Go
toolchain exists) — is what tests/golden/make_shim_fixture.py makes today: database files byte for byte, the reads, and the TSV
the CPU oracle prints for them.  (tests/test_gpu_cli.py::test_shim_fixture_through_kmcp_search holds kmcp-search to the same TSV.)"""
import filecmp
import os

from tests.golden import make_shim_fixture as F
from tests.test_gpu_cli import write_fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "shim", "testdata")


def test_fixture_is_what_the_oracle_prints(oracle_lib, tmp_path):
    db_dir, ids, reads, lines = F.fixture(str(tmp_path))
    assert open(os.path.join(FIX, "expected.tsv")).read() == "\n".join(lines) + "\n"
    write_fastq(str(tmp_path / "reads.fq"), ids, reads)
    assert filecmp.cmp(str(tmp_path / "reads.fq"), os.path.join(FIX, "reads.fq"), shallow=False)
    want = sorted(os.listdir(os.path.join(FIX, "db", "R001")))
    assert want == sorted(os.listdir(db_dir)) and len([f for f in want if f.endswith(".uniki")]) == 2
    for f in want:
        assert filecmp.cmp(os.path.join(db_dir, f), os.path.join(FIX, "db", "R001", f), shallow=False), f
    rows = [ln for ln in lines if not ln.startswith("#")]
    assert len(rows) > 40 and len({r.split("\t")[0] for r in rows}) > 30  # most reads match; some match two references


def test_shim_sources_name_every_entry_point_they_call():
    """No Go here: at least every C symbol the binding calls must be one the header declares and the library exports."""
    import re
    from kmcp_amd import lib
    src = open(os.path.join(ROOT, "shim", "kmcp_gpu.go")).read()
    called = set(re.findall(r"C\.(kmcpg_[a-z_]+)\(", src))
    hdr = open(os.path.join(ROOT, "include", "kmcp_gpu.h")).read()
    assert called and all(re.search(r"\b%s\(" % c, hdr) for c in called), called
    assert called <= set(lib.EXPORTS) | {"kmcpg_last_error"}
    test = open(os.path.join(ROOT, "shim", "kmcp_gpu_test.go")).read()
    for fn in ("OpenGPUDB", "SearchBatch", "NewGPUSearchEngine"):
        assert f"func {fn}(" in src or f") {fn}(" in src
        assert fn in test
