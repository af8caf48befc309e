"""kmcp-search's TSV row formatter (search.go:517-575 rows; FormatFloat 'f'/'e' with 4 digits) without a GPU: the per-query /
per-column fast path against the per-row path, and its fixed-point printer against printf — tests/rowformat_check.cpp under
ASan/UBSan."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rows_equal_row_and_printf(tmp_path):
    exe = str(tmp_path / "rowformat_check")
    lib_dir = os.path.join(ROOT, "kmcp_amd")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "rowformat_check.cpp"), "-L" + lib_dir, "-lkmcpgpu", "-lz", "-lpthread",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "rows identical" in r.stdout, r.stdout


def test_packed_queries_joined_into_a_batch(tmp_path):
    """kmcp-search -g (round 6): a query packed on its own thread and moved behind a batch (Batch::append_packed: shifted copy + re-based
    runs) == the same text packed straight into the batch; tests/packappend_check.cpp under ASan/UBSan."""
    exe = str(tmp_path / "packappend_check")
    lib_dir = os.path.join(ROOT, "kmcp_amd")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "packappend_check.cpp"), "-L" + lib_dir, "-lkmcpgpu", "-lz", "-lpthread",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr[-3000:]
