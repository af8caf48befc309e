"""The bit-sliced counters of k2_cobs (kmcp_amd/csrc/csa.hpp: carry-save groups of 8 / 4 rows; the deferred carries of the
16 / 24-plane kernels; the integer count threshold of a query) compiled for the host, against scalar per-column counts and the
reference's float64 test count by count (tests/csa_check.cpp).  The reference counts
the same rows into per-column bytes / uint16 (util-db-search.go:6811-6972)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bit_sliced_counters_equal_scalar_counts(tmp_path):
    exe = str(tmp_path / "csa_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "csa_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 wrong" in r.stdout, r.stdout
