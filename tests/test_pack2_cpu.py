"""The 2-bit packed upload of long-query batches (kmcp_amd/csrc/pack2.hpp, used by host.cpp's staging): pack + the device's unpack
(restated on the host) reproduce the caller's bases up to spelling (a -> A, u/U -> T: the same ntHash seeds) and every other byte
verbatim; scalar, AVX2 and multi-threaded paths; garbage input is refused rather than sent as millions of exception runs
(tests/pack2_check.cpp).  The reference has no counterpart: it hashes the bytes it read (util-db-search.go:1037-1107)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack2_round_trip(tmp_path):
    exe = str(tmp_path / "pack2_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "pack2_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
