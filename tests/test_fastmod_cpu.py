"""The row address `h % NumSigs` (util-db-search.go:6611, :6811): the one-multiply exact form of kmcp_amd/csrc/fastmod.hpp,
compiled for the host, against the % operator (tests/fastmod_check.cpp: ~20 M operand pairs)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fastmod_equals_the_remainder(tmp_path):
    exe = str(tmp_path / "fastmod_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "fastmod_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 wrong" in r.stdout, r.stdout
