"""kmcp-search's gzip decoder (cli/fast_gunzip.hpp) against zlib: tests/gunzip_check.cpp — generated streams of every level and
strategy, several members, header fields, trailing bytes; damaged and truncated copies must be rejected exactly when zlib rejects
them — under ASan/UBSan.  The reader tests (tests/test_formats_cpu.py) run the decoder inside the CLI."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_gunzip_equals_zlib(tmp_path):
    exe = str(tmp_path / "gunzip_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "gunzip_check.cpp"), "-lz"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "identical to zlib" in r.stdout, r.stdout
