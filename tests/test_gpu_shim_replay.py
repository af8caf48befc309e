"""The cgo shim (shim/kmcp_gpu.go) cannot be compiled here (no Go toolchain), so the exact sequence of C calls it makes is
replayed by a plain-C program (tests/shim_replay.c): malloc'd CSR buffers freed right after kmcpg_submit, tickets waited for
on other threads and out of order, at most three in flight, names cached at open, the error message fetched on the failing
thread.  Its per-batch checksums must equal the ones computed from the Python binding's results on the same batches."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK = (1 << 64) - 1


def _checksum(db, res):
    s = 0
    for i in range(len(res)):
        s = (s * 1099511628211 + int(res.qlen[i]) * 31 + int(res.qkmers[i]) * 7 + int(res.ksize[i])) & MASK
        for m in res.read(i):
            qc = struct.unpack("<Q", struct.pack("<d", float(m["qcov"])))[0]
            s = (s * 1099511628211 + (int(m["col"]) << 20) + int(m["mkmers"]) + qc + len(db.col_info(int(m["col"]))[0])) & MASK
    return s


@pytest.mark.parametrize("asan", [False, True])
def test_shim_call_sequence_replayed_in_c(oracle_lib, tmp_path, asan):
    from kmcp_amd import Database, default_params, lib
    exe = str(tmp_path / "shim_replay")
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "shim_replay.c"), "-o", exe,
           "-L", os.path.join(ROOT, "kmcp_amd"), "-lkmcpgpu", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "kmcp_amd"), "-Wl,-rpath-link,/opt/rocm/lib"]
    if asan:
        cmd[1:1] = ["-fsanitize=address", "-fno-omit-frame-pointer"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    genomes = synth.random_genomes(16, 15000, seed=61)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, n_chunks=2, threads=4)
    batches = [synth.sample_reads(genomes, 400 + 91 * i, 150, sub_rate=0.01, seed=300 + i, frac_random=0.15) + ([b"", b"ACGT"] if i % 2 else []) for i in range(9)]
    with open(tmp_path / "batches.bin", "wb") as fh:
        fh.write(struct.pack("<I", len(batches)))
        for reads in batches:
            seqs, offs = lib.pack_reads(reads)
            fh.write(struct.pack("<I", len(reads)) + offs.tobytes() + seqs.tobytes())
    env = dict(os.environ)
    if asan:
        # the HIP runtime keeps process-lifetime allocations; leaks are not what this run is about, bad accesses are
        env["ASAN_OPTIONS"] = "detect_leaks=0:protect_shadow_gap=0"
    r = subprocess.run([exe, db_dir, str(tmp_path / "batches.bin")], capture_output=True, text=True, timeout=600, env=env)
    if asan and r.returncode != 0 and "AddressSanitizer" not in r.stderr and ("hip" in r.stderr.lower() or "hsa" in r.stderr.lower()):
        pytest.skip("the HIP runtime does not start under ASan on this box: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "error-path ok" and len(lines) == 1 + len(batches)
    with Database.open(db_dir) as db:
        for i, reads in enumerate(batches):
            res = db.search(reads, params=default_params(fpr_buf_size=249))
            want = f"batch {i} reads {len(reads)} matches {int(res.offs[-1])} sum {_checksum(db, res):016x}"
            assert lines[1 + i] == want
            assert int(res.offs[-1]) > 200
