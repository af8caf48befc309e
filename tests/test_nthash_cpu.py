"""The rolling ntHash of k1_seg_roll compiled for the host (kmcp_amd/csrc/nthash.hpp: seeds, rotations, start-up and rolling step with
the kernel's rotated seed tables) against the oracle's k-mer hashes (oracle/kmcp_oracle.c, itself pinned by the reference's golden
tables): every k the kernel takes (<= 128), upper / lower case, N and IUPAC bases, FracMinHash scales."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nt(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("nt") / "nthash_check.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "nthash_check.cpp")], check=True)
    lib = C.CDLL(so)
    lib.nt_roll_all.restype = C.c_long
    lib.nt_roll_all.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int, C.c_ulonglong, C.c_void_p]
    return lib


def test_rolling_hashes_equal_the_oracle(nt, oracle_lib):
    O = oracle_lib
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNRYKMnU", dtype=np.uint8)
    n_kmers = 0
    for k in (1, 2, 7, 15, 21, 31, 32, 33, 63, 64, 65, 127, 128):
        for scale in (1, 1, 3, 200):
            length = int(rng.integers(k, 6000))
            seq = alphabet[rng.integers(0, len(alphabet) if scale == 1 else 16, size=length)].tobytes()
            cfg = O.sketch_cfg(k=k, scale=scale)
            want = O.generate_kmers(seq, cfg)
            out = np.zeros(length + 1, dtype=np.uint64)
            max_hash = (2**64 - 1) // scale if scale > 1 else 0
            n = nt.nt_roll_all(seq, length, k, int(scale > 1), max_hash, out.ctypes.data)
            assert n == len(want), (k, scale, n, len(want))
            assert np.array_equal(out[:n], want), (k, scale)
            n_kmers += n
        # shorter than k: nothing
        assert nt.nt_roll_all(b"ACGT" * 40, k - 1, k, 0, 0, np.zeros(4, np.uint64).ctypes.data) == 0
    assert n_kmers > 50000


def test_two_bit_rolling_form_equals_the_oracle(nt, oracle_lib):
    """k1_seg_roll2's arithmetic on the host (round 5): bytes folded to 2-bit codes four at a time, pair tables F2 / R2, table indices
    from nibble words (spread + funnel shift by k), funnel-shift rotations - against the oracle's k-mer hashes for every k the kernel
    takes, upper and lower case; any other byte (N, IUPAC, U) must be reported (-1: the kernel hands the segment to the byte kernel)."""
    O = oracle_lib
    nt.nt_roll2_all.restype = __import__("ctypes").c_long
    nt.nt_roll2_all.argtypes = nt.nt_roll_all.argtypes
    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    n_kmers = 0
    for k in (1, 2, 7, 15, 16, 17, 21, 31, 32, 33, 47, 48, 63, 64, 65, 96, 127, 128):
        for scale in (1, 3, 200):
            for length in (k, k + 1, k + 15, k + 16, int(rng.integers(k, 5000)), 4096 + k):
                seq = acgt[rng.integers(0, 8 if length % 2 else 4, size=length)].tobytes()
                cfg = O.sketch_cfg(k=k, scale=scale)
                want = O.generate_kmers(seq, cfg)
                out = np.zeros(length + 1, dtype=np.uint64)
                max_hash = (2**64 - 1) // scale if scale > 1 else 0
                n = nt.nt_roll2_all(seq, length, k, int(scale > 1), max_hash, out.ctypes.data)
                assert n == len(want), (k, scale, length, n, len(want))
                assert np.array_equal(out[:n], want), (k, scale, length)
                n_kmers += n
    assert n_kmers > 100000
    # every byte that is not A / C / G / T in either case is refused, wherever it sits (full groups and the tail)
    base = acgt[rng.integers(0, 4, size=200)].copy()
    out = np.zeros(256, dtype=np.uint64)
    assert nt.nt_roll2_all(base.tobytes(), 200, 21, 0, 0, out.ctypes.data) == 180
    for pos in (0, 3, 15, 16, 100, 191, 192, 199):
        for ch in b"NnUuRYKM-*.\x00\xff@[`{BDEFHIJLOPQSVWXZ1":
            s = base.copy()
            s[pos] = ch
            assert nt.nt_roll2_all(s.tobytes(), 200, 21, 0, 0, out.ctypes.data) == -1, (pos, ch)
