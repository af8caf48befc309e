"""The big-genome block rules of `kmcp index` (kmcp/cmd/index.go:787-894), restated in the oracle as the reference's state
machine.  Expected layouts below were derived by hand from the reference loop (one trace per case in the comments); the GPU
builder implements the same rules in a different, declarative form (size tiers) and is compared with the oracle byte for
byte in tests/test_gpu_build.py."""
import numpy as np


def sizes(block_of, nb):
    return [int((block_of == b).sum()) for b in range(1, nb + 1)]


def test_tiers_with_x(oracle_lib):
    O = oracle_lib
    r = O.BlockRules(kmers_x=100, block_size_x=16, kmers_8=200, kmers_1=1000)
    km = [50] * 40 + [150] * 20 + [500] * 10 + [2000] * 3
    nb, bo = O.block_layout(km, 32, r)
    # 0-31 fill a block of 32; 32-39 are closed short when column 40 crosses -x (held back, sBlock := 16);
    # 40-55 fill 16; 56-59 closed short when column 60 crosses -8 (sBlock := 8); 60-67 fill 8; 68-69 closed short when
    # column 70 crosses -1; then one column per block
    assert sizes(bo, nb) == [32, 8, 16, 4, 8, 2, 1, 1, 1]
    assert (np.diff(bo) >= 0).all() and bo[0] == 1 and bo[-1] == 9  # blocks are runs of the ascending list


def test_x_skipped_when_not_smaller_than_b(oracle_lib):
    O = oracle_lib
    r = O.BlockRules(kmers_x=100, block_size_x=256, kmers_8=200, kmers_1=1000)
    km = [50] * 5 + [150] * 3 + [500] * 6 + [2000] * 2
    nb, bo = O.block_layout(km, 8, r)
    # -X (256) >= -b (8): the -x threshold is ignored; 0-7 fill a block; column 8 crosses -8: held back, block size stays 8;
    # 8-13 closed short when column 14 crosses -1; 14 and 15 alone
    assert sizes(bo, nb) == [8, 6, 1, 1]


def test_edges(oracle_lib):
    O = oracle_lib
    r = O.BlockRules(kmers_x=100, block_size_x=16, kmers_8=200, kmers_1=1000)
    nb, bo = O.block_layout([0, 0, 5, 5, 5], 8, r)  # empty columns are skipped
    assert nb == 1 and list(bo) == [0, 0, 1, 1, 1]
    nb, bo = O.block_layout([150, 150, 150], 32, r)  # the very first column is already above -x
    assert nb == 1 and list(bo) == [1, 1, 1]
    nb, bo = O.block_layout([2000, 3000], 32, r)
    assert nb == 2 and list(bo) == [1, 2]
    nb, bo = O.block_layout([50] * 8 + [300], 8, O.BlockRules(kmers_x=100, block_size_x=16, kmers_8=200, kmers_1=1000))
    assert sizes(bo, nb) == [8, 1]  # -X 16 >= -b 8: skipped; the lone column above -8 still gets its own (short) block
    nb, bo = O.block_layout([7] * 20, 8, None)  # defaults: nothing is big
    assert sizes(bo, nb) == [8, 8, 4]
    # defaults are 10M / 256 / 20M / 200M with M = 2^20
    km = [10 << 20] * 300 + [(10 << 20) + 1] * 300 + [(20 << 20) + 1] * 9 + [(200 << 20) + 1]
    nb, bo = O.block_layout(km, 296, None)
    assert sizes(bo, nb) == [296, 4, 256, 44, 8, 1, 1]


def test_random_layouts_match_tier_rule(oracle_lib):
    """Property: the state machine == 'cut every size tier into blocks of its own size' (the form the GPU builder uses)."""
    O = oracle_lib
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(1, 200))
        sblock = int(rng.choice([8, 16, 24, 40, 64]))
        size_x = int(rng.choice([16, 24, 32, 256]))
        tx, t8, t1 = sorted(rng.choice(np.arange(1, 400), 3, replace=False).tolist())
        km = np.sort(rng.integers(0, 500, n)).astype(np.uint64)
        nb, bo = O.block_layout(km, sblock, O.BlockRules(kmers_x=tx, block_size_x=size_x, kmers_8=t8, kmers_1=t1))
        skip_x = size_x >= sblock
        tier = lambda v: 3 if v > t1 else 2 if v > t8 else 1 if (not skip_x and v > tx) else 0
        tsize = [sblock, size_x, sblock if skip_x else 8, 1]
        want, b, i = np.zeros(n, dtype=np.int32), 0, 0
        while i < n:
            if km[i] == 0:
                i += 1
                continue
            t, cnt, b = tier(km[i]), 0, b + 1
            while i < n and cnt < tsize[t] and tier(km[i]) == t:
                want[i] = b
                i, cnt = i + 1, cnt + 1
        assert nb == b and np.array_equal(bo, want), (km, sblock, size_x, tx, t8, t1)
