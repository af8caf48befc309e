"""kmcpg_build_db (`kmcp index` on the GPU from k-mer hash lists) vs the oracle's restatement of index.go: the `.uniki` block
files must be byte-identical (header, row-major Bloom matrix, bit order), `__db.yml` must carry the same values, and a search
over the GPU-built database must equal the oracle's search over its own."""
import filecmp
import os

import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _yml(path):
    out, key = {}, None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("- "):
            out.setdefault(key, []).append(line[2:])
        elif ":" in line:
            key, v = line.split(":", 1)
            out[key] = v.strip() if v.strip() else []
    return out


@pytest.mark.parametrize("case", [
    dict(k=21, n=60, glen=4000, chunks=2, nh=1, fpr=0.3, threads=4, kw={}),                       # several 3-byte-row blocks
    dict(k=31, n=9, glen=60000, chunks=1, nh=3, fpr=0.01, threads=8, kw=dict(scale=10)),           # the demo-searching shape
    dict(k=21, n=1203, glen=700, chunks=1, nh=2, fpr=0.1, threads=1, kw=dict(syncmer_s=11)),       # one 151-byte-row block, ragged
    dict(k=25, n=40, glen=3000, chunks=3, nh=4, fpr=0.05, threads=2, kw=dict(minimizer_w=8), block_size=16),
])
def test_gpu_built_database_is_byte_identical(oracle_lib, tmp_path, case):
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    genomes = synth.random_genomes(case["n"], case["glen"], seed=77)
    cfg = O.sketch_cfg(k=case["k"], **case["kw"])
    cols = synth.make_columns(genomes, cfg, n_chunks=case["chunks"], overlap=100)
    bs = case.get("block_size", 0)
    ref = O.build_db(str(tmp_path / "oracle"), cfg, cols, num_hashes=case["nh"], fpr=case["fpr"], threads=case["threads"], block_size=bs)
    got = lib.build_db(str(tmp_path / "gpu"), cols, k=case["k"], num_hashes=case["nh"], fpr=case["fpr"], threads=case["threads"], block_size=bs,
                       scale=case["kw"].get("scale", 1), minimizer_w=case["kw"].get("minimizer_w", 0), syncmer_s=case["kw"].get("syncmer_s", 0))
    yr, yg = _yml(os.path.join(ref, "__db.yml")), _yml(os.path.join(got, "__db.yml"))
    assert yr["files"] == yg["files"] and len(yr["files"]) >= 1
    for key in ("version", "unikiVersion", "k", "ks", "hashed", "canonical", "scaled", "scale", "minimizer", "minimizer-w", "syncmer", "syncmer-s",
                "hashes", "numNameGroups", "blocksize", "totalKmers"):
        assert yr[key] == yg[key], key
    assert float(yr["fpr"]) == float(yg["fpr"])
    for f in yr["files"]:
        assert filecmp.cmp(os.path.join(ref, f), os.path.join(got, f), shallow=False), f
    # and it searches like the oracle's
    reads = synth.sample_reads(genomes, 200, min(300, case["glen"] - 1), sub_rate=0.01, seed=78, frac_random=0.1)
    t = max(0.55, case["fpr"] + 0.1)
    odb = O.OracleDB(ref)
    try:
        with Database.open(got, device=0) as db:
            res = db.search(reads, params=default_params(min_qcov=t))
        assert synth.assert_parity(odb, res, reads, None, O.default_params(min_qcov=t)) > 50
    finally:
        odb.close()


@pytest.mark.parametrize("size_x", [16, 256])
def test_big_genome_block_rules(oracle_lib, tmp_path, size_x):
    """index.go:787-894 with thresholds scaled down (-x 1500 -8 3000 -1 6000 k-mers): genomes of four size classes end up in
    blocks of -b, -X, 8 and 1 columns; 1-column blocks (1-byte rows) are searched like any other."""
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    base = synth.random_genomes(75, 9000, seed=81)
    lens = [1000] * 40 + [2200] * 20 + [4500] * 11 + [8000] * 4
    genomes = [g[:n] for g, n in zip(base, lens)]
    cfg = O.sketch_cfg(k=21)
    cols = synth.make_columns(genomes, cfg)
    rules = dict(kmers_x=1500, block_size_x=size_x, kmers_8=3000, kmers_1=6000)
    ref = O.build_db(str(tmp_path / "oracle"), cfg, cols, threads=2, block_size=32, rules=O.BlockRules(**rules))
    got = lib.build_db(str(tmp_path / "gpu"), cols, k=21, threads=2, block_size=32, **rules)
    yr, yg = _yml(os.path.join(ref, "__db.yml")), _yml(os.path.join(got, "__db.yml"))
    # -X 16: 40 -> 32+8 | 20 -> 16+4 | 11 -> 8+3 | 4 x 1 = 10 blocks;  -X 256 >= -b 32: 60 -> 32+28 | 11 -> 11 | 4 x 1 = 7 blocks
    assert yr["files"] == yg["files"] and len(yr["files"]) == (10 if size_x == 16 else 7)
    assert yr["blocksize"] == yg["blocksize"] == "32"
    for f in yr["files"]:
        assert filecmp.cmp(os.path.join(ref, f), os.path.join(got, f), shallow=False), f
    reads = synth.sample_reads(genomes, 300, 150, sub_rate=0.01, seed=82, frac_random=0.1)
    odb = O.OracleDB(ref)
    try:
        with Database.open(got, device=0) as db:
            assert db.info.n_blocks == len(yr["files"])
            res = db.search(reads, params=default_params())
        assert synth.assert_parity(odb, res, reads, None, O.default_params()) > 100
    finally:
        odb.close()
    with pytest.raises(lib.KmcpGpuError):
        lib.build_db(str(tmp_path / "bad"), cols, kmers_x=5000, kmers_8=3000, kmers_1=6000)


def test_build_refuses_bad_input(tmp_path):
    import numpy as np
    from kmcp_amd import lib
    with pytest.raises(lib.KmcpGpuError):
        lib.build_db(str(tmp_path / "x"), [("a", 10, 0, 1, np.arange(5, dtype=np.uint64))], num_hashes=7)
    with pytest.raises(lib.KmcpGpuError):
        lib.build_db(str(tmp_path / "y"), [("a", 10, 0, 1, np.arange(5, dtype=np.uint64))], fpr=1.5)


@pytest.mark.parametrize("mode", [1, 2])
def test_uniform_num_sigs_makes_blocks_groupable(oracle_lib, tmp_path, mode):
    """kmcpg_build_cfg.uniform_sigs: chunks of varied length give every block of a `-j 32`-style index its own
    NumSigs (columns are sorted by k-mer count before they are cut into blocks), so nothing can share a gather; with the option the
    blocks of a size tier agree on NumSigs (1: all of them, 2: a 5/4 ladder), the files stay ordinary `.uniki` blocks — the oracle's
    reader (index/serialization.go:383-593 restated) takes them and finds what the GPU finds — and the resident layout puts them side
    by side: one wide row instead of many narrow ones."""
    import numpy as np
    from kmcp_amd import Database, default_params, lib
    O = oracle_lib
    rng = np.random.default_rng(5)
    lens = rng.integers(3000, 9000, size=320)  # a 3x spread of chunk sizes, like genomes of different species
    genomes = [g[:n] for g, n in zip(synth.random_genomes(320, 9000, seed=83), lens)]
    cfg = O.sketch_cfg(k=21)
    cols = synth.make_columns(genomes, cfg)
    plain = lib.build_db(str(tmp_path / "plain"), cols, k=21, threads=8)          # 8 blocks of 40 columns (5-byte rows)
    uni = lib.build_db(str(tmp_path / "uni"), cols, k=21, threads=8, uniform_sigs=mode)
    reads = synth.sample_reads(genomes, 600, 150, sub_rate=0.01, seed=84, frac_random=0.1)
    with Database.open(plain, device=0) as dp, Database.open(uni, device=0) as du:
        nb = dp.info.n_blocks
        assert nb == du.info.n_blocks == 8
        sig_p = [dp.block_info(b)["num_sigs"] for b in range(nb)]
        sig_u = [du.block_info(b)["num_sigs"] for b in range(nb)]
        assert len(set(sig_p)) == nb                                 # the reference's sizing: nothing to group
        assert all(u >= p_ for u, p_ in zip(sig_u, sig_p))           # only ever larger filters
        assert max(sig_u) == max(sig_p)
        if mode == 1:
            assert len(set(sig_u)) == 1
            assert all(du.block_info(b)["stride"] == 64 for b in range(nb))   # 8 x 5 bytes side by side in one 64-byte row
            assert all(dp.block_info(b)["stride"] == 16 for b in range(nb))
        else:
            assert len(set(sig_u)) < nb and du.info.matrix_bytes < 1.25 * dp.info.matrix_bytes
        res_u = du.search(reads, params=default_params())
        res_p = dp.search(reads, params=default_params())
    odb = O.OracleDB(uni)
    try:
        assert [odb.block_info(b)[0] for b in range(odb.nblocks)] == sig_u
        assert synth.assert_parity(odb, res_u, reads, None, O.default_params()) > 300
    finally:
        odb.close()
    # the strong matches (reads sampled from a genome) are found in both databases: only chance k-mers differ with NumSigs
    strong = 0
    for i in range(len(reads)):
        cols_u = {int(m["col"]) for m in res_u.read(i)}
        cols_p = {int(m["col"]) for m in res_p.read(i)}
        for m in res_p.read(i):
            if m["qcov"] >= 0.9:
                assert int(m["col"]) in cols_u, (i, int(m["col"]))
                strong += 1
        for m in res_u.read(i):
            if m["qcov"] >= 0.9:
                assert int(m["col"]) in cols_p, (i, int(m["col"]))
    assert strong > 200
    # kmcp-search says once at open when narrow blocks cannot share a gather, and what to do about it
    import subprocess
    from tests.test_gpu_cli import CLI, write_fastq
    fq = str(tmp_path / "r.fq")
    write_fastq(fq, [f"r{i}" for i in range(20)], reads[:20])
    for db_dir, hint in ((plain, True), (uni, mode != 1)):
        r = subprocess.run([CLI, "-d", os.path.dirname(db_dir), fq, "-o", str(tmp_path / "o.tsv")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert ("distinct NumSigs over 8 narrow blocks" in r.stderr) == hint, r.stderr
