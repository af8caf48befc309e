"""The 2-bit packed upload (host.cpp stage -> pack2.hpp -> k_unpack2): batches of long queries travel as 2-bit codes plus runs of
foreign bytes and must give exactly what the ASCII upload gives — and what the oracle gives — whatever the bytes are: lower case,
U, N runs, IUPAC codes (their seed is 0 but their complement entry depends on the byte's low three bits), bytes >= 128.
KMCPG_PACK is read once per process, so every variant runs in a process of its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, pickle
sys.path.insert(0, %(root)r)
import numpy as np
from tests import synth
from oracle import oracle as O
from kmcp_amd import Database, default_params
reads = pickle.load(open(%(reads)r, "rb"))
odb = O.OracleDB(%(db)r)
with Database.open(%(db)r) as db:
    res = db.search(reads, params=default_params())                  # kmcpg_search_batch
    n = synth.assert_parity(odb, res, reads)
    from kmcp_amd import lib
    seqs, offs = lib.pack_reads(reads)
    t = db.submit(seqs, offs, params=default_params())               # kmcpg_submit / kmcpg_wait
    seqs[:] = ord("N")
    res2 = db.wait(t)
    assert res2.matches.tobytes() == res.matches.tobytes() and np.array_equal(res2.qkmers, res.qkmers)
odb.close()
np.save(%(out)r, np.frombuffer(res.matches.tobytes(), dtype=np.uint8))
print("parity ok", n, int(res.qkmers.sum()))
'''


def _weird_reads(genomes, seed):
    rng = np.random.default_rng(seed)
    base = synth.sample_reads(genomes, 60, 6000, sub_rate=0.002, seed=seed, frac_random=0.1)
    out = []
    for i, r in enumerate(base):
        b = bytearray(r)
        kind = i % 6
        if kind == 1:
            b = bytearray(bytes(b).lower())                      # soft-masked
        elif kind == 2:
            for _ in range(3):                                   # N runs, short and long
                p = int(rng.integers(0, len(b) - 900))
                ln = int(rng.integers(1, 800))
                b[p:p + ln] = b"N" * ln
        elif kind == 3:
            for p in rng.integers(0, len(b), size=40):           # IUPAC codes, gaps, a byte >= 128
                b[int(p)] = int(rng.choice(list(b"RYKMSWBDHVn-*.") + [200]))
        elif kind == 4:
            b = bytearray(bytes(b).replace(b"T", b"U"))            # RNA spelling
        elif kind == 5:
            b = b[:int(rng.integers(21, 200))]                   # short ones among the long
        out.append(bytes(b))
    out.append(b"")
    out.append(b"ACGT")
    return out


@pytest.fixture(scope="module")
def world(oracle_lib, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("pack")
    genomes = synth.random_genomes(16, 30000, seed=77)
    return tmp, genomes, synth.make_db(tmp / "db", genomes, k=21, n_chunks=2, threads=4)


def _run(tmp, db_dir, reads, tag, env_extra):
    import pickle
    rp = str(tmp / f"reads_{tag}.pkl")
    pickle.dump(reads, open(rp, "wb"))
    out = str(tmp / f"matches_{tag}.npy")
    script = tmp / f"run_{tag}.py"
    script.write_text(SCRIPT % dict(root=ROOT, reads=rp, db=db_dir, out=out))
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "parity ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    return np.load(out), r.stdout


def test_packed_upload_equals_ascii_upload_and_the_oracle(world):
    tmp, genomes, db_dir = world
    reads = _weird_reads(genomes, 5)
    packed, o1 = _run(tmp, db_dir, reads, "p1", {"KMCPG_PACK": "1"})   # every batch packed, whatever its size
    plain, o0 = _run(tmp, db_dir, reads, "p0", {"KMCPG_PACK": "0"})
    assert np.array_equal(packed, plain) and len(packed) > 56 * 30
    assert o1.split()[2:] == o0.split()[2:]


def test_a_large_batch_of_long_reads_packs_by_itself(world):
    """default settings: >= 8 MB of bases at >= 1 kb per query goes up packed (10 MB here, several pack threads, pieces included)"""
    tmp, genomes, db_dir = world
    reads = synth.sample_reads(genomes, 1700, 6000, sub_rate=0.002, seed=9, frac_random=0.05, n_rate=0.0005)
    env = {k: v for k, v in (("KMCPG_PACK", None),) if v}
    os.environ.pop("KMCPG_PACK", None)
    a, _ = _run(tmp, db_dir, reads[:300] + reads, "big", env)
    assert len(a) > 56 * 1000


def test_submit_packed_equals_submit_on_the_text(world, oracle_lib):
    """kmcpg_submit_packed (round 6): the caller packs (kmcpg_pack2, record after record at any base position) and the library takes the
    codes as they are.  Same Match records, bit for bit, as kmcpg_submit on the text and as the oracle — N runs, IUPAC, lower case, U,
    bytes >= 128, empty and 4-base queries included; several tickets in flight; a multi-k database takes the unpack-on-host path."""
    import re
    from kmcp_amd import Database, default_params, lib
    tmp, genomes, db_dir = world
    O = oracle_lib
    reads = _weird_reads(genomes, 6)
    seqs, offs = lib.pack_reads(reads)
    codes, exc, total = lib.pack2(reads)                       # one kmcpg_pack2 call per query: every alignment 0..3 occurs
    assert total == int(offs[-1]) and len(exc) > 50
    fold = np.arange(256, dtype=np.uint8)
    for c, f in zip(b"acgtuU", b"ACGTTT"):
        fold[c] = f
    assert np.array_equal(lib.unpack2(codes, total, exc), fold[seqs])  # lower case and U are re-spelled, every other byte is verbatim
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir) as db:
            want = db.wait(db.submit(seqs, offs, params=default_params()))
            t1 = db.submit_packed(codes, offs, exc, params=default_params())
            t2 = db.submit_packed(codes, offs, exc, params=default_params(sort_by=2, min_qcov=0.4))
            codes_copy = codes.copy()
            codes[:] = 0xFF                                     # the caller's buffers are the caller's again at once
            r2, r1 = db.wait(t2), db.wait(t1)
            assert r1.matches.tobytes() == want.matches.tobytes() and np.array_equal(r1.qkmers, want.qkmers) and np.array_equal(r1.qlen, want.qlen)
            assert synth.assert_parity(odb, r1, reads) > 30
            assert synth.assert_parity(odb, r2, reads, oparams=O.default_params(sort_by=2, min_qcov=0.4)) > 30
            # codes in page-locked memory from kmcpg_host_alloc are uploaded from where they are (no staging copy): same records
            with lib.PinnedBytes(len(codes_copy)) as pin:
                pin.a[:len(codes_copy)] = codes_copy
                tk = [db.submit_packed(pin.a, offs, exc, params=default_params()) for _ in range(3)]
                for t_ in tk:
                    assert db.wait(t_).matches.tobytes() == want.matches.tobytes()
            pr = db.wait_pairs(db.submit_packed(codes_copy, offs, exc, params=default_params()))
            assert np.array_equal(pr.offs, want.offs) and np.array_equal(pr.pairs[:, 0], want.matches["col"])
            # runs that lie outside the batch are refused
            bad = np.array([(total - 2, 5, ord("N"))], dtype=lib.EXC_DTYPE)
            with pytest.raises(lib.KmcpGpuError):
                db.submit_packed(codes_copy, offs, bad)
            e = db.wait(db.submit_packed(codes_copy[:8], np.zeros(1, np.uint64), exc[:0]))
            assert len(e) == 0
    finally:
        odb.close()
    # a database with two k-mer sizes re-reads the text of unmatched queries: the packed entry unpacks on the host for it
    cols = []
    for gi, g in enumerate(genomes[:8]):
        h = np.concatenate([O.generate_kmers(g, O.sketch_cfg(k=k)) for k in (21, 31)])
        cols.append((f"g{gi}", len(g), 0, 1, O.sort_unique(h)))
    db2 = O.build_db(str(tmp / "db2k"), O.sketch_cfg(k=31), cols, num_hashes=1, fpr=0.1, threads=4, block_size=8)
    yml = open(db2 + "/__db.yml").read()
    open(db2 + "/__db.yml", "w").write(re.sub(r"ks:\n- 31\n", "ks:\n- 21\n- 31\n", yml))
    with Database.open(db2) as db:
        assert db.ks == [31, 21]
        p = default_params(min_qcov=0.2)
        a = db.wait(db.submit(seqs, offs, params=p))
        b = db.wait(db.submit_packed(codes_copy, offs, exc, params=p))
        assert a.matches.tobytes() == b.matches.tobytes() and np.array_equal(a.ksize, b.ksize) and len(a.matches) > 10


def test_packed_whole_genomes_are_hashed_from_their_codes(world, oracle_lib):
    """Round 6: queries above 65 536 bases that arrive as 2-bit codes (kmcpg_submit_packed, or text that kmcpg_submit packed itself) are
    hashed by k1_seg_roll2 straight from the packed stream; text is made only for the segments that hold a foreign byte.  Same Match
    records as the text upload and as the oracle: clean assemblies, N runs inside and across segments and queries, IUPAC codes, soft
    masking, short and empty queries in the same batch; the handle counts the batches that took the direct form."""
    from kmcp_amd import Database, default_params, lib
    tmp, genomes, db_dir = world
    O = oracle_lib
    rng = np.random.default_rng(11)

    def asm(ids, edit=None):
        b = bytearray(b"".join(genomes[i] for i in ids))
        if edit:
            edit(b)
        return bytes(b)

    def gaps(b):
        for _ in range(4):
            p = int(rng.integers(0, len(b) - 3000))
            ln = int(rng.integers(1, 2500))
            b[p:p + ln] = b"N" * ln
        b[65536 - 10:65536 + 40] = b"N" * 50
        b[-500:] = b"N" * 500

    def head_gap(b):
        b[:200] = b"N" * 200
        for p_ in rng.integers(0, len(b), size=25):
            b[int(p_)] = int(rng.choice(list(b"RYKMSWBDHVn")))

    reads = [asm([0, 1, 2]), asm([3, 4, 5, 6], gaps), asm([7, 8, 9], head_gap), asm([10, 11, 12]).lower(), b"", genomes[13][:5000], asm([14, 15, 0, 1, 2]),
             genomes[3][100:120], asm([5, 6, 7])[:65536 + 20]]
    seqs, offs = lib.pack_reads(reads)
    codes, exc, total = lib.pack2(reads)
    p = default_params(min_qcov=0.15)
    odb = O.OracleDB(db_dir)
    try:
        with Database.open(db_dir) as db:
            want = db.wait(db.submit(seqs, offs, params=p))         # 1.1 MB: below the size kmcpg_submit packs by itself
            assert db.k1_codes_batches() == (0, 0)
            got = db.wait(db.submit_packed(codes, offs, exc, params=p))
            assert db.k1_codes_batches() == (1, 0)
            assert got.matches.tobytes() == want.matches.tobytes() and np.array_equal(got.qkmers, want.qkmers) and np.array_equal(got.qlen, want.qlen)
            assert synth.assert_parity(odb, got, reads, oparams=O.default_params(min_qcov=0.15)) >= 15
            # several in flight, pairs; a batch of short queries alone is expanded whole (no segment path for it)
            tk = [db.submit_packed(codes, offs, exc, params=p) for _ in range(3)]
            for t_ in tk:
                assert db.wait(t_).matches.tobytes() == want.matches.tobytes()
            # >= 8 MB of text with queries of >= 1 kb on average: kmcpg_submit packs it while staging, and the whole genomes among it take the
            # direct form too (16 copies of the batch: every copy's records are the first one's)
            big = reads * 16
            sb, ob = lib.pack_reads(big)
            assert len(sb) >= (8 << 20)
            d0 = db.k1_codes_batches()[0]
            rb = db.wait(db.submit(sb, ob, params=p))
            assert db.k1_codes_batches()[0] > d0, "the staged text was not packed / not hashed from its codes"
            per = len(want.matches)
            assert len(rb.matches) == 16 * per and np.array_equal(rb.qkmers, np.tile(want.qkmers, 16))
            for c in range(16):
                assert rb.matches[c * per:(c + 1) * per].tobytes() == want.matches.tobytes(), c
            # from its fifth batch of whole genomes on a handle alternates between two k-mer workspaces and two kernel streams (the k-mer kernel of
            # one batch beside the COBS kernel of the one before): six more, four in flight, same records
            tk = [db.submit_packed(codes, offs, exc, params=p) for _ in range(4)]
            for _ in range(2):
                assert db.wait(tk.pop(0)).matches.tobytes() == want.matches.tobytes()
                tk.append(db.submit(seqs, offs, params=p))
            for t_ in tk:
                assert db.wait(t_).matches.tobytes() == want.matches.tobytes()
            # ... and from two host threads at once, each with two tickets in flight (the handle's four lanes), packed and text entries mixed
            import threading
            errs = []

            def pump(tid):
                try:
                    mine = []
                    for j in range(6):
                        mine.append(db.submit_packed(codes, offs, exc, params=p) if (tid + j) & 1 else db.submit(seqs, offs, params=p))
                        if len(mine) == 2:
                            if db.wait(mine.pop(0)).matches.tobytes() != want.matches.tobytes():
                                errs.append((tid, j))
                    for t_ in mine:
                        if db.wait(t_).matches.tobytes() != want.matches.tobytes():
                            errs.append((tid, "tail"))
                except Exception as e:  # noqa: BLE001
                    errs.append((tid, repr(e)))

            th = [threading.Thread(target=pump, args=(i,)) for i in range(2)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            assert not errs, errs
            short = [r for r in reads if len(r) <= 5000]
            c2, e2, _ = lib.pack2(short)
            s2, o2 = lib.pack_reads(short)
            a = db.wait(db.submit_packed(c2, o2, e2, params=p))
            assert db.k1_codes_batches()[1] == 1 and db.k1_codes_batches()[0] >= 12
            assert a.matches.tobytes() == db.wait(db.submit(s2, o2, params=p)).matches.tobytes()
    finally:
        odb.close()
