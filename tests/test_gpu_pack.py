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
