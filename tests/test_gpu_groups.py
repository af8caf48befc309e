"""Resident blocks with equal NumSigs are laid side by side in one group of rows (one gather serves them all); the on-disk
format and every result stay the same.  Checked here: a block holds the same bits however it is grouped, the helpers that
address single blocks (plant, read_rows) work on unaligned byte offsets inside a group, and the hit lists of the grouped and
the ungrouped layout are identical — on synthetic indexes with padding bits in every block (21 columns = 3 bytes, 5 spare
bits), partially equal NumSigs, wide remainders (group rows of 1 KB + a narrow tail) and several hash functions, and on a
database read from .uniki files."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _open(spec_or_dir, fuse, **kw):
    from kmcp_amd import Database
    old = os.environ.get("KMCPG_FUSE")
    os.environ["KMCPG_FUSE"] = "1" if fuse else "0"
    try:
        if isinstance(spec_or_dir, str):
            return Database.open(spec_or_dir, **kw)
        return Database.open_synthetic(spec_or_dir, **kw)
    finally:
        if old is None:
            os.environ.pop("KMCPG_FUSE", None)
        else:
            os.environ["KMCPG_FUSE"] = old


def _device_hits(db, reads, params):
    import torch
    from kmcp_amd import lib
    dev = torch.device("cuda:0")
    seqs, offs = lib.pack_reads(reads)
    t_seqs = torch.from_numpy(seqs).to(dev)
    t_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    cap = 1 << 16
    hits = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    qk = torch.zeros(len(reads), dtype=torch.int32, device=dev)
    ql = torch.zeros(len(reads), dtype=torch.int32, device=dev)
    db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), len(reads), len(seqs), max(len(r) for r in reads), hits.data_ptr(), cap, cnt.data_ptr(),
                    qk.data_ptr(), ql.data_ptr(), params=params)
    torch.cuda.synchronize()
    n = int(cnt[0].item())
    assert n <= cap
    h = hits[:n].cpu().numpy().astype(np.int64)
    return h[np.lexsort((h[:, 1], h[:, 0]))], qk.cpu().numpy()


@pytest.mark.parametrize("cols_per_block,n_blocks,sigs_step,num_hashes", [
    (21, 7, 0, 1),      # seven 3-byte blocks in one 21-byte row (padding bits inside the row)
    (21, 7, 1, 1),      # all NumSigs different: nothing grouped
    (312, 32, 0, 1),    # BASELINE configs[1] shape: 32 x 39 B -> one 1248-byte row = a 1-KB tile + a 256-byte tail
    (1000, 9, 0, 3),    # 125-byte blocks, three hash functions: 1125-byte row
    (14976, 2, 0, 1),   # two GTDB-width blocks: 3744-byte row
])
def test_grouped_layout_equals_ungrouped(oracle_lib, cols_per_block, n_blocks, sigs_step, num_hashes):
    from kmcp_amd import default_params, lib
    O = oracle_lib
    rng = np.random.default_rng(5)
    reads = [bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)) for _ in range(96)]
    cfg = O.sketch_cfg(k=21)
    spec = lib.SynthSpec(k=21, num_hashes=num_hashes, fpr=0.3, n_blocks=n_blocks, cols_per_block=cols_per_block, num_sigs=50021, kmers_per_col=2000,
                         seed=9, sigs_step=sigs_step)
    ncols = n_blocks * cols_per_block
    params = default_params()
    out = {}
    for fuse in (True, False):
        with _open(spec, fuse) as db:
            strides = {db.block_info(b)["stride"] for b in range(n_blocks)}
            if fuse and sigs_step == 0:
                assert len(strides) == 1 and min(strides) >= n_blocks * ((cols_per_block + 7) // 8)  # one group
            for i, r in enumerate(reads[:64]):  # planted columns: real hits in every block, first and last columns included
                km = O.generate_kmers(r, cfg)
                db.plant((i * 37) % ncols, km)
                db.plant(ncols - 1 - (i % n_blocks) * cols_per_block, km[: 80 + i % 40])
            rows = [db.read_rows(b, np.arange(0, db.block_info(b)["num_sigs"], 997, dtype=np.uint64)) for b in range(n_blocks)]
            for b in (0, n_blocks - 1):  # the contiguous read-back (strided copy for wide rows, packed by a kernel for narrow ones)
                bi = db.block_info(b)
                tail = np.zeros((5000, bi["row_bytes"]), dtype=np.uint8)
                db.read_row_range(b, bi["num_sigs"] - 5000, tail)
                assert np.array_equal(tail, db.read_rows(b, np.arange(bi["num_sigs"] - 5000, bi["num_sigs"], dtype=np.uint64)))
            hits, qk = _device_hits(db, reads, params)
            # expected counts from the rows resident in HBM, by the oracle's arithmetic (exact h % NumSigs, AND over the hashes)
            want = []
            for i, r in enumerate(reads):
                km = O.generate_kmers(r, cfg)
                cmin = max(params.min_matched, int(np.floor(len(km) * params.min_qcov)) + 1)
                for b in range(n_blocks):
                    ns = np.uint64(db.block_info(b)["num_sigs"])
                    acc = None
                    for t in range(num_hashes):
                        hv = km if num_hashes == 1 else ((km >> np.uint64(32)).astype(np.uint32) + km.astype(np.uint32) * np.uint32(t)).astype(np.uint64)
                        bits = db.read_rows(b, hv % ns)
                        acc = bits if acc is None else (acc & bits)
                    c = np.unpackbits(acc, axis=1)[:, :cols_per_block].sum(axis=0)
                    want += [(i, b * cols_per_block + int(col), int(c[col])) for col in np.nonzero(c >= cmin)[0]]
            assert sorted(map(tuple, hits.tolist())) == sorted(want)
            out[fuse] = (rows, hits, qk)
    for a, b in zip(out[True][0], out[False][0]):
        assert np.array_equal(a, b)  # the same bits in every block, grouped or not
    assert np.array_equal(out[True][1], out[False][1]) and np.array_equal(out[True][2], out[False][2])
    assert len(out[True][1]) >= 100


def test_grouped_database_from_files(oracle_lib, tmp_path):
    """Equal-length chunks give equal NumSigs in every block (BASELINE configs[1] is built this way): the file loader repacks the
    blocks into one group; results equal the oracle's and the ungrouped layout's."""
    from kmcp_amd import default_params
    O = oracle_lib
    genomes = synth.random_genomes(24, 6000, seed=21)
    db_dir = synth.make_db(tmp_path, genomes, k=21, n_chunks=1, threads=3)  # 3 blocks of 8 columns, 5989 k-mers in every column
    reads = synth.sample_reads(genomes, 300, 150, sub_rate=0.02, seed=3, frac_random=0.1)
    odb = O.OracleDB(db_dir)
    res = {}
    for fuse in (True, False):
        with _open(db_dir, fuse, device=0) as db:
            nb = db.info.n_blocks
            sigs = {db.block_info(b)["num_sigs"] for b in range(nb)}
            assert nb == 3 and len(sigs) == 1
            r = db.search(reads, params=default_params())
            assert synth.assert_parity(odb, r, reads) > 200
            res[fuse] = [[(int(m["col"]), int(m["mkmers"])) for m in r.read(i)] for i in range(len(reads))]
    odb.close()
    assert res[True] == res[False]


def test_uniki_files_with_padding_bits_set_in_fused_groups(oracle_lib, tmp_path):
    """the hit emission of k2_cobs no longer validates a set bit against the segment table — it relies on the load
    (k_repack) clearing the padding bits of every member's last byte.  Files `kmcp index` writes never carry such bits
    (index.go:1157); here they are set in every row of every block of a database whose blocks (12 columns: 4 spare bits each) share
    NumSigs and are therefore laid side by side in one group: the results — grouped and ungrouped — are those of the clean files
    and of the oracle, and no hit names a column that does not exist."""
    import shutil
    from kmcp_amd import default_params, lib
    O = oracle_lib
    genomes = synth.random_genomes(44, 5000, seed=61)
    cols = synth.make_columns(genomes, O.sketch_cfg(k=21))
    clean = lib.build_db(str(tmp_path / "clean"), cols, k=21, threads=4, block_size=12, uniform_sigs=1)  # blocks of 12, 12, 12, 8 columns
    dirty = str(tmp_path / "dirty" / "R001")
    shutil.copytree(os.path.dirname(clean), os.path.dirname(dirty))
    reads = synth.sample_reads(genomes, 500, 150, sub_rate=0.01, seed=62, frac_random=0.2)
    with _open(clean, True) as db:
        nb = int(db.info.n_blocks)
        bi = [db.block_info(b) for b in range(nb)]
        assert len({b["num_sigs"] for b in bi}) == 1 and [b["n_cols"] for b in bi] == [12, 12, 12, 8]
        assert len({b["stride"] for b in bi}) == 1 and bi[0]["stride"] >= 16  # one group: 2 + 2 + 2 + 1 bytes side by side
        want = db.search(reads, params=default_params(min_qcov=0.31, min_matched=1))
    files = sorted(f for f in os.listdir(dirty) if f.endswith(".uniki"))
    assert len(files) == nb
    touched = 0
    for f, b in zip(files, bi):
        spare = (8 - b["n_cols"] % 8) % 8
        if not spare:
            continue
        path = os.path.join(dirty, f)
        raw = bytearray(open(path, "rb").read())
        hdr = len(raw) - b["num_sigs"] * b["row_bytes"]  # the rows are the tail of the file (serialization.go:140, :379)
        rows = np.frombuffer(raw, dtype=np.uint8, offset=hdr).reshape(b["num_sigs"], b["row_bytes"])
        rows[:, -1] |= np.uint8((1 << spare) - 1)  # columns n_cols .. 8 * row_bytes - 1: bits 0 .. spare-1 of the last byte
        open(path, "wb").write(raw)
        touched += 1
    assert touched == 3
    for fuse in (True, False):
        with _open(dirty, fuse) as db:
            got = db.search(reads, params=default_params(min_qcov=0.31, min_matched=1))
            assert int(got.matches["col"].max()) < 44
            for f in ("qlen", "qkmers", "offs"):
                assert np.array_equal(getattr(got, f), getattr(want, f)), (fuse, f)
            assert got.matches.tobytes() == want.matches.tobytes(), fuse
    odb = O.OracleDB(clean)
    try:
        assert synth.assert_parity(odb, want, reads, None, O.default_params(min_qcov=0.31, min_matched=1)) > 300
    finally:
        odb.close()
