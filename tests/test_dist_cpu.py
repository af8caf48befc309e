"""The N>1 path on CPU (gloo, world_size 2): block sharding by libkmcpgpu (metadata-only handles), the hit-list
exchange of kmcp_amd.dist and kmcpg_finalize on rank 0.  The GPU half of each rank is stood in for by the oracle's
per-block counts (test-only), so the merged result must equal the oracle's single-process answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_hits(O, odb, db, reads, p):
    """(read, col, count) of this rank's resident blocks, integer thresholds only (what k2_cobs emits)."""
    hits, qk, ql = [], [], []
    local = [b for b in range(db.info.n_blocks) if db.block_info(b)["local"]]
    for i, r in enumerate(reads):
        ql.append(len(r))
        if len(r) < p.min_qlen:
            qk.append(0)
            continue
        km = O.generate_kmers(r, odb.cfg)
        if len(km) < p.min_matched:
            qk.append(0)
            continue
        if len(km) > p.dedup_threshold:
            km = O.sort_unique(km)
        n = len(km)
        qk.append(n)
        cmin = max(p.min_matched, int(np.floor(float(n) * p.min_qcov)) + 1)
        for b in local:
            cnt = odb.block_counts(b, km)
            base = db.block_info(b)["col_base"]
            for c in np.nonzero(cnt >= cmin)[0]:
                hits.append((i, base + int(c), int(cnt[c])))
    return hits, qk, ql, local


def _worker(rank, world, db_dir, port, reads, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmcp_amd.dist import gather_hits, hits_checksum, hits_to_numpy
        from kmcp_amd.lib import Database, default_params
        from oracle import oracle as O
        odb = O.OracleDB(db_dir)
        db = Database.open(db_dir, device=-1, shard_rank=rank, shard_count=world)  # metadata only: no GPU here
        p = default_params()
        hits, qk, ql, local = _shard_hits(O, odb, db, reads, p)
        # every block lives on exactly one rank
        flags = torch.zeros(db.info.n_blocks, dtype=torch.int64)
        flags[local] = 1
        dist.all_reduce(flags)
        assert bool((flags == 1).all())
        cap = 4 * len(reads) + 16
        buf = torch.zeros((cap, 3), dtype=torch.int32)
        if hits:
            buf[:len(hits)] = torch.tensor(hits, dtype=torch.int64).to(torch.int32)
        parts = gather_hits(buf, torch.tensor([len(hits)], dtype=torch.int64), dst=0)
        if rank == 0:
            res = db.finalize(hits_to_numpy(parts), np.array(qk, dtype=np.int32), np.array(ql, dtype=np.int32), params=p)
            n = synth.assert_parity(odb, res, reads)
            ret["hits"] = n
            ret["per_rank"] = [int(x.shape[0]) for x in parts]
            # the N-invariant the bench line prints (bench.py sanity_batch.hits_checksum): the merged list of `world` shards has
            # the checksum of the one-shard list, in whatever order the shards' pieces arrive
            merged = torch.cat(parts).numpy().astype(np.int64)
            ret["checksum"] = hits_checksum(merged)
            ret["checksum_shuffled"] = hits_checksum(merged[np.random.default_rng(rank).permutation(len(merged))])
            with Database.open(db_dir, device=-1, shard_rank=0, shard_count=1) as one:  # N = 1: every block on one rank
                h1 = _shard_hits(O, odb, one, reads, p)[0]
            ret["checksum_one_rank"] = hits_checksum(np.array(h1, dtype=np.int64).reshape(-1, 3))
        else:
            assert parts is None
        # GPU entry points must refuse a metadata-only handle
        from kmcp_amd.lib import KmcpGpuError
        try:
            db.kmers_device(0, 0, 0, 0, 0, 1, 0, None, 1)
            raise AssertionError("metadata-only handle did GPU work")
        except KmcpGpuError as e:
            assert "metadata-only" in str(e) or "null" in str(e)
        with Database.open(db_dir, device=-1) as whole:
            try:
                whole.search(reads[:1])
                raise AssertionError("metadata-only handle searched")
            except KmcpGpuError as e:
                assert "metadata-only" in str(e)
        db.close()
        odb.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_finalize_gloo(oracle_lib, tmp_path, world):
    genomes = synth.random_genomes(30, 9000, seed=50)
    db_dir = synth.make_db(tmp_path, genomes, k=21, n_chunks=2, overlap=150, threads=8)  # 60 columns -> 8 blocks
    reads = synth.sample_reads(genomes, 150, 150, sub_rate=0.01, seed=51, frac_random=0.1) + [b"", genomes[0][:25]]
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, db_dir, _free_port(), reads, ret), nprocs=world, join=True)
        assert ret["hits"] > 100
        assert sum(1 for x in ret["per_rank"] if x > 0) >= 2  # the hits really came from several shards
        assert ret["checksum"] == ret["checksum_one_rank"] == ret["checksum_shuffled"] != "0000000000000000"


def test_shard_balance_by_bytes(oracle_lib, tmp_path):
    """Greedy partition by matrix bytes: uneven blocks end up balanced."""
    from kmcp_amd.lib import Database
    genomes = synth.random_genomes(40, 3000, seed=52) + synth.random_genomes(9, 30000, seed=53)
    db_dir = synth.make_db(tmp_path, genomes, k=21, threads=6)
    loads = []
    for r in range(4):
        with Database.open(db_dir, device=-1, shard_rank=r, shard_count=4) as db:
            loads.append(int(db.info.matrix_bytes_local))
            total = int(db.info.matrix_bytes)
    assert sum(loads) == total
    assert max(loads) <= 0.5 * total
