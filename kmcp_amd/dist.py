"""Multi-GPU driver: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The index's independent .uniki blocks are partitioned over the ranks by libkmcpgpu (kmcpg_open with
shard_rank/shard_count); every rank searches the whole batch against its blocks; the per-read hit lists are
brought to rank 0 — the one real exchange step of the path (the reference concatenates the replies of its
per-block goroutines the same way, kmcp/cmd/util-db-search.go:946-964) — and finalized there.

The collective part works on any backend (the CPU tests run it over gloo with world_size 2).
"""
import numpy as np
import torch
import torch.distributed as dist

from .lib import HIT_DTYPE


def gather_hits(hits: torch.Tensor, count: torch.Tensor, dst: int = 0, group=None, force_collectives: bool = False):
    """hits: int32 [cap, 3] buffer of this rank, count: int64 [1] number of valid rows (device tensors for nccl,
    CPU tensors for gloo).  Returns on `dst` a list with every rank's valid rows (tensors on the same device,
    rank order), None elsewhere.  Two collectives: all_gather of the 8-byte counts, gather of the padded buffers —
    a few bytes per read, nowhere near the xGMI links' 153 GB/s."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not force_collectives:  # (tests run the collectives on a one-rank RCCL group too)
        return [hits[:int(count.item())]]
    if dist.get_backend(group) == "gloo" and hits.is_cuda:
        # debugging aid (several ranks sharing one GPU, where RCCL refuses duplicate devices): stage through the host
        hits, count = hits.cpu(), count.cpu()
    counts = torch.zeros(world, dtype=torch.int64, device=hits.device)
    dist.all_gather_into_tensor(counts, count.reshape(1).to(torch.int64), group=group)
    c = counts.cpu().tolist()  # every rank learns every count: the gather below moves only max(c) rows per rank
    m = max(c)
    if m > hits.shape[0]:
        raise OverflowError(f"hit buffer overflow on a rank: {m} > {hits.shape[0]}")
    if m == 0:
        return [hits[:0] for _ in range(world)] if rank == dst else None
    send = hits[:m].contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return [bufs[r][:c[r]] for r in range(world)]


def hits_to_numpy(parts):
    """list of int32 [n_i, 3] tensors -> one HIT_DTYPE array (read, col, count)."""
    if not parts:
        return np.zeros(0, dtype=HIT_DTYPE)
    cat = torch.cat([p.reshape(-1, 3) for p in parts]).cpu().contiguous().numpy()
    return np.ascontiguousarray(cat).view(np.uint32).reshape(-1, 3).copy().view(HIT_DTYPE).reshape(-1)


def hits_checksum(h):
    """Order-independent 64-bit checksum of a hit list (int array [n, 3] of (read, column, count)): the sum mod 2^64 of a mixed
    64-bit word per tuple.  The synthetic index and the batches are functions of the seeds only, so the merged list — and this
    number — must be the same at N = 1, 2, 4, 8 GPUs."""
    if len(h) == 0:
        return "0000000000000000"
    a = np.ascontiguousarray(h).astype(np.uint64)
    with np.errstate(over="ignore"):
        x = a[:, 0] * np.uint64(0x9E3779B97F4A7C15) + a[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) + a[:, 2] * np.uint64(0x165667B19E3779F9)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
        return "%016x" % int(np.add.reduce(x, dtype=np.uint64))


def _take_reads(seqs, offs, idx):
    """the reads `idx` of a packed batch as a packed batch of their own"""
    o = offs.astype(np.int64)
    lens = o[idx + 1] - o[idx]
    so = np.concatenate([[0], np.cumsum(lens)])
    src = np.repeat(o[idx] - so[:-1], lens) + np.arange(int(so[-1]))
    return np.ascontiguousarray(seqs[src]), so.astype(offs.dtype)


class ShardedSearcher:
    """kmcp search over N GPUs of one node: rank r holds shard r of the database."""

    def __init__(self, db_dir, device=None, group=None):
        from .lib import Database
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dev_index = torch.cuda.current_device() if device is None else device
        self.dev = torch.device("cuda", self.dev_index)
        self.db = Database.open(db_dir, device=self.dev_index, shard_rank=self.rank, shard_count=self.world)

    def search(self, seqs: np.ndarray, offs: np.ndarray, params=None, seqs2=None, offs2=None):
        """Every rank passes the same batch (host arrays).  Returns a BatchResult on rank 0, None elsewhere.

        A database with several k-mer sizes is walked as the reference does (util-db-search.go:764, :1016-1022): queries that
        were searched with the largest k and matched nothing go again with the next smaller one — rank 0 knows which (it holds
        the finalized result) and tells the others.  --try-se (paired input) searches the mates of such queries on their own
        first (:831-850, :1001-1014)."""
        from .lib import default_params
        p = params or default_params()
        n = len(offs) - 1
        try_se = bool(p.try_se) and seqs2 is not None
        q0 = type(p).from_buffer_copy(p)
        q0.try_se = 0
        res = self._search_once(seqs, offs, q0, seqs2, offs2)
        ks = [int(p.k)] if p.k > 0 else list(self.db.ks)
        if n == 0 or (len(ks) < 2 and not try_se):
            return res
        final = None
        if self.rank == 0:  # matched, or never searched (too short / fewer than MinMatched k-mers): final (:854-869)
            final = (np.diff(res.offs.astype(np.int64)) > 0) | (res.qkmers <= 0)

        def again(k, mate, whole_query):
            """the queries rank 0 still holds open, searched again: all of the query (mate None) or one mate on its own"""
            todo = self._bcast_indices(np.nonzero(~final)[0].astype(np.int64) if self.rank == 0 else None)
            if len(todo) == 0:
                return False
            q = type(p).from_buffer_copy(p)
            q.k = k
            q.try_se = 0
            if mate is None:
                sub, so = _take_reads(seqs, offs, todo)
                sub2, so2 = _take_reads(seqs2, offs2, todo) if seqs2 is not None else (None, None)
            else:
                sub, so = _take_reads(seqs2 if mate else seqs, offs2 if mate else offs, todo)
                sub2 = so2 = None
                q.min_qlen = 0  # the length gate was applied once, before k-mer generation (:831-850)
            r2 = self._search_once(sub, so, q, sub2, so2)
            if self.rank != 0:
                return True
            # splice the sub-batch into the batch result (the queries of `todo` had no matches so far)
            res.qlen[todo] = r2.qlen
            res.ksize[todo] = k
            searched = r2.qkmers > 0
            if whole_query:
                res.qkmers[todo] = np.where(searched, r2.qkmers, 0)  # a fresh QueryResult: NumKmers never set
            else:
                res.qkmers[todo] = np.where(searched, r2.qkmers, res.qkmers[todo])
            got = np.diff(r2.offs.astype(np.int64))
            final[todo[~searched | (got > 0)]] = True
            if got.sum():
                old_cnt = np.diff(res.offs.astype(np.int64))
                new_cnt = old_cnt.copy()
                new_cnt[todo] = got
                new_offs = np.concatenate([[0], np.cumsum(new_cnt)]).astype(res.offs.dtype)
                merged = np.empty(int(new_offs[-1]), dtype=res.matches.dtype)
                owner = np.repeat(np.arange(n), old_cnt)
                merged[new_offs[owner].astype(np.int64) + (np.arange(len(res.matches)) - res.offs[owner].astype(np.int64))] = res.matches
                owner2 = np.repeat(np.arange(len(todo)), got)
                merged[new_offs[todo[owner2]].astype(np.int64) + (np.arange(len(r2.matches)) - r2.offs[owner2].astype(np.int64))] = r2.matches
                res.matches, res.offs = merged, new_offs
            return True

        # the order of handleQuery: the whole query, then with --try-se read 1 and read 2 on their own (:831-850, :1001-1014),
        # then everything again with the next smaller k (:1016-1022)
        for ik, k in enumerate(ks):
            if ik > 0 and not again(k, None, True):
                break
            if try_se:
                for mate in (0, 1):
                    if not again(k, mate, False):
                        break
        return res

    def _bcast_indices(self, idx):
        """rank 0's int64 index array on every rank"""
        if self.world == 1:
            return idx
        on_host = dist.get_backend(self.group) == "gloo"
        dev = torch.device("cpu") if on_host else self.dev
        m = torch.tensor([len(idx) if self.rank == 0 else 0], dtype=torch.int64, device=dev)
        dist.broadcast(m, src=0, group=self.group)
        t = torch.from_numpy(idx).to(dev) if self.rank == 0 else torch.empty(int(m.item()), dtype=torch.int64, device=dev)
        if int(m.item()):
            dist.broadcast(t, src=0, group=self.group)
        return t.cpu().numpy()

    def _search_once(self, seqs, offs, p, seqs2=None, offs2=None):
        n = len(offs) - 1
        t_seqs = torch.from_numpy(seqs).to(self.dev)
        t_offs = torch.from_numpy(offs.view(np.int64)).to(self.dev)
        t2 = o2 = None
        total = int(offs[-1])
        maxlen = int(np.diff(offs.astype(np.int64)).max()) if n else 0
        if seqs2 is not None:
            t2 = torch.from_numpy(seqs2).to(self.dev)
            o2 = torch.from_numpy(offs2.view(np.int64)).to(self.dev)
            total += int(offs2[-1])
            maxlen = max(maxlen, int(np.diff(offs2.astype(np.int64)).max()) if n else 0)
        cap = 8 * n + 4096
        while True:
            hits = torch.empty((cap, 3), dtype=torch.int32, device=self.dev)
            cnt = torch.zeros(2, dtype=torch.int64, device=self.dev)
            qk = torch.zeros(max(n, 1), dtype=torch.int32, device=self.dev)
            ql = torch.zeros(max(n, 1), dtype=torch.int32, device=self.dev)
            self.db.query_device(t_seqs.data_ptr(), t_offs.data_ptr(), n, total, maxlen, hits.data_ptr(), cap, cnt.data_ptr(),
                                 qk.data_ptr(), ql.data_ptr(), params=p, d_seqs2=t2.data_ptr() if t2 is not None else None,
                                 d_offs2=o2.data_ptr() if o2 is not None else None, stream=torch.cuda.current_stream(self.dev).cuda_stream)
            need = cnt[:1].clone()
            if self.world > 1:
                if dist.get_backend(self.group) == "gloo":
                    need = need.cpu()  # debugging aid, see gather_hits
                dist.all_reduce(need, op=dist.ReduceOp.MAX, group=self.group)
            if int(need.item()) <= cap:
                break
            cap = int(need.item()) * 5 // 4  # some rank overflowed its buffer: every rank reruns with room
        parts = gather_hits(hits, cnt[:1], dst=0, group=self.group) if self.world > 1 else [hits[:int(cnt[0].item())]]
        if self.rank != 0:
            return None
        return self.db.finalize(hits_to_numpy(parts), qk[:n].cpu().numpy(), ql[:n].cpu().numpy(), params=p)

    def close(self):
        self.db.close()
