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
        """Every rank passes the same batch (host arrays).  Returns a BatchResult on rank 0, None elsewhere."""
        from .lib import default_params
        p = params or default_params()
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
