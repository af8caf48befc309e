#!/usr/bin/env python3
"""bench.py — reads/s searched (150 bp, k=21) against a GTDB-scale COBS index on MI355X.

One "step" = one pass of the hot path (K1 ntHash k-mer generation + K2 COBS query + hit hand-over) over
one batch of synthetic 150-bp reads that is already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gtdb|config1] [--batch-reads B]

N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`:
one rank per GPU; the index's independent blocks are partitioned over the ranks (libkmcpgpu shards by
bytes), every rank searches the whole batch against its blocks and the per-read hit lists are gathered on
rank 0 over RCCL.  Total work is fixed as N grows => "scaling": "strong".

The synthetic index is generated directly in HBM: 32 blocks x 14 976 columns
(NumRowBytes 1 872) x 968 700 rows = 58.03 GB, bits i.i.d. Bernoulli(0.30) like a Bloom filter at fpr 0.3;
90 % of the reads are 1 %-mutated, randomly reverse-complemented copies of 150-bp fragments whose k-mers
were planted into a random column, 10 % are uniform random.

The JSON line reports the GTDB-scale workload (the configuration BASELINE.json's metric is quoted on).  At N=1 the
same line carries, under "secondary", the numbers of BASELINE.json configs[1] (10 k chunks, 39-byte rows).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.2-6.3 TB/s is what copies/gathers reach

WORKLOADS = {
    # GTDB r202 k=21 x10 chunks: 58.03 GB in 32 blocks (docs/database-time-and-mem-v2021.12.md:20-36)
    # every block has its own NumSigs, as in a real database (blocks hold genomes of ascending size): 967 708 + 64 i rows
    "gtdb": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=14976, num_sigs=967708, sigs_step=64, kmers_per_col=345510,
                 batch_reads=524288, kernel="k2_cobs<64,8,false>",
                 name="gtdb-scale synthetic: 32 blocks x 14976 cols x 968700 sigs (58.03 GB), 150bp k=21"),
    # 10 k chunks, `kmcp index -j 32`: 32 blocks x 312 columns, 39-byte rows (BASELINE.json configs[1])
    "config1": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=312, num_sigs=1121470, kmers_per_col=400000,
                    batch_reads=1048576, kernel="k2_cobs<4,8,false>",
                    name="10k-chunk synthetic: 32 blocks x 312 cols x 1121470 sigs (1.4 GB), 150bp k=21"),
    # the same 9 984 columns indexed as ONE block (`kmcp index -b 9984`): 130 gathers of 1 248 B per read instead of 4 160 of 39 B
    "config1_wide": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=9984, num_sigs=1121470, kmers_per_col=400000,
                         batch_reads=1048576, kernel="k2_cobs<64,8,false>",
                         name="10k-chunk synthetic as one block: 1 x 9984 cols x 1121470 sigs (1.4 GB), 150bp k=21"),
}
READ_LEN = 150


def effective_cpus():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def make_batch(dev, n_reads, n_cols, seed):
    """(fragments to plant, their target columns, the reads actually searched), all uint8/int32 on device."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    code = torch.randint(0, 4, (n_reads, READ_LEN), generator=g, device=dev)
    cols = torch.randint(0, n_cols, (n_reads,), generator=g, device=dev).to(torch.int32)
    is_random = torch.rand(n_reads, generator=g, device=dev) < 0.10
    cols[is_random] = -1  # 0xFFFFFFFF: not planted
    # the read: 1 % substitutions, then half of them reverse-complemented
    sub = torch.rand(n_reads, READ_LEN, generator=g, device=dev) < 0.01
    rcode = torch.where(sub, torch.randint(0, 4, (n_reads, READ_LEN), generator=g, device=dev), code)
    rc = torch.rand(n_reads, generator=g, device=dev) < 0.5
    rcode = torch.where(rc[:, None], 3 - rcode.flip(1), rcode)  # A<->T, C<->G under the ACGT code
    frag = acgt[code].contiguous().view(-1)
    reads = acgt[rcode].contiguous().view(-1)
    offs = (torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * READ_LEN).contiguous()
    return frag, cols.contiguous(), reads, offs


class Ctx:
    pass


def run_workload(name, ctx, steps, warmup, batch_reads=0, cpu_baseline=True, cpu_target_s=8.0, cpu_sample_reads=0):
    """Builds the synthetic index of workload `name` in HBM, times `steps` steps, returns the result dict (rank 0)."""
    from kmcp_amd import Database, default_params, lib
    from kmcp_amd.dist import gather_hits

    world, rank, dev, dev_index = ctx.world, ctx.rank, ctx.dev, ctx.dev_index
    wl = dict(WORKLOADS[name])
    B = batch_reads or wl["batch_reads"]
    spec = lib.SynthSpec(k=wl["k"], num_hashes=wl["num_hashes"], fpr=wl["fpr"], n_blocks=wl["n_blocks"],
                         cols_per_block=wl["cols_per_block"], num_sigs=wl["num_sigs"], kmers_per_col=wl["kmers_per_col"], seed=42,
                         sigs_step=wl.get("sigs_step", 0))
    free_b, _ = torch.cuda.mem_get_info(dev)
    need = wl["n_blocks"] * wl["num_sigs"] * ((wl["cols_per_block"] + 7) // 8 + 64) / world
    if need > 0.9 * free_b:
        raise SystemExit(f"workload needs {need/1e9:.1f} GB of HBM on this rank, {free_b/1e9:.1f} GB free")
    t0 = time.time()
    db = Database.open_synthetic(spec, device=dev_index, shard_rank=rank, shard_count=world)
    torch.cuda.synchronize()
    info = db.info
    n_cols = int(info.n_cols)
    params = default_params()  # kmcp search defaults: -t 0.55 -c 10 -m 30 -f 0.01 -u 256
    db.set_profiling(True)

    # ---- batches resident in HBM; distinct data per step (cycled if K+W is large)
    n_batches = max(1, min(steps + warmup, 4))
    batches = []
    for i in range(n_batches):
        frag, cols, reads, offs = make_batch(dev, B, n_cols, seed=1000 + i)
        db.plant_reads_device(frag.data_ptr(), offs.data_ptr(), B, B * READ_LEN, READ_LEN, cols.data_ptr())
        batches.append((reads, offs, cols))
        del frag
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    cap = 4 * B + 4096
    d_hits = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    d_qk = torch.zeros(B, dtype=torch.int32, device=dev)
    d_ql = torch.zeros(B, dtype=torch.int32, device=dev)
    h_hits = torch.empty((cap * world, 3), dtype=torch.int32).pin_memory()
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(i):
        """K1+K2 on this rank's blocks, hit lists to rank 0 (RCCL), hit tuples to host memory. Returns #hits on rank 0.
        Holds collectives: every rank must call it the same number of times."""
        reads, offs, _ = batches[i % n_batches]
        db.query_device(reads.data_ptr(), offs.data_ptr(), B, B * READ_LEN, READ_LEN, d_hits.data_ptr(), cap, d_cnt.data_ptr(),
                        d_qk.data_ptr(), d_ql.data_ptr(), params=params, stream=stream)
        if world == 1:
            n = int(d_cnt[0].item())
            assert n <= cap, "hit buffer overflow"
            h_hits[:n].copy_(d_hits[:n], non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            return n
        parts = gather_hits(d_hits, d_cnt[:1], dst=0)  # RCCL: all_gather(counts) + gather(hit buffers) over xGMI
        if rank != 0:
            torch.cuda.current_stream(dev).synchronize()
            return 0
        pos = 0
        for part in parts:
            c = part.shape[0]
            h_hits[pos:pos + c].copy_(part, non_blocking=True)
            pos += c
        torch.cuda.current_stream(dev).synchronize()
        return pos

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    barrier()
    k2_ms, k1_ms, n_hits_total = [], [], 0
    t_start = time.perf_counter()
    for i in range(steps):
        n_hits_total += step(warmup + i)
        a, b = db.last_timing()
        k1_ms.append(a)
        k2_ms.append(b)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if ctx.same_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (k2_cobs) on this rank: algorithmic bytes per launch:
    #      sum over reads of kept k-mers x sum over local blocks of numHashes x NumRowBytes, + qLen, + 12 B per hit
    last = (warmup + steps - 1) % n_batches
    qk = d_qk.cpu().numpy().astype(np.int64)
    kmers_per_launch = int(qk.sum())
    alg_bytes = kmers_per_launch * int(info.row_bytes_sum_local) * int(info.num_hashes) + B * READ_LEN + 12 * (n_hits_total // max(1, steps))
    k2_avg_ms = float(np.mean(k2_ms))
    achieved = alg_bytes / (k2_avg_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            key = f"{name}:{B}:{world}"
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

    out = {
        "metric": "reads/sec searched (150bp, k=21) vs GTDB-scale index",
        "value": B * steps / elapsed,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32 bitwise (bit-sliced counters), u64 hashes",
        "data": "synthetic",
        "config": {"workload": wl["name"], "batch_reads": B, "read_len": READ_LEN, "k": wl["k"], "num_hashes": wl["num_hashes"],
                   "index_bytes": int(info.matrix_bytes), "index_bytes_this_rank": int(info.matrix_bytes_local),
                   "blocks": int(info.n_blocks), "columns": n_cols, "parallelism": f"block-shard x{world}",
                   "search_flags": "-t 0.55 -c 10 -m 30 -f 0.01 -u 256"},
        "roofline": {"bound": "hbm", "kernel": wl["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms": k2_avg_ms, "kmers_kernel_ms": float(np.mean(k1_ms))},
        "hits_per_step": n_hits_total / max(1, steps),
        "setup_s": setup_s,
    }

    # ---- sanity on the last batch: planted reads must come back with their column (the step holds collectives: every
    #      rank takes part, rank 0 evaluates the merged hit list)
    n_last = step(last)
    hh = None
    if rank == 0:
        hh = h_hits[:n_last].numpy().astype(np.int64)
        cols_last = batches[last][2].cpu().numpy().astype(np.int64)
        got = set(zip(hh[:, 0].tolist(), hh[:, 1].tolist()))
        planted = np.nonzero(cols_last >= 0)[0]
        out["planted_recall"] = sum((int(r), int(cols_last[r])) in got for r in planted[:20000]) / max(1, min(len(planted), 20000))

    # ---- the drop-in boundary with host buffers (PCIe-inclusive; reported beside `value`, never as `value`): one batch through
    #      kmcpg_search_batch — reads in host memory in; H2D, K1, K2, D2H of the hits, float64 thresholds, FPR, sort; matches out
    if rank == 0 and world == 1:
        reads_hb = batches[last][0].cpu().numpy()
        offs_hb = batches[last][1].cpu().numpy().astype(np.uint64)
        db.search_packed_count(reads_hb, offs_hb, params=params)  # first call sizes the staging buffers
        t1 = time.perf_counter()
        n_matches = db.search_packed_count(reads_hb, offs_hb, params=params)  # the C call alone, result freed, nothing copied to numpy
        dt = time.perf_counter() - t1
        out["host_boundary"] = {"value": B / dt, "unit": "reads/s", "ms_per_batch": dt * 1e3, "matches": n_matches,
                                "note": "kmcpg_search_batch: host buffers in, finalized matches out (PCIe + host finalize included)"}
        del reads_hb, offs_hb

    # ---- CPU baseline: the oracle (C restatement of the reference algorithm), timed on this box's host cores on a bounded
    #      sample: the first S blocks copied back from HBM and the first R reads of the last batch.
    if rank == 0 and world == 1 and cpu_baseline:
        from oracle import oracle as O
        S = 2 if name == "gtdb" else wl["n_blocks"]  # gtdb: 2 of the 32 blocks x the whole batch is ~10 s on 16 threads
        blocks = []
        for b in range(S):
            bi = db.block_info(b)
            rows = np.empty((bi["num_sigs"], bi["row_bytes"]), dtype=np.uint8)
            chunk = 65536
            for r0 in range(0, bi["num_sigs"], chunk):
                idx = np.arange(r0, min(bi["num_sigs"], r0 + chunk), dtype=np.uint64)
                rows[r0:r0 + len(idx)] = db.read_rows(b, idx)
            blocks.append((bi["num_sigs"], bi["n_cols"], bi["col_base"], rows))
        odb = O.OracleDB.from_memory(O.sketch_cfg(k=wl["k"]), wl["num_hashes"], wl["fpr"], blocks, wl["kmers_per_col"])
        threads = effective_cpus()
        reads_h = batches[last][0].cpu().numpy()
        offs_h = batches[last][1].cpu().numpy().astype(np.uint64)
        R = cpu_sample_reads or 256
        tcpu = 0.0
        while True:  # grow the sample until it is several seconds of CPU work
            t1 = time.perf_counter()
            oqk, ohits = odb.search_batch(reads_h[:R * READ_LEN], offs_h[:R + 1], O.default_params(), threads=threads)
            tcpu = time.perf_counter() - t1
            if cpu_sample_reads or tcpu >= cpu_target_s or R >= B:
                break
            R = min(B, int(R * max(2.0, 1.5 * cpu_target_s / max(tcpu, 1e-3))))
        # same-run parity on the sample: GPU hits of these reads restricted to the sampled blocks == oracle hits
        hi_col = blocks[-1][2] + blocks[-1][1]
        g = hh[(hh[:, 0] < R) & (hh[:, 1] < hi_col)]
        g = g[np.lexsort((g[:, 1], g[:, 0]))]
        parity = bool(np.array_equal(g, ohits.astype(np.int64))) and bool(np.array_equal(oqk[:R], qk[:R]))
        frac_blocks = S / wl["n_blocks"]
        out["cpu_baseline"] = {"value": R / tcpu * frac_blocks, "unit": "reads/s", "cores": threads, "kind": "port",
                               "sample": f"{R} reads x {S} of {wl['n_blocks']} blocks in {tcpu:.2f} s on {threads} threads "
                                         f"(oracle ko_search_batch, index rows copied back from HBM); value scaled by {frac_blocks:.4f} "
                                         "to the whole index", "parity_on_sample": parity, "sample_hits": int(len(ohits))}
        odb.close()
        del blocks
        assert parity, "GPU hits differ from the CPU oracle on the sample"

    db.close()
    del d_hits, batches
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="gtdb", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-reads", type=int, default=0)
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = size the CPU sample to several seconds of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] numbers that ride along at N=1")
    args = ap.parse_args()

    ctx = Ctx()
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != ctx.world:
        if ctx.world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
                             "--master-port P bench.py --gpus N ...")
        args.gpus = ctx.world
    # KMCP_BENCH_SAME_GPU=1 (debugging on a 1-GPU box): every rank uses GPU 0 and the exchange runs over gloo, because
    # RCCL refuses two ranks on one device.  Never set by the driver; the measured path is nccl = RCCL over xGMI.
    ctx.same_gpu = os.environ.get("KMCP_BENCH_SAME_GPU") == "1"
    ctx.dev_index = 0 if ctx.same_gpu else local_rank
    torch.cuda.set_device(ctx.dev_index)
    ctx.dev = torch.device("cuda", ctx.dev_index)
    if ctx.world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ctx.same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=ctx.dev)

    out = run_workload(args.workload, ctx, args.steps, args.warmup, args.batch_reads, cpu_baseline=not args.no_cpu_baseline,
                       cpu_sample_reads=args.cpu_sample_reads)
    if ctx.world == 1 and args.workload == "gtdb" and not args.no_secondary and not args.batch_reads:
        sec = run_workload("config1", ctx, min(args.steps, 3), 1, cpu_baseline=not args.no_cpu_baseline, cpu_target_s=3.0)
        keys = ("value", "unit", "ms_per_step", "config", "roofline", "planted_recall", "host_boundary", "cpu_baseline")
        out["secondary"] = {"config1": {k: sec[k] for k in keys if k in sec}}
        # the same columns indexed as one wide block: what the block layout (`kmcp index -b`) is worth on this hardware
        wide = run_workload("config1_wide", ctx, min(args.steps, 3), 1, cpu_baseline=False)
        out["secondary"]["config1_wide"] = {k: wide[k] for k in keys if k in wide}
    if ctx.rank == 0:
        print(json.dumps(out))
    if ctx.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
