#!/usr/bin/env python3
"""bench.py — reads/s searched (150 bp, k=21) against a GTDB-scale COBS index on MI355X.

One "step" = one pass of the hot path over one batch of synthetic 150-bp reads that is already resident in HBM: K1 ntHash
k-mer generation + K2 COBS query + K3 (hit list grouped by read, -T, per-query order) on the GPU, 8-byte pairs to the host, and the
expansion to finalized matches in host memory (kmcpg_finalize_grouped); the host half of step i overlaps the kernels of step i+1.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gtdb|config1|gtdb_unchunked_k31|config2_genome_search|config4_hifi|...]

N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; a plain
`python bench.py --gpus N` re-executes itself under that launcher on a free local port.  One rank per GPU; the index's independent blocks are partitioned over the ranks (libkmcpgpu shards by
bytes), every rank searches the whole batch against its blocks, the per-read hit lists are gathered on
rank 0 over RCCL and K3 runs there over the concatenation.  Total work is fixed as N grows => "scaling": "strong".
Every line carries `sanity_batch.hits_checksum` (the same at every N by construction), `ranks` (who ran, per-rank kernel times) and,
at N > 1, `parity_at_n` (the merged hit list of a sample against the CPU oracle over rows fetched from every rank).

The synthetic index is generated directly in HBM: 32 blocks x 14 976 columns
(NumRowBytes 1 872) x 968 700 rows = 58.03 GB, bits i.i.d. Bernoulli(0.30) like a Bloom filter at fpr 0.3;
90 % of the reads are 1 %-mutated, randomly reverse-complemented copies of 150-bp fragments whose k-mers
were planted into a random column, 10 % are uniform random.

The JSON line reports the GTDB-scale workload (the configuration BASELINE.json's metric is quoted on).  At N=1 the
same line carries, under "secondary", BASELINE.json configs[1] (10 k chunks, 39-byte rows; grouped and ungrouped), the
configuration the reference's own published short-read numbers are quoted on (unchunked GTDB, k = 31, -b 1024, -t 0.8),
configs[2] (genome search: FracMinHash scale 1000, 3 hashes, 50 k references, queries with ~9 relatives each in the index) and
configs[4] (HiFi ~10 kb reads sampled from planted chunks of a Closed-Syncmer index), each with roofline, CPU baseline and
same-run oracle parity.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# what random 128-byte gathers that miss L2 reach on this device (tools/ubench_cache.cpp, profiles/r02_ubench_cache.txt: 7.0-7.4 TB/s
# for working sets of 64 MiB .. 1 GiB, Infinity-Cache- or HBM-resident alike): the L2->fabric path, the kernel's practical ceiling
FABRIC_CEILING_GBS = 7200.0

WORKLOADS = {
    # GTDB r202 k=21 x10 chunks: 58.03 GB in 32 blocks (docs/database-time-and-mem-v2021.12.md:20-36)
    # every block has its own NumSigs, as in a real database (blocks hold genomes of ascending size): 967 708 + 64 i rows
    "gtdb": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=14976, num_sigs=967708, sigs_step=64, kmers_per_col=345510,
                 batch_reads=524288, kernel="k2_cobs<64,8,false>",
                 name="gtdb-scale synthetic: 32 blocks x 14976 cols x 968700 sigs (58.03 GB), 150bp k=21"),
    # what ONE of 8 ranks holds of the GTDB-scale index (4 of the 32 blocks): per-GPU kernel behaviour at N = 8 on one GPU
    "gtdb_eighth": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=4, cols_per_block=14976, num_sigs=967708, sigs_step=64, kmers_per_col=345510,
                        batch_reads=524288, kernel="k2_cobs<64,8,false>",
                        name="one eighth of the gtdb-scale synthetic index: 4 blocks x 14976 cols (7.25 GB), 150bp k=21"),
    # ... and what one of 2 / 4 ranks holds (16 / 8 blocks): the per-GPU step at N = 2 and N = 4, measured on one GPU
    "gtdb_half": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=16, cols_per_block=14976, num_sigs=967708, sigs_step=64, kmers_per_col=345510,
                      batch_reads=524288, kernel="k2_cobs<64,8,false>",
                      name="one half of the gtdb-scale synthetic index: 16 blocks x 14976 cols (29 GB), 150bp k=21"),
    "gtdb_quarter": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=8, cols_per_block=14976, num_sigs=967708, sigs_step=64, kmers_per_col=345510,
                         batch_reads=524288, kernel="k2_cobs<64,8,false>",
                         name="one quarter of the gtdb-scale synthetic index: 8 blocks x 14976 cols (14.5 GB), 150bp k=21"),
    # 10 k chunks, `kmcp index -j 32`: 32 blocks x 312 columns, 39-byte rows (BASELINE.json configs[1])
    # (equal-length chunks => the same NumSigs in every block: libkmcpgpu lays them side by side, one 1248-byte gather per k-mer)
    "config1": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=312, num_sigs=1121470, kmers_per_col=400000,
                    batch_reads=1048576, kernel="k2_cobs<64,8,false> + k2_cobs<16,8,false> (grouped) / k2_cobs<4,8,false> (KMCPG_FUSE=0)",
                    name="10k-chunk synthetic: 32 blocks x 312 cols x 1121470 sigs (1.4 GB), 150bp k=21"),
    # the same 9 984 columns indexed as ONE block (`kmcp index -b 9984`): 130 gathers of 1 248 B per read instead of 4 160 of 39 B
    "config1_wide": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=9984, num_sigs=1121470, kmers_per_col=400000,
                         batch_reads=1048576, kernel="k2_cobs<64,8,false>",
                         name="10k-chunk synthetic as one block: 1 x 9984 cols x 1121470 sigs (1.4 GB), 150bp k=21"),
    # The reference's own published short-read benchmark (benchmarks/searching/README.md:38-63, 180-229): GTDB r202 representatives
    # UNCHUNKED (47 894 genomes), `kmcp compute -k 31`, `kmcp index -f 0.3 -n 1 -b 1024` => 47 blocks x 1024 columns (128-byte rows),
    # 55.15 GB; 150-bp reads searched with `-t 0.8`: 1.14-1.41 M reads in 53.4-72.8 s on 40 threads = 18.9-21.3 k reads/s.  Blocks
    # hold genomes of ascending size, so every block has its own NumSigs: 3.0 M + 267 k i rows (0.5-5.5 M k-mers per genome at
    # fpr 0.3), 55.2 GB in total.
    "gtdb_unchunked_k31": dict(k=31, num_hashes=1, fpr=0.3, n_blocks=47, cols_per_block=1024, num_sigs=3000000, sigs_step=267000,
                               kmers_per_col=3200000, batch_reads=1048576, kernel="k2_cobs<8,8,false>", min_qcov=0.8,
                               genome_query=dict(n=4, min_len=4600000, max_len=5600000, min_qcov=0.5),
                               metric="reads/sec searched (150bp, k=31, -t 0.8) vs the unchunked GTDB index of the reference's published benchmark",
                               name="gtdb r202 unchunked synthetic: 47 blocks x 1024 cols, 3.0-15.3 M sigs (55.2 GB), 150bp k=31, -t 0.8"),
    # BASELINE.json configs[2] — genome search (reference: benchmarks/searching/README.md:382-432): whole
    # assemblies as queries (`kmcp search -g --sort-by jacc -t 0.4`) against a FracMinHash (scale 1000) index of 50 048 references
    # with 3 hash functions at fpr 0.001, `index -j 8` => 8 blocks x 6 256 columns (782-byte rows).  The references come in
    # families: every query genome has `relatives` mutated copies in the index (substitution rate 0, 0.5 %, ... 4.5 %: the
    # shared-21-mer fraction falls from 1 to 0.38, so ~9 of the 10 pass -t 0.4 — with decreasing qCov/jacc, the threshold
    # cutting through the family); the query itself is a 0.5 %-mutated, randomly reverse-complemented copy; 10 % of the queries
    # are unrelated genomes.  ~8 000 sketch k-mers per query: the sort+unique path and 16 counter planes.
    "config2_genome_search": dict(k=21, num_hashes=3, fpr=0.001, n_blocks=8, cols_per_block=6256, num_sigs=431000, sigs_step=13, kmers_per_col=10000,
                                  scale=1000, batch_reads=256, read_len=4000000, relatives=10, rel_step=0.005, sub_rate=0.005, distinct_batches=2,
                                  # (one assembly per call: the reference's published genome-search case, 0.53-0.62 s hot on 8 threads, benchmarks/searching/README.md:382-432)
                                  genome_query=dict(n=4, min_len=3600000, max_len=5600000, min_qcov=0.4, reference_s=[0.53, 0.62]),
                                  min_qcov=0.4, sort_by=2, unit="queries/s", cpu_sample_start=16, kernel="k2_cobs<64,16,true,false,8> (k1_kmers_wg<0> + k1_seg_hash beside it; batches below ~190 genomes take the chunked form <64,16,true,true,8>)",
                                  metric="genomes/sec searched (4-Mbp assemblies, FracMinHash scale 1000, k=21, 3 hashes, -t 0.4) vs a 50 k-reference index",
                                  name="genome search, synthetic: 8 blocks x 6256 cols x 431 k sigs (2.7 GB), 3 hashes, fpr 0.001, scale 1000; "
                                       "queries = 4-Mbp genomes with 10 relatives each in the index"),
    # BASELINE.json configs[4] — HiFi long reads (reference: benchmarks/mock-hifi-zymo/README.md:52-53): reads
    # of ~N(10 kb, 2 kb) with 0.1 % errors sampled from the chunks of a Closed-Syncmer (k=21, s=11) 10 k-chunk database.  A
    # database built by `kmcp compute -S 11` + `kmcp index -j 32` has a NumSigs of its own in every block (the number of syncmers
    # differs from chunk to chunk, the block takes the maximum: index.go:936-946) => sigs_step > 0, 39-byte rows, nothing grouped.
    "config4_hifi": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=312, num_sigs=300000, sigs_step=7, kmers_per_col=100000, syncmer_s=11,
                         batch_reads=16384, read_len=("normal", 10000, 2000, 2000, 20000), sub_rate=0.001, unit="reads/s", cpu_sample_start=64,
                         kernel="k2_cobs<4,16,false,false,8> (k1_windows_wave<2> beside it)",
                         metric="reads/sec searched (HiFi ~10 kb, Closed Syncmer s=11, k=21) vs a 10k-chunk index",
                         name="HiFi synthetic: 32 blocks x 312 cols x ~300 k sigs (0.37 GB), closed syncmer s=11 k=21, reads ~N(10 kb, 2 kb) 0.1 % errors"),
    # ... and the same database built with kmcpg_build_cfg.uniform_sigs = 1 (one NumSigs for all blocks: one 1248-byte gather per k-mer)
    "config4_hifi_uniform_sigs": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=312, num_sigs=300000, sigs_step=0, kmers_per_col=100000,
                                      syncmer_s=11, batch_reads=16384, read_len=("normal", 10000, 2000, 2000, 20000), sub_rate=0.001, unit="reads/s",
                                      cpu_sample_start=64, kernel="k2_cobs<64,16,false,false,8> + k2_cobs<16,16,false,false,8> (k1_windows_wave<2> beside them)",
                                      metric="reads/sec searched (HiFi ~10 kb, Closed Syncmer s=11, k=21) vs a 10k-chunk index with one NumSigs",
                                      name="HiFi synthetic, blocks with equal NumSigs (grouped rows): 32 x 312 cols x 300 k sigs, closed syncmer s=11 k=21"),
    # a mid-sized database as `kmcp index -j 32` cuts it: 100 000 chunks -> 32 blocks x 3 125 columns = 391-byte rows (round 5: the 32-lane form)
    "mid_rows": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=3125, num_sigs=1121470, sigs_step=64, kmers_per_col=400000,
                     batch_reads=1048576, kernel="k2_cobs<32,8,false,false,4>",
                     metric="reads/sec searched (150bp, k=21) vs a 100k-chunk index of 391-byte rows",
                     name="100k-chunk synthetic: 32 blocks x 3125 cols x ~1.12 M sigs (14 GB), 150bp k=21"),
    # ... and a 200 000-chunk one: 32 blocks x 6 250 columns = 782-byte rows (832-byte pitch), single hash, short reads
    "mid_rows_782": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=6250, num_sigs=1121470, sigs_step=64, kmers_per_col=400000,
                         batch_reads=1048576, kernel="k2_cobs<64,8,false,false,4>",
                         metric="reads/sec searched (150bp, k=21) vs a 200k-chunk index of 782-byte rows",
                         name="200k-chunk synthetic: 32 blocks x 6250 cols x ~1.12 M sigs (28 GB), 150bp k=21"),
    # ... a 150 000-chunk one: 32 blocks x 4 688 columns = 586-byte rows (640-byte pitch): the 64-lane form with 40 of 64 lanes busy, or — with
    # KMCPG_SPLIT_TILES=2 — a 512-byte tile on the 32-lane form + a 128-byte tile on the 8-lane form (profiles/r06_lpr_640.txt)
    "mid_rows_586": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=4688, num_sigs=1121470, sigs_step=64, kmers_per_col=400000,
                         batch_reads=1048576, kernel="k2_cobs<64,8,false,false,4>",
                         metric="reads/sec searched (150bp, k=21) vs a 150k-chunk index of 586-byte rows",
                         name="150k-chunk synthetic: 32 blocks x 4688 cols x ~1.12 M sigs (23 GB), 150bp k=21"),
    "mid_rows_575": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=4600, num_sigs=1121470, sigs_step=64, kmers_per_col=400000,
                         batch_reads=1048576, kernel="k2_cobs<64,8,false,false,4>", metric="reads/sec searched (150bp, k=21) vs an index of 575-byte rows",
                         name="synthetic: 32 blocks x 4600 cols x ~1.12 M sigs (23 GB), 575-byte rows (576-byte pitch), 150bp k=21"),
    "mid_rows_750": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=32, cols_per_block=6000, num_sigs=1121470, sigs_step=64, kmers_per_col=400000,
                         batch_reads=1048576, kernel="k2_cobs<64,8,false,false,4>", metric="reads/sec searched (150bp, k=21) vs an index of 750-byte rows",
                         name="synthetic: 32 blocks x 6000 cols x ~1.12 M sigs (27 GB), 750-byte rows (768-byte pitch), 150bp k=21"),
    # EXPERIMENT (VERDICT r4 #3 gate, profiles/r05_rowsort_gate.txt): ONE narrow block of the HiFi index and enough reads to fill the
    # chip with (read, block) units; KMCPG_DEBUG_ROWSORT=1|2 re-orders every read's k-mers by the row they address
    "config4_oneblock": dict(k=21, num_hashes=1, fpr=0.3, n_blocks=1, cols_per_block=312, num_sigs=300000, sigs_step=0, kmers_per_col=100000, syncmer_s=11,
                             batch_reads=131072, read_len=("normal", 10000, 2000, 2000, 20000), sub_rate=0.001, unit="reads/s", cpu_sample_start=64,
                             distinct_batches=1, kernel="k2_cobs<4,16,false,false,8>",
                             metric="reads/sec searched (HiFi ~10 kb, Closed Syncmer s=11, k=21) vs ONE 312-column block (experiment)",
                             name="HiFi synthetic, one block: 1 x 312 cols x 300 k sigs (19 MB), closed syncmer s=11 k=21"),
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


class Batch:
    """One batch of synthetic queries resident in HBM: `reads` (uint8 ASCII), CSR `offs` (int64[n+1]), the column each query was
    sampled from (`cols`, -1 = unrelated), total bases and the longest query."""
    pass


def _read_lengths(spec, n, seed):
    if isinstance(spec, int):
        return torch.full((n,), spec, dtype=torch.int64)
    kind, mean, sd, lo, hi = spec
    assert kind == "normal"
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.clamp((torch.randn(n, generator=g) * sd + mean).long(), lo, hi)


def _mutate(code, rate, g, chunk=1 << 28):
    """substitutions at `rate` per base (the substitute is uniform over ACGT, as in SURVEY.md 8d's read models)"""
    if rate <= 0:
        return code
    out = code.clone()
    for a in range(0, code.numel(), chunk):  # chunked: a 4-Mbp-genome batch is 0.5 G bases
        v = out[a:a + chunk]
        sub = torch.rand(v.numel(), generator=g, device=v.device) < rate
        v[sub] = torch.randint(0, 4, (int(sub.sum().item()),), generator=g, device=v.device, dtype=torch.uint8)
    return out


def make_batch(dev, wl, n_reads, n_cols, seed, plant):
    """Builds one batch and plants what its queries were sampled from: plant(ascii, offs, n, total, maxlen, cols) ORs the Bloom
    bits of every (sketched) k-mer of fragment i into column cols[i] (kmcpg_plant_reads_device: same K1 as a query, so the index
    holds exactly what `kmcp compute` + `kmcp index` would have put there for these sequences).  With `relatives` = R each
    fragment is planted R times, as copies with substitution rates 0, rel_step, 2 rel_step ... into R different columns."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    lens = _read_lengths(wl.get("read_len", READ_LEN), n_reads, seed).to(dev)
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(lens, 0)
    total, maxlen = int(offs[-1].item()), int(lens.max().item())
    code = torch.randint(0, 4, (total,), generator=g, device=dev, dtype=torch.uint8)
    cols = torch.randint(0, n_cols, (n_reads,), generator=g, device=dev).to(torch.int32)
    is_random = torch.rand(n_reads, generator=g, device=dev) < float(os.environ.get("KMCP_BENCH_RANDOM_FRAC", "0.10"))  # (experiments: 1 = no read has a home)
    cols[is_random] = -1  # 0xFFFFFFFF: not planted
    R = int(wl.get("relatives", 1))
    for r in range(R):
        rel = _mutate(code, r * wl.get("rel_step", 0.0), g)
        # relative r of a query lives 7919 r columns further (another block for most of them)
        cr = torch.where(cols >= 0, (cols.to(torch.int64) + 7919 * r) % n_cols, torch.full_like(cols, -1, dtype=torch.int64)).to(torch.int32).contiguous()
        frag = acgt[rel.long()] if rel.numel() < (1 << 28) else torch.cat([acgt[rel[a:a + (1 << 28)].long()] for a in range(0, rel.numel(), 1 << 28)])
        plant(frag, offs, n_reads, total, maxlen, cr)
        torch.cuda.synchronize()
        del rel, frag
    # the query: substitutions, then half of them reverse-complemented (A<->T, C<->G = 3 - code)
    q = _mutate(code, wl.get("sub_rate", 0.01), g)
    rc = torch.rand(n_reads, generator=g, device=dev) < 0.5
    if isinstance(wl.get("read_len", READ_LEN), int) and maxlen <= 4096:
        q2 = q.view(n_reads, maxlen)
        q = torch.where(rc[:, None], 3 - q2.flip(1), q2).contiguous().view(-1)
    else:
        # ragged reads: position p of a reverse-complemented read i takes 3 - base at offs[i] + offs[i+1] - 1 - p (one gather per
        # 2^27 positions; a loop over the reads would be tens of thousands of tiny launches)
        out = torch.empty_like(q)
        for a in range(0, total, 1 << 27):
            b = min(total, a + (1 << 27))
            pos = torch.arange(a, b, device=dev, dtype=torch.int64)
            rid = torch.searchsorted(offs, pos, right=True) - 1
            flip = rc[rid]
            src = torch.where(flip, offs[rid] + offs[rid + 1] - 1 - pos, pos)
            v = q[src]
            out[a:b] = torch.where(flip, 3 - v, v)
            del pos, rid, flip, src, v
        q = out
    bt = Batch()
    bt.reads = (acgt[q.long()] if q.numel() < (1 << 28) else torch.cat([acgt[q[a:a + (1 << 28)].long()] for a in range(0, q.numel(), 1 << 28)])).contiguous()
    bt.offs, bt.cols, bt.total, bt.maxlen, bt.n = offs.contiguous(), cols.contiguous(), total, maxlen, n_reads
    return bt


_D2D = {}


def d2d_copy_gbps(dev):
    """device-to-device copy rate of this box in GB/s (bytes read + bytes written over time), measured once per process"""
    if dev in _D2D:
        return _D2D[dev]
    try:
        n = 2 << 30
        a = torch.empty(n, dtype=torch.uint8, device=dev)
        b = torch.empty(n, dtype=torch.uint8, device=dev)
        a.zero_()
        b.copy_(a)
        torch.cuda.synchronize()
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        torch.cuda.empty_cache()
        _D2D[dev] = best
    except Exception:
        _D2D[dev] = None
    return _D2D[dev]


class Ctx:
    pass


def host_memory_available():
    """Bytes of host RAM this process may still take (cgroup limit honoured)."""
    import psutil
    avail = psutil.virtual_memory().available
    for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(f).read().strip()
            if v != "max":
                used = 0
                for u in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                    try:
                        used = int(open(u).read())
                        break
                    except Exception:
                        pass
                avail = min(avail, int(v) - used)
            break
        except Exception:
            continue
    return avail


def run_workload(name, ctx, steps, warmup, batch_reads=0, cpu_baseline=True, cpu_target_s=8.0, cpu_sample_reads=0, extras=True):
    """Builds the synthetic index of workload `name` in HBM, times `steps` steps, returns the result dict (rank 0).

    One step = reads resident in HBM -> K1 + K2 on this rank's blocks -> hit lists to rank 0 (RCCL when N > 1) -> D2H ->
    kmcpg_finalize on the host (float64 thresholds, FPR, sort): finalized matches in host memory.  The host half of step i runs
    while the GPU works on step i+1 (it is inside the timed region for every one of the K steps)."""
    from kmcp_amd import Database, default_params, lib
    from kmcp_amd.dist import gather_hits, hits_checksum

    world, rank, dev, dev_index = ctx.world, ctx.rank, ctx.dev, ctx.dev_index
    coll = ctx.collective  # the N > 1 code path (also taken by a one-rank RCCL group under KMCP_BENCH_FORCE_DIST=1: tests)
    wl = dict(WORKLOADS[name])
    B = batch_reads or wl["batch_reads"]
    spec = lib.SynthSpec(k=wl["k"], num_hashes=wl["num_hashes"], fpr=wl["fpr"], n_blocks=wl["n_blocks"],
                         cols_per_block=wl["cols_per_block"], num_sigs=wl["num_sigs"], kmers_per_col=wl["kmers_per_col"], seed=42,
                         sigs_step=wl.get("sigs_step", 0), scale=wl.get("scale", 0), syncmer_s=wl.get("syncmer_s", 0),
                         minimizer_w=wl.get("minimizer_w", 0))
    free_b, _ = torch.cuda.mem_get_info(dev)
    need = wl["n_blocks"] * (wl["num_sigs"] + wl.get("sigs_step", 0) * (wl["n_blocks"] - 1) / 2) * ((wl["cols_per_block"] + 7) // 8 + 64) / world
    if need > 0.9 * free_b:
        raise SystemExit(f"workload needs {need/1e9:.1f} GB of HBM on this rank, {free_b/1e9:.1f} GB free")
    t0 = time.time()
    db = Database.open_synthetic(spec, device=dev_index, shard_rank=rank, shard_count=world)
    torch.cuda.synchronize()
    info = db.info
    n_cols = int(info.n_cols)
    params = default_params()  # kmcp search defaults: -t 0.55 -c 10 -m 30 -f 0.01 -u 256
    params.min_qcov = wl.get("min_qcov", params.min_qcov)
    params.sort_by = wl.get("sort_by", 0)
    db.set_profiling(True)
    unit = wl.get("unit", "reads/s")

    # ---- batches resident in HBM; distinct data per step (cycled if K+W is large)
    n_batches = max(1, min(steps + warmup, wl.get("distinct_batches", 4)))

    def plant(frag, offs, n, total, maxlen, cols):
        db.plant_reads_device(frag.data_ptr(), offs.data_ptr(), n, total, maxlen, cols.data_ptr())

    batches = [make_batch(dev, wl, B, n_cols, 1000 + i, plant) for i in range(n_batches)]
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    cap = (4 + 2 * int(wl.get("relatives", 1))) * B + 4096
    main = torch.cuda.current_stream(dev)
    # D2H (and, N > 1, the exchange) of a finished step while the next step's kernels run on `main`: a high-priority stream, so
    # that its small copies / collectives are not queued behind a 60-500 ms kernel that fills the chip
    # (one side stream and one second kernel stream per process, not per workload: torch never gives a stream back, HIP multiplexes all
    # live streams onto GPU_MAX_HW_QUEUES hardware queues, and a kernel stream that shares its queue with another one serialises behind it —
    # the genome search as the fifth workload of a run lost its whole two-stream gain that way, profiles/r06_cobs_overlap.txt)
    if getattr(ctx, "side_stream", None) is None:
        ctx.side_stream = torch.cuda.Stream(dev, priority=-1)
    side = ctx.side_stream
    # Two kernel streams taken in turn by consecutive steps, so that K1 of step i + 1 (VALU-bound) runs beside K2 of step i (memory-bound); the
    # K2s themselves follow each other (query.cpp cobs_ev).  Default: for batches of WHOLE GENOMES only — what the library does for its own
    # lanes and workspace slots (engine.hpp whole_genome_batch: plain / FracMinHash k-mers, single-end, a query above 65 536 bases): their K1 is
    # one fat kernel that fits beside the COBS kernel (genome search +9 %, profiles/r06_cobs_overlap.txt); every other shape loses 3-11 % to the
    # cross-stream waits (profiles/r05_k1_beside_k2.txt).  KMCP_BENCH_STREAMS=1 / 2 forces one / two (with KMCPG_WS_SLOTS to match).
    whole_genomes = not wl.get("syncmer_s") and not wl.get("minimizer_w") and max(b.maxlen for b in batches) > 65536
    two_streams = os.environ.get("KMCP_BENCH_STREAMS", "2" if whole_genomes else "1") == "2"
    if two_streams and getattr(ctx, "kernel_stream2", None) is None:
        ctx.kernel_stream2 = torch.cuda.Stream(dev)
    kstreams = [main, ctx.kernel_stream2] if two_streams else [main, main]
    poll = os.environ.get("KMCP_BENCH_POLL") == "1"  # experiment: busy-poll hipEventQuery instead of hipEventSynchronize
    # experiment (profiles/r05_restart_stall.txt): wait for a step by polling a word of PINNED memory that the step's last copy writes (its
    # sequence number), i.e. without asking the runtime whether an event has completed
    flag_wait = os.environ.get("KMCP_BENCH_FLAG") == "1"
    seq_no = [0]

    class Buf:  # device outputs of one step in flight + their pinned host copies
        def __init__(self):
            self.d_hits = torch.empty((cap, 3), dtype=torch.int32, device=dev)
            self.d_cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            self.d_qk = torch.zeros(B, dtype=torch.int32, device=dev)
            self.d_ql = torch.zeros(B, dtype=torch.int32, device=dev)
            self.h_cnt = torch.zeros(2, dtype=torch.int64).pin_memory()
            self.d_seq = torch.zeros(1, dtype=torch.int64, device=dev)
            self.h_seq = torch.zeros(1, dtype=torch.int64).pin_memory()
            self.want_seq = 0
            self.h_hits = torch.empty((cap * world, 3), dtype=torch.int32).pin_memory()
            self.h_qk = torch.empty(B, dtype=torch.int32).pin_memory()
            self.h_ql = torch.empty(B, dtype=torch.int32).pin_memory()
            # K3 (device half of finalize): matches grouped by read, filtered by -T, in final order — 8 bytes each + the reads' offsets
            self.d_pairs = torch.empty((cap * (world if rank == 0 else 1), 2), dtype=torch.int32, device=dev)
            self.d_roffs = torch.zeros(B + 2, dtype=torch.int64, device=dev)
            self.d_cat = torch.empty((cap * world, 3), dtype=torch.int32, device=dev) if (coll and rank == 0) else None
            self.d_ncat = torch.zeros(1, dtype=torch.int64, device=dev)
            self.h_pairs = torch.empty((cap * world, 2), dtype=torch.int32).pin_memory()
            self.h_roffs = torch.zeros(B + 2, dtype=torch.int64).pin_memory()
            self.k3_start = torch.cuda.Event(enable_timing=True)
            self.k3_end = torch.cuda.Event(enable_timing=True)
            # (experiment KMCP_BENCH_EVT_TIMING=1: a timing-enabled event, profiles/r05_restart_stall.txt)
            self.kernels_done = torch.cuda.Event(enable_timing=os.environ.get("KMCP_BENCH_EVT_TIMING") == "1")
            self.copied = torch.cuda.Event()
            self.used = False
            self.grouped = False

    bufs = [Buf(), Buf()]  # two steps in flight, with and without a collective
    for b_, st_ in zip(bufs, kstreams):
        b_.stream = st_

    use_k3 = os.environ.get("KMCP_BENCH_K3", "1") != "0"  # 0: the round-3 host half (kmcpg_finalize on the raw hit list)
    k3_ms = []

    def gpu_half(i, bf, raw=False):
        """Enqueues K1 + K2 of batch i on this rank's blocks and, at N = 1, K3 behind them (nothing waits here).
        raw: leave the hit list as K2 emitted it (the sanity step: checksum and oracle diff work on the raw tuples)."""
        bt = batches[i % n_batches]
        ks = bf.stream
        if bf.used:
            ks.wait_event(bf.copied)  # the previous step that used these buffers has left them
        with torch.cuda.stream(ks):
            db.query_device(bt.reads.data_ptr(), bt.offs.data_ptr(), B, bt.total, bt.maxlen, bf.d_hits.data_ptr(), cap, bf.d_cnt.data_ptr(),
                            bf.d_qk.data_ptr(), bf.d_ql.data_ptr(), params=params, stream=ks.cuda_stream)
            bf.h_cnt.copy_(bf.d_cnt, non_blocking=True)
            bf.grouped = use_k3 and not raw
            if bf.grouped and not coll:
                bf.k3_start.record(ks)
                db.group_device(bf.d_hits.data_ptr(), bf.d_cnt.data_ptr(), cap, bf.d_qk.data_ptr(), B, bf.d_pairs.data_ptr(), bf.d_roffs.data_ptr(),
                                params=params, stream=ks.cuda_stream)
                bf.k3_end.record(ks)
                bf.h_roffs.copy_(bf.d_roffs, non_blocking=True)
            if flag_wait:
                seq_no[0] += 1
                bf.want_seq = seq_no[0]
                bf.d_seq.fill_(bf.want_seq)
                bf.h_seq.copy_(bf.d_seq, non_blocking=True)  # the step's last command: the word arrives when everything before it is done
            bf.kernels_done.record(ks)
        bf.used = True

    def exchange(bf):
        """Waits for the step's kernels; hit lists to rank 0 (RCCL when N > 1) and on their way to pinned host memory
        (`copied` fires when the host may read).  Returns #hits on rank 0.  Holds collectives when N > 1."""
        if not coll:
            if flag_wait:
                while int(bf.h_seq[0]) != bf.want_seq:
                    pass
            elif poll:
                while not bf.kernels_done.query():
                    pass
            else:
                bf.kernels_done.synchronize()
            n = int(bf.h_cnt[0])
            assert n <= cap, "hit buffer overflow"
            side.wait_event(bf.kernels_done)
            with torch.cuda.stream(side):
                if bf.grouped:
                    kept = int(bf.h_roffs[B])
                    bf.h_pairs[:kept].copy_(bf.d_pairs[:kept], non_blocking=True)
                else:
                    bf.h_hits[:n].copy_(bf.d_hits[:n], non_blocking=True)
                bf.h_qk.copy_(bf.d_qk, non_blocking=True)
                bf.h_ql.copy_(bf.d_ql, non_blocking=True)
                bf.copied.record(side)
            return n
        # N > 1: the exchange of step i runs on the side stream, behind step i's kernels only — the kernels of step i+1 are already
        # enqueued on `main` and keep the GPU busy while the counts are all-gathered, the hit buffers gathered (RCCL over xGMI)
        # and rank 0 copies the merged list to pinned host memory
        side.wait_event(bf.kernels_done)
        n = 0
        with torch.cuda.stream(side):
            parts = gather_hits(bf.d_hits, bf.d_cnt[:1], dst=0, force_collectives=True)  # all_gather(counts) + gather(hit buffers)
            if rank == 0:
                if bf.grouped:
                    # the shards' lists side by side on this GPU, then K3 over the concatenation: grouped, filtered, ordered here,
                    # 8 bytes per match to the host
                    for part in parts:
                        c = part.shape[0]
                        bf.d_cat[n:n + c].copy_(part, non_blocking=True)
                        n += c
                    bf.d_ncat.fill_(n)
                    bf.k3_start.record(side)
                    db.group_device(bf.d_cat.data_ptr(), bf.d_ncat.data_ptr(), cap * world, bf.d_qk.data_ptr(), B, bf.d_pairs.data_ptr(),
                                    bf.d_roffs.data_ptr(), params=params, stream=side.cuda_stream)
                    bf.k3_end.record(side)
                    bf.h_roffs.copy_(bf.d_roffs, non_blocking=True)
                    bf.h_pairs[:n].copy_(bf.d_pairs[:n], non_blocking=True)  # (at most n survive -T; h_roffs says how many)
                else:
                    for part in parts:
                        c = part.shape[0]
                        bf.h_hits[n:n + c].copy_(part, non_blocking=True)
                        n += c
                bf.h_qk.copy_(bf.d_qk, non_blocking=True)
                bf.h_ql.copy_(bf.d_ql, non_blocking=True)
            bf.copied.record(side)
        return n

    def host_half(bf, n):
        """kmcpg_finalize on rank 0: float64 thresholds, FPR, Match values, sort -> finalized matches in host memory."""
        if rank != 0:
            return 0
        bf.copied.synchronize()
        if bf.grouped:
            k3_ms.append(bf.k3_start.elapsed_time(bf.k3_end))
            kept = int(bf.h_roffs[B])
            return db.finalize_grouped(bf.h_pairs[:kept].numpy(), bf.h_roffs.numpy(), bf.h_qk.numpy(), bf.h_ql.numpy(), params=params, count_only=True)
        return db.finalize_count(bf.h_hits[:n].numpy(), bf.h_qk.numpy(), bf.h_ql.numpy(), params=params)

    def barrier():
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(first, count, with_host=True, times=None):
        """`count` steps starting at batch `first`, two in flight: the kernels of step i+1 are enqueued before step i's hits are
        exchanged (N > 1: RCCL on the side stream), fetched and finalized.  Returns (#hits, #matches) on rank 0."""
        hits = matches = 0
        if count <= 0:
            return hits, matches
        gpu_half(first, bufs[0])
        for j in range(count):
            bf = bufs[j % 2]
            t0 = time.perf_counter()
            if j + 1 < count:
                gpu_half(first + j + 1, bufs[(j + 1) % 2])
            ta = time.perf_counter()
            n = exchange(bf)
            tb = time.perf_counter()
            hits += n
            if with_host:
                matches += host_half(bf, n)
            tc = time.perf_counter()
            if times is not None:
                times.append(db.last_timing(age=1 if j + 1 < count else 0))
            if os.environ.get("KMCP_BENCH_TRACE"):
                print(f"rank {rank} step {j}: enqueue of the next {1e3*(ta-t0):.2f} ms, exchange {1e3*(tb-ta):.2f} ms, host half {1e3*(tc-tb):.2f} ms, timing {1e3*(time.perf_counter()-tc):.2f} ms", file=sys.stderr)
        return hits, matches

    run_steps(0, warmup)
    barrier()
    times = []
    t_start = time.perf_counter()
    n_hits_total, n_matches_total = run_steps(warmup, steps, times=times)
    barrier()
    elapsed = time.perf_counter() - t_start
    k1_ms = [t_[0] for t_ in times]
    k2_ms = [t_[1] for t_ in times]
    if coll:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if ctx.same_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (k2_cobs) on this rank.
    #  algorithmic bytes per launch: sum over reads of kept k-mers x sum over local blocks of numHashes x NumRowBytes,
    #  + qLen, + 12 B per hit.  The kernel does NOT move all of them: exact sector pruning stops loading rows whose columns cannot
    #  reach the threshold any more, so algorithmic bytes / time can exceed the chip's peak and is reported as `algorithmic_gbps`.
    #  `achieved` / `frac` are what the kernel actually moved, measured in this run (below: the kernel counts its own row loads
    #  on the very batches of the timed steps; the committed --pmc FETCH_SIZE pass of the same command is carried beside it).
    last = (warmup + steps - 1) % n_batches
    qk = bufs[(steps - 1) % len(bufs)].d_qk.cpu().numpy().astype(np.int64)
    kmers_per_launch = int(qk.sum())
    bases_per_launch = float(np.mean([batches[(warmup + j) % n_batches].total for j in range(steps)]))
    alg_bytes = kmers_per_launch * int(info.row_bytes_sum_local) * int(info.num_hashes) + int(bases_per_launch) + 12 * (n_hits_total // max(1, steps))
    k2_avg_ms = float(np.mean(k2_ms))
    effective = alg_bytes / (k2_avg_ms * 1e-3) / 1e9
    pmc, pmc_src = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            key = f"{name}{'_ungrouped' if os.environ.get('KMCPG_FUSE') == '0' else ''}:{B}:{world}"
            if key in tj:
                pmc = tj[key]["hbm_bytes_per_launch"]
                pmc_src = tj[key].get("source")
        except Exception:
            pmc = None

    # DRAM-side vs Infinity-Cache-side share of the fabric traffic, from the committed counter passes of the same command
    # (profiles/traffic.json "mall": {...})
    mall = None
    try:
        mall = json.load(open(tfile)).get("mall", {}).get(f"{name}{'_ungrouped' if os.environ.get('KMCPG_FUSE') == '0' else ''}:{B}:{world}")
    except Exception:
        mall = None

    out = {
        "metric": wl.get("metric", "reads/sec searched (150bp, k=21) vs GTDB-scale index"),
        "value": B * steps / elapsed,
        "unit": unit,
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32 bitwise (bit-sliced counters), u64 hashes",
        "data": "synthetic",
        "value_definition": "queries resident in HBM -> finalized (target, mKmers, qCov, tCov, jacc, FPR) tuples in host memory (the bench contract: "
                            "inputs are in HBM when the timed region starts; the host half of a step overlaps the next step's kernels).  SURVEY.md 8(d)'s "
                            "number - batch bytes in HOST memory -> hit tuples in HOST memory, PCIe both ways included - is `value_host_to_host` "
                            "(details under host_boundary); it is measured in the same run at N = 1",
        "value_host_to_host": None,
        "config": {"workload": wl["name"], "batch_reads": B, "read_len": wl.get("read_len", READ_LEN), "bases_per_batch": int(bases_per_launch),
                   "mean_kmers_per_query": kmers_per_launch / B, "k": wl["k"], "num_hashes": wl["num_hashes"],
                   "index_bytes": int(info.matrix_bytes), "index_bytes_this_rank": int(info.matrix_bytes_local),
                   "blocks": int(info.n_blocks), "columns": n_cols, "parallelism": f"block-shard x{world}", "kernel_streams": 2 if two_streams else 1,
                   "search_flags": f"-t {params.min_qcov:g} -c 10 -m 30 -f 0.01 -u 256 -s {('qcov', 'tcov', 'jacc')[params.sort_by]}"},
        "roofline": {"bound": "hbm", "kernel": wl["kernel"], "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                     "traffic_source": None,
                     "bound_detail": "L2->fabric read traffic (what FETCH_SIZE counts): requests that miss the XCD's L2 and are served by the Infinity "
                                     "Cache (MALL, 256 MiB) or by HBM.  `peak` is the HBM3E spec peak; `frac` therefore compares fabric-side bytes with "
                                     "the DRAM peak and part of those bytes never left the die - see mall_split for how many",
                     "definition": "achieved = bytes the kernel moved per launch (traffic) / its mean HIP-event duration over the timed steps; "
                                   "frac = achieved / peak.  algorithmic_gbps = ALGORITHMIC bytes (SURVEY 8d) / the same duration: larger than "
                                   "achieved, and possibly than peak, by what exact sector pruning never fetches",
                     "mall_split": mall,
                     "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_gbps": effective, "algorithmic_over_peak": effective / HBM_PEAK_GBS,
                     "traffic_pmc": pmc, "traffic_pmc_source": pmc_src,
                     "measured_ceiling": {"gbps": FABRIC_CEILING_GBS, "what": "random 128-B gathers that miss L2 (L2->fabric path), tools/ubench_cache.cpp",
                                          "source": "profiles/r02_ubench_cache.txt"},
                     "kernel_ms": k2_avg_ms, "kmers_kernel_ms": float(np.mean(k1_ms)),
                     "finalize_kernels_ms": (float(np.mean(k3_ms[-steps:])) if k3_ms else None)},
        "hits_per_step": n_hits_total / max(1, steps),
        "matches_per_step": n_matches_total / max(1, steps),
        "setup_s": setup_s,
    }

    # ---- sanity on the last batch: planted reads must come back with their column (the step holds collectives: every
    #      rank takes part, rank 0 evaluates the merged hit list)
    gpu_half(last, bufs[0], raw=True)
    n_last = exchange(bufs[0])
    torch.cuda.synchronize()
    hh = None
    if rank == 0:
        hh = bufs[0].h_hits[:n_last].numpy().astype(np.int64)
        cols_last = batches[last].cols.cpu().numpy().astype(np.int64)
        got = set(zip(hh[:, 0].tolist(), hh[:, 1].tolist()))
        planted = np.nonzero(cols_last >= 0)[0]
        out["planted_recall"] = sum((int(r), int(cols_last[r])) in got for r in planted[:20000]) / max(1, min(len(planted), 20000))
        # the merged hit list of this batch is a function of the seeds only: same number at every N (checked against the CPU oracle
        # below, on a sample, at every N)
        out["sanity_batch"] = {"hits": int(len(hh)), "hits_checksum": hits_checksum(hh), "batch_seed": 1000 + last,
                               "note": "order-independent 64-bit checksum of the merged (read, column, count) list of one batch: identical at N = 1, 2, 4, 8"}
    # who ran: world size, backend, and every rank's own kernel time / traffic (the slowest rank sets the step)
    out["ranks"] = {"world_size": dist.get_world_size() if coll else 1, "backend": dist.get_backend() if coll else None,
                    "exchange": ("torch.distributed all_gather_into_tensor(counts) + gather(hit buffers) on a side stream" if coll else "none (one rank)")}
    mine = {"rank": rank, "device": torch.cuda.get_device_name(dev), "blocks": int(info.n_blocks_local), "index_bytes": int(info.matrix_bytes_local),
            "k1_ms": float(np.mean(k1_ms)), "k2_ms": k2_avg_ms, "hits_last_batch": int(bufs[0].h_cnt[0])}
    if coll:
        per = [None] * dist.get_world_size()
        dist.all_gather_object(per, mine)
    else:
        per = [mine]
    out["ranks"]["per_rank"] = per
    out["ranks"]["k2_ms_min_max"] = [min(p_["k2_ms"] for p_ in per), max(p_["k2_ms"] for p_ in per)]

    # ---- what the kernel moved: every distinct batch of the timed steps once more with the kernel counting its own row loads
    #      (profiling level 2: one atomic per wave and row group, 16 B per lane and row actually loaded, row padding included,
    #      pruned rows not; the counts do not depend on timing, so this is the traffic of the timed launches themselves)
    def measure_gathered(env=None):
        old = {}
        for k_, v_ in (env or {}).items():
            old[k_] = os.environ.get(k_)
            os.environ[k_] = v_
        try:
            db.set_profiling(2)
            per_batch = []
            for i in range(n_batches):
                gpu_half(i, bufs[0])
                torch.cuda.synchronize()
                bufs[0].copied.record(bufs[0].stream)
                per_batch.append((db.last_gathered_bytes(), db.last_hash_bytes()))
        finally:
            db.set_profiling(True)
            for k_, v_ in old.items():
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_
        return per_batch

    rf = out["roofline"]
    per_batch = measure_gathered()
    row_b = float(np.mean([per_batch[(warmup + j) % n_batches][0] for j in range(steps)]))
    hash_b = float(np.mean([per_batch[(warmup + j) % n_batches][1] for j in range(steps)]))
    gathered = row_b + hash_b
    rf["gathered_bytes_per_launch"] = gathered
    rf["row_bytes_per_launch"] = row_b
    rf["hash_bytes_per_launch"] = hash_b
    rf["traffic"] = gathered
    rf["traffic_source"] = ("live: bytes the k2_cobs launches of the timed steps asked the memory system for, counted by the kernel itself: its 16-byte "
                            "row loads (kmcpg_last_gathered_bytes) + the 8-byte k-mer hashes it reads once per (read, slot) (kmcpg_last_hash_bytes); "
                            "cross-check = traffic_pmc (rocprofv3 --pmc FETCH_SIZE pass of the same command)")
    rf["achieved"] = gathered / (k2_avg_ms * 1e-3) / 1e9
    rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
    rf["traffic_over_algorithmic"] = gathered / alg_bytes
    rf["measured_ceiling"]["achieved_over_it"] = rf["achieved"] / FABRIC_CEILING_GBS
    # the device-to-device copy rate of THIS box beside the vendor peak (a streaming figure: half of the bytes are
    # writes, all of them sequential; the kernel's gathers are reads of 1 KB at random rows)
    d2d = d2d_copy_gbps(dev)
    if d2d:
        rf["d2d_copy"] = {"gbps": d2d, "what": "torch copy of 2 GiB device to device on this box, read + written bytes / time (best of 5)"}
    if pmc:
        rf["pmc_gbps"] = pmc / (k2_avg_ms * 1e-3) / 1e9
        rf["frac_pmc"] = rf["pmc_gbps"] / HBM_PEAK_GBS
        rf["live_over_pmc"] = gathered / pmc
    # Narrow rows (the dominant kernel runs 4 or 8 lanes per row: rows of up to 128 bytes, one request each): the kernel's own count is
    # what it ASKED for — L2 serves a fifth of those requests (the hashes re-read per (k-mer, block), rows that share a 128-byte line),
    # so requested bytes / time is not a fraction of the HBM peak.  `frac` for these workloads is the fabric-side figure of the committed
    # FETCH_SIZE pass of the same command (x 1024 x 1: 64-byte requests are tallied at their size) over the kernel time of THIS run; the
    # live count stays beside it as `requested_over_peak`.  For 16-B-per-lane streaming forms the two agree to 0.2 %.
    strides_local = [bi_["stride"] for bi_ in (db.block_info(b_) for b_ in range(int(info.n_blocks))) if bi_["local"]]
    narrow_rows = bool(strides_local) and 0 < max(strides_local) <= 128
    rf["requested_over_peak"] = rf["frac"]
    rf["frac_basis"] = "requested bytes (kernel's own count) = fabric bytes to 0.2 % for 16-B-per-lane streaming rows"
    if narrow_rows:
        if pmc:
            rf["traffic_requested"] = gathered
            rf["traffic"] = pmc
            rf["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE pass of the same command (" + str(pmc_src) + "): narrow-row form, the live count is `traffic_requested`"
            rf["achieved"] = rf["pmc_gbps"]
            rf["frac"] = rf["frac_pmc"]
            rf["traffic_over_algorithmic"] = pmc / alg_bytes
            rf["frac_basis"] = "fabric bytes (FETCH_SIZE) / kernel time of this run / HBM peak; narrow rows: requested bytes include what L2 served"
        else:
            rf["frac_basis"] = "REQUESTED bytes (no FETCH_SIZE pass committed for this command): an upper bound of the fabric-side fraction on narrow rows"

    if extras:
        # ---- the kernel alone and the data-independent variant: no host half; then sector pruning switched off (every row byte
        #      of every k-mer is fetched whatever the index holds: the pruning gain depends on the data, this number does not)
        def kernel_only(nsteps, env=None):
            old = {}
            for k_, v_ in (env or {}).items():
                old[k_] = os.environ.get(k_)
                os.environ[k_] = v_
            try:
                tms = []
                barrier()
                t1 = time.perf_counter()
                run_steps(0, nsteps, with_host=False, times=tms)
                barrier()
                dt = time.perf_counter() - t1
                ms = [t_[1] for t_ in tms]
            finally:
                for k_, v_ in old.items():
                    if v_ is None:
                        os.environ.pop(k_, None)
                    else:
                        os.environ[k_] = v_
            return dt / nsteps, float(np.mean(ms))
        per_step, _ = kernel_only(max(2, min(steps, 8)))
        out["device_only"] = {"value": B / per_step, "unit": unit, "ms_per_step": per_step * 1e3,
                              "note": "reads in HBM -> raw (read, column, count) hit tuples in host memory, no host half (round 1's `value`)"}
        local_strides = [bi["stride"] for bi in (db.block_info(b_) for b_ in range(int(info.n_blocks))) if bi["local"]]
        stride0 = max(local_strides) if local_strides else 0
        if 0 < stride0 <= 64:
            # narrow rows (one 64-byte request per (k-mer, block)): the bound is the rate at which the L2->fabric path serves
            # requests that miss L2, not bytes — 56e9/s whatever their size up to 128 B (profiles/r02_ubench_cache.txt)
            req = rf["row_bytes_per_launch"] / stride0 + rf["hash_bytes_per_launch"] / 128.0  # one request per row; hashes arrive in 128-byte lines
            out["roofline"]["requests"] = {"what": "L2->fabric requests (one per narrow row + one per 128-byte line of hashes)",
                                           "per_launch": req, "per_s": req / (k2_avg_ms * 1e-3),
                                           "ubench_per_s": "55-60e9 for 64-B gathers over >= 64 MiB, 158e9 L2-resident (profiles/r02_ubench_cache.txt): a "
                                                           "measured rate of another kernel, not a ceiling"}
        # sector pruning switched off: every row byte of every k-mer is fetched whatever the index holds (traffic = algorithmic
        # bytes + row padding), the data-independent figure of the same kernel
        _, k2_np = kernel_only(max(2, min(steps, 4)), {"KMCPG_PRUNE": "0"})
        g_np = float(np.mean([r_ + h_ for r_, h_ in measure_gathered({"KMCPG_PRUNE": "0"})]))
        rf["prune_off"] = {"kernel_ms": k2_np, "traffic": g_np, "achieved": g_np / (k2_np * 1e-3) / 1e9, "frac": g_np / (k2_np * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_gbps": alg_bytes / (k2_np * 1e-3) / 1e9, "algorithmic_over_peak": alg_bytes / (k2_np * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic_over_algorithmic": g_np / alg_bytes}

    # ---- the drop-in boundary with host buffers (PCIe-inclusive; reported beside `value`, never as `value`): batches through
    #      kmcpg_submit / kmcpg_wait — reads in host memory in; H2D, K1, K2, D2H of the hits, float64 thresholds, FPR, sort;
    #      finalized matches in host memory out — three batches in flight, and one batch alone through kmcpg_search_batch
    if rank == 0 and world == 1 and extras:
        hb = [(b_.reads.cpu().numpy(), b_.offs.cpu().numpy().astype(np.uint64)) for b_ in batches]
        import threading
        NB = min(max(8, 2 * steps), 16)
        HT = 2  # host threads, each keeping two batches in flight (the C++ CLI runs two searcher threads the same way)

        submit_one = [lambda i: db.submit(*hb[i % len(hb)], params=params)]

        def pump(t_, nb):
            tk = []
            for i in range(t_, nb, HT):
                if len(tk) == 2:
                    db.wait(tk.pop(0), count_only=True)
                tk.append(submit_one[0](i))
            while tk:
                db.wait(tk.pop(0), count_only=True)

        def pumped(nb):
            th = [threading.Thread(target=pump, args=(t_, nb)) for t_ in range(HT)]
            t1 = time.perf_counter()
            [x.start() for x in th]
            [x.join() for x in th]
            return (time.perf_counter() - t1) / nb

        # one batch alone through kmcpg_search_batch (the C call, result freed, nothing copied to numpy): warm-up call first (it sizes
        # the lanes' staging and hit buffers), then the best of two
        db.search_packed_count(hb[0][0], hb[0][1], params=params)
        single, n_matches = 1e30, 0
        for _ in range(2):
            t1 = time.perf_counter()
            n_matches = db.search_packed_count(hb[0][0], hb[0][1], params=params)
            single = min(single, time.perf_counter() - t1)
        # batches through kmcpg_submit / kmcpg_wait, four in flight: an untimed round first (every lane meets a whole batch once: a lane
        # that last held a quarter-batch piece of the call above would otherwise grow its pinned buffers inside the timed region)
        pumped(2 * HT)
        dt = pumped(NB)
        out["value_host_to_host"] = B / dt
        out["host_boundary"] = {"value": B / dt, "unit": unit, "ms_per_batch": dt * 1e3, "batches": NB, "host_threads": HT, "in_flight": 2 * HT,
                                "single_batch_ms": single * 1e3, "single_batch_reads_per_s": B / single, "matches": n_matches,
                                "note": "kmcpg_submit/kmcpg_wait: host buffers in, finalized matches out (staging copy, PCIe both ways and "
                                        "the host half included); single_batch = one kmcpg_search_batch call on its own (the library sends a large "
                                        "batch through its lanes as up to 4 pieces, so upload / kernels / copy / expansion overlap inside the call)"}
        # long queries (genomes, HiFi reads): the same batches through kmcpg_submit_packed — 2-bit codes + exception runs, packed once up front
        # as a reader that packs while it parses hands them over (kmcp-search -g does): no text is read inside the timed region
        if bases_per_launch / B >= 1000:
            hbp, pins = [], []
            for r_, o_ in hb:
                pin = lib.PinnedBytes((len(r_) + 3) // 4 + 8)  # kmcpg_host_alloc: the reader packs straight into page-locked memory
                pins.append(pin)
                pin.a[:] = 0
                codes_, exc_, _tot = lib.pack2([r_], codes=pin.a)
                hbp.append((codes_, o_, exc_))
            submit_one[0] = lambda i: db.submit_packed(*hbp[i % len(hbp)], params=params)
            pumped(2 * HT)
            dtp = pumped(NB)
            out["value_host_to_host_packed"] = B / dtp
            out["host_boundary"]["packed"] = {"value": B / dtp, "unit": unit, "ms_per_batch": dtp * 1e3,
                                              "note": "kmcpg_submit_packed / kmcpg_wait: codes (a quarter of the text, in memory from kmcpg_host_alloc: uploaded "
                                                      "from where the reader packed them, no staging copy) + exception runs in, finalized matches out"}
            # ... and codes in ordinary memory (copied to the lane's pinned staging inside the call)
            hbq = [(np.array(c_[:]), o_, e_) for c_, o_, e_ in hbp]
            submit_one[0] = lambda i: db.submit_packed(*hbq[i % len(hbq)], params=params)
            pumped(2 * HT)
            dtq = pumped(NB)
            out["host_boundary"]["packed_staged"] = {"value": B / dtq, "unit": unit, "ms_per_batch": dtq * 1e3}
            del hbp, hbq
            for pin in pins:
                pin.close()
        del hb
    # ---- CPU oracle on a bounded sample.  N = 1: the cpu_baseline leg (timed, ALL blocks copied back from HBM when host memory
    #      allows) + parity of the GPU hits on that sample.  N > 1: the same parity check on rank 0 over rows fetched from every
    #      rank (each owner reads its blocks back, rank 0 receives them), untimed: the merged hit list of a multi-GPU run is
    #      checked against the CPU restatement in the very run that produced it.
    want_oracle = cpu_baseline and (world == 1 or os.environ.get("KMCP_BENCH_PARITY_N", "1") != "0")
    if want_oracle:
        import shutil
        index_bytes = int(info.matrix_bytes)
        S = wl["n_blocks"]
        if rank == 0:
            avail = host_memory_available()
            if not index_bytes * 1.15 + (8 << 30) < avail:
                S = max(1, min(wl["n_blocks"], int((avail - (8 << 30)) * 0.8 / (index_bytes / wl["n_blocks"]))))
        if coll:
            t_s = torch.tensor([S], dtype=torch.int64, device="cpu" if ctx.same_gpu else dev)
            dist.broadcast(t_s, src=0)
            S = int(t_s.item())
        t1 = time.perf_counter()
        blocks = []
        owners = None
        if coll:
            loc = torch.tensor([1 if db.block_info(b)["local"] else 0 for b in range(S)], dtype=torch.int64, device="cpu" if ctx.same_gpu else dev)
            allloc = [torch.zeros_like(loc) for _ in range(world)]
            dist.all_gather(allloc, loc)
            owners = [next(r_ for r_ in range(world) if int(allloc[r_][b].item())) for b in range(S)]
        for b in range(S):
            bi = db.block_info(b)
            own = owners[b] if owners else 0
            rows = None
            if rank == own or rank == 0:
                rows = np.empty((bi["num_sigs"], bi["row_bytes"]), dtype=np.uint8)
            if rank == own:
                chunk = 262144
                for r0 in range(0, bi["num_sigs"], chunk):
                    db.read_row_range(b, r0, rows[r0:r0 + chunk])
            if own != 0:  # block b travels owner -> rank 0 (RCCL send/recv of device tensors; host tensors over gloo)
                if rank == own:
                    t_ = torch.from_numpy(rows)
                    dist.send(t_ if ctx.same_gpu else t_.to(dev), dst=0)
                elif rank == 0:
                    t_ = torch.empty(rows.shape, dtype=torch.uint8, device="cpu" if ctx.same_gpu else dev)
                    dist.recv(t_, src=own)
                    rows[...] = t_.cpu().numpy()
                if rank != 0:
                    rows = None
            if rank == 0:
                blocks.append((bi["num_sigs"], bi["n_cols"], bi["col_base"], rows))
        copy_s = time.perf_counter() - t1
    if want_oracle and rank == 0:
        from oracle import oracle as O
        odb = O.OracleDB.from_memory(O.sketch_cfg(k=wl["k"], scale=max(1, wl.get("scale", 1)), minimizer_w=wl.get("minimizer_w", 0), syncmer_s=wl.get("syncmer_s", 0)),
                                     wl["num_hashes"], wl["fpr"], blocks, wl["kmers_per_col"])
        threads = effective_cpus()
        reads_h = batches[last].reads.cpu().numpy()
        offs_h = batches[last].offs.cpu().numpy().astype(np.uint64)
        oparams = O.default_params(min_qcov=params.min_qcov)

        def timed(refshape, target_s, fixed=0):
            R = fixed or cpu_sample_reads or wl.get("cpu_sample_start", 256)
            while True:  # grow the sample until it is several seconds of CPU work
                t2 = time.perf_counter()
                oqk, ohits = odb.search_batch(reads_h[:int(offs_h[R])], offs_h[:R + 1], oparams, threads=threads, refshape=refshape)
                tcpu = time.perf_counter() - t2
                if fixed or cpu_sample_reads or tcpu >= target_s or R >= B:
                    return R, tcpu, oqk, ohits
                R = min(B, int(R * max(2.0, 1.5 * target_s / max(tcpu, 1e-3))))

        # N > 1: a fixed small sample (a check, not a baseline); N = 1: sized to `cpu_target_s` seconds of CPU work
        R, tcpu, oqk, ohits = timed(False, cpu_target_s, fixed=0 if world == 1 else min(B, 8 * wl.get("cpu_sample_start", 256)))
        # same-run parity on the sample: GPU hits of these reads restricted to the sampled blocks == oracle hits
        hi_col = blocks[-1][2] + blocks[-1][1]
        g = hh[(hh[:, 0] < R) & (hh[:, 1] < hi_col)]
        g = g[np.lexsort((g[:, 1], g[:, 0]))]
        if os.environ.get("KMCP_BENCH_FAULT") == "parity" and len(g):  # test hook (tests/test_gpu_multirank.py): what a broken exchange would look like
            g = g[:-1]
        parity = bool(np.array_equal(g, ohits.astype(np.int64))) and bool(np.array_equal(oqk[:R], qk[:R]))
        frac_blocks = S / wl["n_blocks"]
        sample_txt = (f"{R} queries ({int(offs_h[R])} bases) x {S} of {wl['n_blocks']} blocks ({sum(b_[3].nbytes for b_ in blocks)/1e9:.1f} GB of index rows "
                      f"copied back from HBM in {copy_s:.1f} s) in {tcpu:.2f} s on {threads} threads: oracle ko_search_batch (OpenMP over queries, LUT vertical counters)"
                      + ("" if S == wl["n_blocks"] else f"; host memory holds only {S} blocks: value scaled by {frac_blocks:.4f}"))
        if world == 1:
            out["cpu_baseline"] = {
                "value": R / tcpu * frac_blocks, "unit": unit, "cores": threads, "kind": "port", "sample": sample_txt,
                "sample_short": f"{R} queries x {S}/{wl['n_blocks']} blocks, {tcpu:.1f} s on {threads} threads (oracle port, OpenMP)",
                "parity_on_sample": parity, "sample_hits": int(len(ohits)),
                "reference_binary": {"kmcp": shutil.which("kmcp"), "go": shutil.which("go"),
                                     "note": "BASELINE.md 3.1 probe: the Go reference is timed instead when a kmcp binary is on PATH (none in this image)"},
            }
            if wl["num_hashes"] == 1:  # the reference-shaped leg restates the single-hash worker (util-db-search.go:6811-6972)
                R2, tcpu2, oqk2, ohits2 = timed(True, cpu_target_s)
                same = bool(np.array_equal(ohits2, ohits[ohits[:, 0] < R2])) if R2 <= R else bool(np.array_equal(ohits2[ohits2[:, 0] < R], ohits))
                out["cpu_baseline"]["reference_shaped"] = {
                    "value": R2 / tcpu2 * frac_blocks, "unit": unit, "cores": threads,
                    "sample": f"{R2} queries x {S} blocks in {tcpu2:.2f} s: one worker per block, 64 buffered rows, byte transposition + "
                              "Count8 per column byte (util-db-search.go:6811-6972, :213-219), AVX2 movemask Count8",
                    "same_hits_as_port": same}
        else:
            out["parity_at_n"] = {"parity_on_sample": parity, "sample": sample_txt, "sample_hits": int(len(ohits)),
                                  "note": "rank 0 ran the CPU oracle on rows fetched from every rank's shard and compared the merged multi-GPU hit list of the sample with it"}
        odb.close()
        del blocks
        if not parity:
            # The line must still come out: a first run on hardware this code has never seen (N > 1) has to show WHAT broke, not a
            # traceback in one rank's stderr and seven ranks waiting at a barrier.  main() prints the line and exits nonzero.
            dg, do = {tuple(x) for x in g.tolist()}, {tuple(x) for x in ohits.astype(np.int64).tolist()}
            out["parity_failure"] = {"gpu_only": len(dg - do), "oracle_only": len(do - dg), "qkmers_differ": int(np.count_nonzero(oqk[:R] != qk[:R])),
                                     "first_gpu_only": sorted(dg - do)[:4], "first_oracle_only": sorted(do - dg)[:4]}
            print(f"bench.py: PARITY FAILURE on the sample: {out['parity_failure']}", file=sys.stderr)
    if coll:
        dist.barrier()

    # ---- (last: this leg plants whole genomes into the index, which the oracle comparison above must not see)
    # ---- the reference's OTHER published case on this index (`kmcp search -g -t 0.5`, one 4.6-5.6-Mbp genome, ALL of its k-mers,
    #      against the unchunked GTDB index: 12.7-13.7 s hot on 40 threads, benchmarks/searching/README.md:139-163): one query of ~5 M k-mers at a
    #      time through kmcpg_search_batch — host text in, finalized matches out; sort + unique on the device, the chunked COBS kernel
    if wl.get("genome_query") and rank == 0 and world == 1 and extras:
        gq = wl["genome_query"]
        gg = torch.Generator(device=dev)
        gg.manual_seed(4242)
        acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        glens = [int(x) for x in np.linspace(gq["min_len"], gq["max_len"], gq["n"])]
        goffs = torch.zeros(len(glens) + 1, dtype=torch.int64, device=dev)
        goffs[1:] = torch.cumsum(torch.tensor(glens, dtype=torch.int64, device=dev), 0)
        gseq = acgt[torch.randint(0, 4, (int(goffs[-1].item()),), generator=gg, device=dev)].contiguous()
        gcols = torch.randint(0, n_cols, (len(glens),), generator=gg, device=dev).to(torch.int32).contiguous()
        db.plant_reads_device(gseq.data_ptr(), goffs.contiguous().data_ptr(), len(glens), int(goffs[-1].item()), max(glens), gcols.data_ptr())
        torch.cuda.synchronize()
        gp = default_params()
        gp.min_qcov = gq.get("min_qcov", 0.5)
        gp.sort_by = wl.get("sort_by", 0)
        gh, go = gseq.cpu().numpy(), goffs.cpu().numpy().astype(np.uint64)
        per, found = [], 0
        for i_ in range(len(glens)):
            s_i = np.ascontiguousarray(gh[int(go[i_]):int(go[i_ + 1])])
            o_i = np.array([0, len(s_i)], dtype=np.uint64)
            best = 1e30
            for _ in range(3):
                t1 = time.perf_counter()
                db.search_packed_count(s_i, o_i, params=gp)
                best = min(best, time.perf_counter() - t1)
            per.append(best)
            res = db.search_packed(s_i, o_i, params=gp)
            mt = res.matches[int(res.offs[0]):int(res.offs[1])]
            found += int(any(int(m_["col"]) == int(gcols[i_].item()) and int(m_["mkmers"]) == int(res.qkmers[0]) for m_ in mt))
        db.search_packed_count(gh, go, params=gp)  # (untimed: the lanes' buffers grow to the batch)
        tb = 1e30
        for _ in range(2):
            t1 = time.perf_counter()
            db.search_packed_count(gh, go, params=gp)
            tb = min(tb, time.perf_counter() - t1)
        out["whole_genome_query"] = {"genomes": len(glens), "min_len": min(glens), "max_len": max(glens), "min_qcov": gp.min_qcov,
                                     "ms_per_genome": float(np.mean(per)) * 1e3, "ms_per_genome_min_max": [min(per) * 1e3, max(per) * 1e3],
                                     "ms_per_genome_in_one_batch": tb / len(glens) * 1e3, "planted_found_with_every_kmer": found,
                                     "reference_published_s": gq.get("reference_s", [12.7, 13.7]),
                                     "note": "one kmcpg_search_batch call per genome (best of 3): host text in, finalized matches out; the reference: "
                                             "`kmcp search -g` of one genome against this index layout, hot (whole genomes, all k-mers, -t 0.5, 40 threads: "
                                             "benchmarks/searching/README.md:139-163; FracMinHash sketches, -t 0.4, 8 threads: :382-432)"}
        del gseq, gh

    db.close()
    del bufs, batches
    torch.cuda.empty_cache()
    return out



# ---------------------------------------------------------------------------------------------------------------------------------
# "report also end-to-end incl. FASTQ parse + TSV write" (reference: search.go:793-1000 reader, :448-588 writer).
# BASELINE configs[1] as a database ON DISK and 10 M reads as a FASTQ FILE, searched by the kmcp-search binary (C++ host above the
# C ABI: parallel FASTQ reader, kmcpg_search_batch_pairs, parallel row formatter) into a TSV file and into /dev/null; wall clock of
# the whole process (exec to exit: HIP start-up and the upload of the index included).  A prefix of the TSV is byte-compared with the
# rows the CPU oracle prints for the same reads.
# ---------------------------------------------------------------------------------------------------------------------------------
def _pick_workdir(need_bytes):
    import shutil
    import tempfile
    for d in ("/dev/shm", tempfile.gettempdir(), ROOT):
        try:
            if shutil.disk_usage(d).free > 1.2 * need_bytes:
                return tempfile.mkdtemp(prefix="kmcp_cli_e2e_", dir=d)
        except OSError:
            continue
    return None


def _write_fastq(path, reads, first_id):
    """reads: uint8 [n, L] -> four-line FASTQ records `@r<9 digits>` / bases / `+` / quality 'I' x L, appended to `path`"""
    n, L = reads.shape
    rec = np.empty((n, 12 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    ids = np.arange(first_id, first_id + n, dtype=np.int64)
    for d in range(9):
        rec[:, 2 + d] = (ids // 10 ** (8 - d)) % 10 + ord("0")
    rec[:, 11] = ord("\n")
    rec[:, 12:12 + L] = reads
    rec[:, 12 + L:15 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 15 + L:15 + 2 * L] = ord("I")
    rec[:, 15 + 2 * L] = ord("\n")
    with open(path, "ab") as fh:
        fh.write(rec.tobytes())


def run_cli_end_to_end(ctx, n_reads=10_000_000, check_reads=100_000, check_budget_s=25.0):
    import re
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from kmcp_amd import Database, lib
    cli = os.path.join(ROOT, "kmcp_amd", "kmcp-search")
    if not os.path.exists(cli):
        return {"skipped": "kmcp_amd/kmcp-search is not built"}
    wl = dict(WORKLOADS["config1"])
    L = READ_LEN
    need = n_reads * (16 + 2 * L + 140) + 2.0e9
    work = _pick_workdir(need)
    if work is None:
        return {"skipped": f"no directory with {need/1e9:.1f} GB free"}
    out = {"workdir": os.path.dirname(work), "reads": n_reads}
    try:
        t0 = time.time()
        spec = lib.SynthSpec(k=wl["k"], num_hashes=wl["num_hashes"], fpr=wl["fpr"], n_blocks=wl["n_blocks"], cols_per_block=wl["cols_per_block"],
                             num_sigs=wl["num_sigs"], kmers_per_col=wl["kmers_per_col"], seed=42, sigs_step=wl.get("sigs_step", 0))
        db = Database.open_synthetic(spec, device=ctx.dev_index)
        n_cols = int(db.info.n_cols)

        def plant(frag, offs, n, total, maxlen, cols):
            db.plant_reads_device(frag.data_ptr(), offs.data_ptr(), n, total, maxlen, cols.data_ptr())

        fq = os.path.join(work, "reads.fq")
        first_reads = None
        done = 0
        while done < n_reads:  # 90 % of the reads are mutated, randomly reverse-complemented fragments planted into a random column
            nb = min(1 << 20, n_reads - done)
            bt = make_batch(ctx.dev, wl, nb, n_cols, 7000 + done // (1 << 20), plant)
            h = bt.reads.cpu().numpy().reshape(nb, L)
            if first_reads is None:
                first_reads = h[:check_reads].copy()
            _write_fastq(fq, h, done)
            done += nb
            del bt, h
        torch.cuda.synchronize()
        db_root = os.path.join(work, "db")
        db_dir = db.save(db_root)
        db.close()
        torch.cuda.empty_cache()
        out["setup_s"] = time.time() - t0
        out["fastq_bytes"] = os.path.getsize(fq)
        out["db_bytes"] = sum(os.path.getsize(os.path.join(db_dir, f)) for f in os.listdir(db_dir))

        tsv = os.path.join(work, "out.tsv")
        env = dict(os.environ)
        for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k_, None)

        def one(dest):
            if dest != "/dev/null" and os.path.exists(dest):
                os.unlink(dest)  # (truncating last run's 0.9 GB of tmpfs pages inside the timed process cost 0.1-0.3 s: not part of a search)
            t1 = time.perf_counter()
            r = subprocess.run([cli, "-d", db_root, fq, "-o", dest], capture_output=True, text=True, env=env, timeout=900)
            wall = time.perf_counter() - t1
            if r.returncode != 0:
                raise RuntimeError("kmcp-search failed: " + r.stderr[-2000:])
            m = re.search(r"pipeline: ([\d.]+) s in the GPU library, ([\d.]+) s formatting/writing, ([\d.]+) s waiting for the reader; reader: ([\d.]+) s parsing, "
                          r"([\d.]+) s\s+blocked; ([\d.]+) s before the search started", r.stderr)
            split = dict(zip(("gpu_library_s", "formatting_s", "waiting_for_reader_s", "reader_parsing_s", "reader_blocked_s", "before_search_s"),
                             (float(x) for x in m.groups()))) if m else {}
            me = re.search(r"elapsed time: ([\d.]+)s", r.stderr)
            if me and split:
                split["elapsed_in_main_s"] = float(me.group(1))
                split["search_phase_s"] = float(me.group(1)) - split["before_search_s"]
            mm = re.search(r"matches: (\d+), checksum ([0-9a-f]{16})", r.stderr)
            return wall, split, (int(mm.group(1)), mm.group(2)) if mm else None

        one(tsv)  # untimed: first touch of the binary, the database files and the driver
        runs_file = [one(tsv) for _ in range(3)]
        runs_null = [one("/dev/null") for _ in range(3)]
        best_f = min(runs_file, key=lambda x: x[0])
        best_n = min(runs_null, key=lambda x: x[0])
        rows = best_f[2][0] if best_f[2] else None
        out.update({
            "value": n_reads / best_f[0], "unit": "reads/s", "wall_s": best_f[0], "wall_s_all": [r_[0] for r_ in runs_file],
            "value_dev_null": n_reads / best_n[0], "wall_s_dev_null": best_n[0], "wall_s_dev_null_all": [r_[0] for r_ in runs_null],
            "rows": rows, "rows_per_s": (rows / best_f[0]) if rows else None, "tsv_bytes": os.path.getsize(tsv),
            # the same run without what a process pays once whatever it searches: exec + dynamic loading + exit (wall - elapsed_in_main), HIP
            # runtime start-up (0.2 s, tools/ubench_init.cpp) and the upload of the index (before_search): first batch submitted -> last row written
            "value_search_phase": (n_reads / best_f[1]["search_phase_s"]) if best_f[1].get("search_phase_s") else None,
            "split": best_f[1], "split_dev_null": best_n[1],
            "definition": "wall clock of the kmcp-search process (exec to exit), best of 3: FASTQ file -> parallel parse -> kmcpg_search_batch_pairs -> "
                          "parallel row formatting -> TSV file (value) or /dev/null (value_dev_null); HIP start-up and the index upload included",
            "checksums_agree": len({r_[2] for r_ in runs_file + runs_null}) == 1,
        })
        # ---- the TSV's prefix against the oracle's rows (test infrastructure as the checker: ko_search + ko_format_match per read)
        from oracle import oracle as O
        import ctypes as C
        odb = O.OracleDB(db_dir)
        OL = O.lib()
        p = O.default_params()
        threads = effective_cpus()
        t2 = time.perf_counter()

        def rows_of(lo, hi):
            buf = C.create_string_buffer(4096)
            lines = []
            for i in range(lo, hi):
                if time.perf_counter() - t2 > check_budget_s:
                    return lo, i, lines
                res = O.Result()
                r = first_reads[i].tobytes()
                OL.ko_search(odb.h, r, len(r), None, 0, C.byref(p), C.byref(res))
                for j in range(max(0, res.nmatches)):
                    OL.ko_format_match(buf, 4096, b"r%09d" % i, C.byref(res), C.byref(res.matches[j]), i)
                    lines.append(buf.value)
                OL.ko_result_free(C.byref(res))
            return lo, hi, lines

        step = 512
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(lambda lo: rows_of(lo, min(lo + step, len(first_reads))), range(0, len(first_reads), step)))
        odb.close()
        covered = 0
        want = []
        for lo, hi, lines in parts:  # the longest prefix every chunk of which was finished inside the budget
            if hi < min(lo + step, len(first_reads)):
                break
            covered = hi
            want.extend(lines)
        got = []
        with open(tsv, "rb") as fh:
            for ln in fh:
                if ln.startswith(b"#"):
                    continue
                if int(ln.rsplit(b"\t", 1)[1]) >= covered:
                    break
                got.append(ln)
        out["parity_on_sample"] = bool(covered > 0 and got == want)
        out["sample"] = f"first {covered} reads: {len(want)} TSV rows byte-compared with the oracle's (ko_search + ko_format_match, {threads} threads, {time.perf_counter()-t2:.1f} s)"
        out["sample_reads"] = covered
        out["sample_rows"] = len(want)
        if not out["parity_on_sample"]:
            out["parity_failure"] = {"gpu_only": len(set(got) - set(want)), "oracle_only": len(set(want) - set(got)), "qkmers_differ": 0}
    finally:
        keep = os.environ.get("KMCP_BENCH_KEEP")  # experiments (tools/ab/r06_cli_probe.sh): leave the database and the reads behind
        if keep:
            shutil.rmtree(keep, ignore_errors=True)
            shutil.move(work, keep)
        else:
            shutil.rmtree(work, ignore_errors=True)
    return out


def run_cli_genome_search(ctx, n_genomes=256, check_genomes=3):
    """BASELINE configs[2] through the kmcp-search binary: the genome-search index on disk (kmcpg_save_db), `n_genomes` assemblies of 4 Mbp as
    FASTA FILES (80 bases per line), `kmcp-search -g -t 0.4 -s jacc` over all of them (reference: search.go:885-915, benchmarks/searching/
    README.md:382-432), wall clock exec to exit; the rows of the first `check_genomes` files are compared with the oracle's."""
    import ctypes as C
    import re
    import shutil
    import subprocess
    from kmcp_amd import Database, lib
    cli = os.path.join(ROOT, "kmcp_amd", "kmcp-search")
    if not os.path.exists(cli):
        return {"skipped": "kmcp_amd/kmcp-search is not built"}
    wl = dict(WORKLOADS["config2_genome_search"])
    L = int(wl["read_len"])
    work = _pick_workdir(n_genomes * (L + L // 80 + 64) + 4.0e9)
    if work is None:
        return {"skipped": "no directory with enough free space"}
    out = {"workdir": os.path.dirname(work), "genomes": n_genomes}
    try:
        t0 = time.time()
        spec = lib.SynthSpec(k=wl["k"], num_hashes=wl["num_hashes"], fpr=wl["fpr"], n_blocks=wl["n_blocks"], cols_per_block=wl["cols_per_block"],
                             num_sigs=wl["num_sigs"], kmers_per_col=wl["kmers_per_col"], seed=42, sigs_step=wl.get("sigs_step", 0), scale=wl.get("scale", 0))
        db = Database.open_synthetic(spec, device=ctx.dev_index)
        n_cols = int(db.info.n_cols)

        def plant(frag, offs, n, total, maxlen, cols):
            db.plant_reads_device(frag.data_ptr(), offs.data_ptr(), n, total, maxlen, cols.data_ptr())

        bt = make_batch(ctx.dev, wl, n_genomes, n_cols, 9000, plant)
        g = bt.reads.cpu().numpy().reshape(n_genomes, L)
        del bt
        files = []
        nl = np.full((L // 80, 1), ord("\n"), dtype=np.uint8)
        for i in range(n_genomes):
            fn = os.path.join(work, f"asm{i:05d}.fasta")
            with open(fn, "wb") as fh:
                fh.write(b">asm%05d synthetic assembly\n" % i)
                fh.write(np.concatenate([g[i].reshape(L // 80, 80), nl], axis=1).tobytes())
            files.append(fn)
        torch.cuda.synchronize()
        db_root = os.path.join(work, "db")
        db_dir = db.save(db_root)
        db.close()
        torch.cuda.empty_cache()
        out["setup_s"] = time.time() - t0
        tsv = os.path.join(work, "out.tsv")
        env = dict(os.environ)
        for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k_, None)
        lst = os.path.join(work, "files.txt")
        with open(lst, "w") as fh:
            fh.write("\n".join(files) + "\n")

        def one():
            if os.path.exists(tsv):
                os.unlink(tsv)
            t1 = time.perf_counter()
            r = subprocess.run([cli, "-d", db_root, "-g", "-t", str(wl["min_qcov"]), "-s", "jacc", "--infile-list", lst, "-o", tsv], capture_output=True, text=True,
                               env=env, timeout=900)
            wall = time.perf_counter() - t1
            if r.returncode != 0:
                raise RuntimeError("kmcp-search -g failed: " + r.stderr[-2000:])
            m = re.search(r"([\d.]+) s before the search started", r.stderr)
            me = re.search(r"elapsed time: ([\d.]+)s", r.stderr)
            return wall, (float(me.group(1)) - float(m.group(1))) if (m and me) else None

        one()
        runs = [one() for _ in range(3)]
        best = min(runs, key=lambda x: x[0])
        rows = sum(1 for ln in open(tsv, "rb") if not ln.startswith(b"#"))
        out.update({"value": n_genomes / best[0], "unit": "queries/s", "wall_s": best[0], "wall_s_all": [r_[0] for r_ in runs], "rows": rows,
                    "value_search_phase": (n_genomes / best[1]) if best[1] else None,
                    "definition": "wall clock of `kmcp-search -g` (exec to exit, best of 3) over FASTA files of 4 Mbp each in /dev/shm: parallel file "
                                  "readers pack every assembly to 2-bit codes as they join its lines, kmcpg_submit_packed, rows formatted from pairs"})
        from oracle import oracle as O
        odb = O.OracleDB(db_dir)
        OL = O.lib()
        p = O.default_params(min_qcov=wl["min_qcov"], sort_by=2)
        want = []
        buf = C.create_string_buffer(4096)
        t2 = time.perf_counter()
        for i in range(min(check_genomes, n_genomes)):
            res = O.Result()
            whole = g[i].tobytes()
            OL.ko_search(odb.h, whole, len(whole), None, 0, C.byref(p), C.byref(res))
            for j in range(max(0, res.nmatches)):
                OL.ko_format_match(buf, 4096, b"asm%05d" % i, C.byref(res), C.byref(res.matches[j]), i)
                want.append(buf.value)
            OL.ko_result_free(C.byref(res))
        odb.close()
        got = []
        with open(tsv, "rb") as fh:
            for ln in fh:
                if not ln.startswith(b"#") and int(ln.rsplit(b"\t", 1)[1]) < min(check_genomes, n_genomes):
                    got.append(ln)
        out["parity_on_sample"] = bool(want and got == want)
        out["sample"] = f"rows of the first {min(check_genomes, n_genomes)} assemblies ({len(want)}) byte-compared with the oracle's ({time.perf_counter()-t2:.1f} s)"
        out["sample_reads"] = min(check_genomes, n_genomes)
        if not out["parity_on_sample"]:
            out["parity_failure"] = {"gpu_only": len(set(got) - set(want)), "oracle_only": len(set(want) - set(got)), "qkmers_differ": 0}
    finally:
        keep = os.environ.get("KMCP_BENCH_KEEP")
        if keep:
            shutil.rmtree(keep, ignore_errors=True)
            shutil.move(work, keep)
        else:
            shutil.rmtree(work, ignore_errors=True)
    return out

# ---------------------------------------------------------------------------------------------------------------------------------
# The ONE JSON line.  Numbers only, < 6 KB (the driver keeps an 8 KB tail of stdout and parses the line out of it): every sentence
# (definitions, sources, sample descriptions) and every sub-measurement lives in the sidecar `bench_detail.json`.
# ---------------------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000
_HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_pmc", "algorithmic_bytes_per_launch", "algorithmic_over_peak",
              "traffic_over_algorithmic", "kernel_ms", "kmers_kernel_ms", "finalize_kernels_ms", "requested_over_peak")


def _num(v, digits=6):
    """floats to `digits` significant digits (the line is for parsers and tables, the sidecar keeps full precision)"""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    return v


def _roofline_numbers(rf):
    r = {k: _num(rf.get(k)) for k in _ROOF_KEYS if k in rf}
    if isinstance(r.get("kernel"), str):
        r["kernel"] = r["kernel"].split(" ")[0]  # the template instance; what runs beside it is in the sidecar
    if rf.get("prune_off"):
        r["prune_off_frac"] = _num(rf["prune_off"].get("frac"))
    return r


def _cpu_numbers(cb, short_sample=True):
    if not cb:
        return None
    c = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
         "parity_on_sample": cb.get("parity_on_sample")}
    if short_sample and cb.get("sample_short"):
        c["sample"] = cb["sample_short"]
    if cb.get("reference_shaped"):
        c["reference_shaped"] = {"value": _num(cb["reference_shaped"].get("value"))}
    return c


def _secondary_numbers(o):
    rf = o.get("roofline") or {}
    cb = o.get("cpu_baseline") or {}
    d = {"value": o.get("value"), "unit": o.get("unit"), "ms_per_step": o.get("ms_per_step"), "value_host_to_host": o.get("value_host_to_host"),
         "value_host_to_host_packed": o.get("value_host_to_host_packed"),
         "kernel_ms": rf.get("kernel_ms"), "frac": rf.get("frac"), "requested_over_peak": rf.get("requested_over_peak"),
         "algorithmic_over_peak": rf.get("algorithmic_over_peak"),
         "traffic_over_algorithmic": rf.get("traffic_over_algorithmic"), "cpu": cb.get("value"), "cpu_cores": cb.get("cores"),
         "cpu_reference_shaped": (cb.get("reference_shaped") or {}).get("value"), "parity_on_sample": cb.get("parity_on_sample"),
         "planted_recall": o.get("planted_recall"),
         # (the kmcp-search end-to-end leg)
         "value_dev_null": o.get("value_dev_null"), "value_search_phase": o.get("value_search_phase"), "wall_s": o.get("wall_s"), "rows_per_s": o.get("rows_per_s"), "reads": o.get("reads"), "genomes": o.get("genomes"),
         "sample_reads": o.get("sample_reads"),
         # (the published whole-genome query on this index: ms per 5-Mbp genome, all k-mers; reference 12 700-13 700 ms)
         "genome_query_ms": (o.get("whole_genome_query") or {}).get("ms_per_genome"),
         "genome_query_found": (o.get("whole_genome_query") or {}).get("planted_found_with_every_kmer")}
    if "parity_on_sample" in o:
        d["parity_on_sample"] = o["parity_on_sample"]
    for k, v in (o.get("split") or {}).items():
        d[k] = v
    if o.get("error") or o.get("skipped"):
        return {"error": str(o.get("error") or o.get("skipped"))[:120]}
    return {k: _num(v, 5) for k, v in d.items() if v is not None}


def compact_line(out, detail_path="bench_detail.json"):
    """The contract line from a full result dict: headline keys + config + roofline + cpu_baseline + numeric secondaries."""
    line = {k: _num(out.get(k)) for k in _HEAD_KEYS}
    cfg = out.get("config") or {}
    line["config"] = {k: _num(cfg[k]) for k in ("workload", "batch_reads", "read_len", "mean_kmers_per_query", "k", "num_hashes", "index_bytes", "index_bytes_this_rank",
                                                "blocks", "columns", "parallelism", "search_flags") if k in cfg}
    line["roofline"] = _roofline_numbers(out.get("roofline") or {})
    cb = _cpu_numbers(out.get("cpu_baseline"))
    if cb:
        line["cpu_baseline"] = cb
    for k in ("value_host_to_host", "value_host_to_host_packed", "planted_recall", "hits_per_step", "matches_per_step"):
        if out.get(k) is not None:
            line[k] = _num(out[k])
    sb = out.get("sanity_batch")
    if sb:
        line["sanity_batch"] = {"hits": sb.get("hits"), "hits_checksum": sb.get("hits_checksum"), "batch_seed": sb.get("batch_seed")}
    rk = out.get("ranks")
    if rk:
        line["ranks"] = {"world_size": rk.get("world_size"), "backend": rk.get("backend"), "ranks_reporting": sorted(p_["rank"] for p_ in rk.get("per_rank", [])),
                         "k2_ms_min_max": [_num(x) for x in rk.get("k2_ms_min_max", [])]}
    if out.get("parity_at_n"):
        line["parity_at_n"] = {"parity_on_sample": out["parity_at_n"].get("parity_on_sample"), "sample_hits": out["parity_at_n"].get("sample_hits")}
    if out.get("parity_failure"):  # (the process exits with status 3 after printing this line)
        line["parity_ok"] = False
        line["parity_failure"] = {k: out["parity_failure"].get(k) for k in ("gpu_only", "oracle_only", "qkmers_differ")}
    if out.get("secondary"):
        line["secondary"] = {nm: _secondary_numbers(o) for nm, o in out["secondary"].items()}
    line["detail"] = detail_path
    txt = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    if len(txt) > LINE_LIMIT:  # never let the line outgrow the driver again: shed the secondaries' minor keys, then the secondaries
        for nm in list(line.get("secondary", {})):
            line["secondary"][nm] = {k: v for k, v in line["secondary"][nm].items() if k in ("value", "unit", "ms_per_step", "frac", "kernel_ms", "cpu", "parity_on_sample")}
        txt = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    if len(txt) > LINE_LIMIT:
        line.pop("secondary", None)
        txt = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    assert len(txt) <= LINE_LIMIT, len(txt)
    return txt


def _finite(o):
    """NaN / inf -> None, recursively (strict JSON for the sidecar)"""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return _finite(float(o))
    return o


def write_detail(out):
    """The full record (every sub-measurement and every sentence) beside bench.py and, on a gpurun box, under gpurun_out/."""
    txt = json.dumps(_finite(out), indent=1)
    written = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if d != ROOT and not os.path.isdir(d):
                continue
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                f.write(txt + "\n")
            written.append(os.path.join(d, "bench_detail.json"))
        except OSError:
            pass
    return written


def main():
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner to stdout when a
    # communicator is created): for the duration of the run file descriptor 1 points at stderr, and only the JSON line goes
    # to the real stdout at the very end.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This process holds up to three torch streams of its own
    # beside the library handle's four (kernels x 2, upload, copy-back): with four queues the uploads of the host-to-host legs end up behind
    # kernels (genome search, packed entry: 43.6 k -> 48.1 k genomes/s with eight; `value` itself does not move: profiles/r06_cobs_overlap.txt)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="gtdb", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-reads", type=int, default=0)
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = size the CPU sample to several seconds of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] numbers that ride along at N=1")
    ap.add_argument("--cli-only", type=int, default=0, metavar="READS", help="run only the kmcp-search end-to-end leg on this many reads and print its record (negative: the -g leg on that many genomes)")
    ap.add_argument("--no-extras", action="store_true", help="timed steps only (profiling runs: no pruning-off / host-boundary launches in the trace)")
    args = ap.parse_args()

    ctx = Ctx()
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != ctx.world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # started the way `--gpus 1` is started (plain `python bench.py --gpus N`): become the launcher the driver would
            # have used - one rank per GPU under torch.distributed.run on a free local port; rank 0 prints the line
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.dup2(real_stdout, 1)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("OMP_NUM_THREADS", str(max(1, effective_cpus() // args.gpus)))
            os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                      "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
        args.gpus = ctx.world
    # KMCP_BENCH_SAME_GPU=1 (debugging on a 1-GPU box): every rank uses GPU 0 and the exchange runs over gloo, because
    # RCCL refuses two ranks on one device.  Never set by the driver; the measured path is nccl = RCCL over xGMI.
    ctx.same_gpu = os.environ.get("KMCP_BENCH_SAME_GPU") == "1"
    ctx.dev_index = 0 if ctx.same_gpu else local_rank
    torch.cuda.set_device(ctx.dev_index)
    ctx.dev = torch.device("cuda", ctx.dev_index)
    # KMCP_BENCH_FORCE_DIST=1 (tests on a 1-GPU box): a one-rank RCCL group, and the N > 1 code path of this script with it
    force = os.environ.get("KMCP_BENCH_FORCE_DIST") == "1" and ctx.world == 1
    ctx.collective = ctx.world > 1 or force
    if ctx.collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")  # the collectives of step i run beside the kernels of step i+1
        if force:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("MASTER_PORT", "29749"), world_size=1, rank=0, device_id=ctx.dev)
        elif ctx.same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=ctx.dev)

    if args.cli_only:
        rec = run_cli_genome_search(ctx, n_genomes=-args.cli_only) if args.cli_only < 0 else run_cli_end_to_end(ctx, n_reads=args.cli_only, check_reads=min(100_000, args.cli_only))
        os.dup2(real_stdout, 1)
        os.write(1, (json.dumps(_finite(rec)) + "\n").encode())
        sys.exit(3 if rec.get("parity_failure") or rec.get("error") else 0)
    out = run_workload(args.workload, ctx, args.steps, args.warmup, args.batch_reads, cpu_baseline=not args.no_cpu_baseline,
                       cpu_sample_reads=args.cpu_sample_reads, extras=not args.no_extras)
    if ctx.world == 1 and args.workload == "gtdb" and not args.no_secondary and not args.batch_reads:
        sec = run_workload("config1", ctx, min(max(args.steps, 5), 20), 2, cpu_baseline=not args.no_cpu_baseline, cpu_target_s=3.0)
        keys = ("value", "value_host_to_host", "value_host_to_host_packed", "unit", "ms_per_step", "config", "roofline", "planted_recall", "sanity_batch", "device_only", "host_boundary",
                "cpu_baseline", "hits_per_step", "matches_per_step", "setup_s", "parity_failure", "whole_genome_query")
        out["secondary"] = {"config1": {k: sec[k] for k in keys if k in sec}}
        # the same index with every block on its own (what a database with a different NumSigs per block gets): KMCPG_FUSE=0
        os.environ["KMCPG_FUSE"] = "0"
        try:
            unf = run_workload("config1", ctx, min(args.steps, 3), 1, cpu_baseline=False)
        finally:
            os.environ.pop("KMCPG_FUSE", None)
        out["secondary"]["config1_ungrouped"] = {k: unf[k] for k in keys if k in unf}
        # the configuration of the reference's own published short-read numbers (unchunked GTDB, k = 31, -b 1024, -t 0.8)
        pub = run_workload("gtdb_unchunked_k31", ctx, min(max(args.steps, 5), 20), 2, cpu_baseline=not args.no_cpu_baseline, cpu_target_s=3.0)
        out["secondary"]["gtdb_unchunked_k31"] = {k: pub[k] for k in keys + ("metric",) if k in pub}
        out["secondary"]["gtdb_unchunked_k31"]["published"] = {
            "value": [18.9e3, 21.3e3], "unit": "reads/s", "threads": 40,
            "source": "reference benchmarks/searching/README.md:186-229 (1.14-1.41 M reads in 53.4-72.8 s, kmcp v0.9.0, hot page cache)"}
        # BASELINE.json configs[2] and configs[4] with queries that MATCH (families of relatives / reads sampled from planted chunks)
        # (steps of 4-10 ms: 40 of them, so that the first step after an idle period — sometimes published late, profiles/r05_restart_stall.txt —
        # is 1/40th of the average and not 1/6th)
        for nm, st_ in (("config2_genome_search", 40), ("config4_hifi", 40), ("config4_hifi_uniform_sigs", 40)):
            r_ = run_workload(nm, ctx, max(args.steps, st_), 5, cpu_baseline=not args.no_cpu_baseline and nm != "config4_hifi_uniform_sigs",
                              cpu_target_s=3.0)
            out["secondary"][nm] = {k: r_[k] for k in keys + ("metric",) if k in r_}
        if os.environ.get("KMCP_BENCH_CLI", "1") != "0":
            try:
                out["secondary"]["cli_end_to_end"] = run_cli_end_to_end(ctx)
            except Exception as e:  # the leg must not take the headline line with it
                out["secondary"]["cli_end_to_end"] = {"error": repr(e)[:300]}
            try:
                out["secondary"]["cli_genome_search"] = run_cli_genome_search(ctx)
            except Exception as e:
                out["secondary"]["cli_genome_search"] = {"error": repr(e)[:300]}
        out["secondary"]["config2_genome_search"]["published"] = {
            "value": [1.6, 1.9], "unit": "queries/s", "threads": 8,
            "source": "reference benchmarks/searching/README.md:382-432 (genome search against GTDB, FracMinHash scale 1000: 0.53-0.62 s per query, 8 threads)"}
    if ctx.collective:
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)  # C stdio buffers of the libraries (they go where fd 1 points now: stderr)
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    if ctx.rank == 0:
        write_detail(out)
        os.write(1, (compact_line(_finite(out)) + "\n").encode())
        failed = bool(out.get("parity_failure")) or any(bool(o.get("parity_failure")) for o in (out.get("secondary") or {}).values())
        if failed:
            sys.exit(3)


if __name__ == "__main__":
    main()
