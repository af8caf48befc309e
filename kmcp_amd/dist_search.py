"""`kmcp search` over the GPUs of one node, one process per GPU (RCCL over xGMI):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m kmcp_amd.dist_search -d <db> reads.fq.gz -o out.tsv [-t 0.55 -c 10 -m 30 -f 0.01 -u 256 -s qcov -n 0 -K -H]

Every rank opens shard rank/N of the database (libkmcpgpu partitions the index blocks by bytes), reads the same input,
searches each batch against its blocks; the per-read hit lists are gathered on rank 0 (kmcp_amd.dist.gather_hits), finalized
there (kmcpg_finalize) and written as the reference's 15-column TSV with its trailer (kmcp/cmd/search.go:436-438, 517-575,
1022-1025).  Single-end input only here (ShardedSearcher.search itself takes pairs and --try-se); the single-process C++ CLI
`kmcp-search --gpus N` covers the rest of the flags.
"""
import argparse
import gzip
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from .dist import ShardedSearcher, hits_checksum
from .lib import default_params

HEADER = "#query\tqLen\tqKmers\tFPR\thits\ttarget\tchunkIdx\tchunks\ttLen\tkSize\tmKmers\tqCov\ttCov\tjacc\tqueryIdx\n"


def read_fastx(path):
    """(id, sequence) records of a FASTA/FASTQ file, gzip transparently; ID = header up to the first blank."""
    raw = sys.stdin.buffer if path == "-" else open(path, "rb")
    head = raw.peek(2)[:2] if hasattr(raw, "peek") else b""
    fh = gzip.open(raw) if head == b"\x1f\x8b" else raw
    name, seq, fastq, need = None, [], False, 0  # need: quality characters still to skip (FASTQ)
    for line in fh:
        line = line.rstrip(b"\r\n")
        if need > 0:
            need -= len(line)
            continue
        if not line:
            continue
        if name is None:  # expecting a header
            if line[:1] not in (b">", b"@"):
                raise SystemExit(f"{path}: not FASTA/FASTQ")
            fastq = line[:1] == b"@"
            name = (line[1:].split() or [b""])[0]
        elif fastq and line[:1] == b"+":
            s = b"".join(seq)
            yield name, s
            need, name, seq = len(s), None, []
        elif not fastq and line[:1] == b">":
            yield name, b"".join(seq)
            name, seq = (line[1:].split() or [b""])[0], []
        else:
            seq.append(line)
    if name is not None and not fastq:
        yield name, b"".join(seq)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="kmcp_amd.dist_search", add_help=True)
    ap.add_argument("-d", "--db-dir", required=True)
    ap.add_argument("-o", "--out-file", default="-")
    ap.add_argument("-t", "--min-query-cov", type=float, default=0.55)
    ap.add_argument("-T", "--min-target-cov", type=float, default=0.0)
    ap.add_argument("-c", "--min-kmers", type=int, default=10)
    ap.add_argument("-m", "--min-query-len", type=int, default=30)
    ap.add_argument("-f", "--max-fpr", type=float, default=0.01)
    ap.add_argument("-u", "--kmer-dedup-threshold", type=int, default=256)
    ap.add_argument("-s", "--sort-by", default="qcov", choices=["qcov", "tcov", "jacc"])
    ap.add_argument("-S", "--do-not-sort", action="store_true")
    ap.add_argument("-n", "--keep-top-scores", type=int, default=0)
    ap.add_argument("-K", "--keep-unmatched", action="store_true")
    ap.add_argument("-H", "--no-header-row", action="store_true")
    ap.add_argument("--gpu-batch", type=int, default=131072)
    ap.add_argument("files", nargs="*", default=["-"])
    a = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # KMCP_DIST_SAME_GPU=1 (tests on a 1-GPU box): every rank uses GPU 0 and the exchange runs over gloo, because RCCL refuses
    # two ranks on one device.  The production path is nccl = RCCL over xGMI, one GPU per rank.
    same_gpu = os.environ.get("KMCP_DIST_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    sub = [os.path.join(a.db_dir, d) for d in sorted(os.listdir(a.db_dir)) if os.path.exists(os.path.join(a.db_dir, d, "__db.yml"))]
    if not sub:
        raise SystemExit(f"invalid kmcp database: {a.db_dir}")
    srch = ShardedSearcher(sub[0], device=local_rank)
    db = srch.db
    if a.min_query_cov <= db.info.fpr:
        raise SystemExit(f"query coverage threshold ({a.min_query_cov:f}) should not be smaller than FPR of single bloom filter of index database "
                         f"({db.info.fpr:f})")
    p = default_params(min_qlen=a.min_query_len, min_matched=a.min_kmers, min_qcov=a.min_query_cov, min_tcov=a.min_target_cov, max_fpr=a.max_fpr,
                       dedup_threshold=a.kmer_dedup_threshold, sort_by={"qcov": 0, "tcov": 1, "jacc": 2}[a.sort_by], do_not_sort=int(a.do_not_sort),
                       top_n_scores=a.keep_top_scores)
    out = None
    if rank == 0:
        out = sys.stdout if a.out_file == "-" else (gzip.open(a.out_file, "wt") if a.out_file.endswith(".gz") else open(a.out_file, "w"))
        if not a.no_header_row:
            out.write(HEADER)
    total = matched = n_matches = 0
    check = 0  # order-independent checksum of the (queryIdx, column, mKmers) tuples: the same on any number of ranks

    def flush(ids, seqs):
        nonlocal total, matched, n_matches, check
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        buf = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if offs[-1] else np.zeros(1, dtype=np.uint8)
        res = srch.search(buf, offs, params=p)
        if rank != 0:
            total += len(ids)
            return
        if len(res.matches):
            owner = np.repeat(np.arange(len(ids), dtype=np.int64), np.diff(res.offs.astype(np.int64))) + total
            tup = np.stack([owner, res.matches["col"].astype(np.int64), res.matches["mkmers"].astype(np.int64)], axis=1)
            check = (check + int(hits_checksum(tup), 16)) & 0xFFFFFFFFFFFFFFFF
            n_matches += len(res.matches)
        for i, qid in enumerate(ids):
            qidx = total + i
            k = int(res.ksize[i])  # the k-mer size that answered (multi-k databases, search.go:530)
            ms = res.read(i)
            if len(ms) == 0:
                if a.keep_unmatched:
                    out.write(f"{qid.decode()}\t{res.qlen[i]}\t{res.qkmers[i]}\t0\t0\t\t-1\t0\t0\t{k}\t0\t0\t0\t0\t{qidx}\n")
                continue
            matched += 1
            for m in ms:
                name, _, _, _ = db.col_info(int(m["col"]))
                ti = int(m["target_idx"])
                out.write("%s\t%d\t%d\t%.4e\t%d\t%s\t%d\t%d\t%d\t%d\t%d\t%.4f\t%.4f\t%.4f\t%d\n" % (
                    qid.decode(), res.qlen[i], res.qkmers[i], m["fpr"], len(ms), name, ti & 0xFFFF, ti >> 16, int(m["gsize"]), k, int(m["mkmers"]),
                    m["qcov"], m["tcov"], m["jacc"], qidx))
        total += len(ids)

    ids, seqs, nb = [], [], 0
    for f in a.files:
        for qid, s in read_fastx(f):
            ids.append(qid)
            seqs.append(s)
            nb += len(s)
            if len(ids) >= a.gpu_batch or nb >= (64 << 20):
                flush(ids, seqs)
                ids, seqs, nb = [], [], 0
    if ids:
        flush(ids, seqs)
    if rank == 0:
        out.write(f"# input queries: {total}\n# matched queries: {matched}\n")
        out.write("# matched percentage: %s%%\n" % ("%.4f" % (matched / total * 100) if total else "NaN"))
        if out is not sys.stdout:
            out.close()
        print(f"kmcp_amd.dist_search: {world} rank(s), backend {dist.get_backend() if world > 1 else 'none'}: matches: {n_matches}, checksum {check:016x} "
              "(order-independent over (queryIdx, column, mKmers): the same on any number of GPUs, and the one kmcp-search logs)", file=sys.stderr)
    srch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
