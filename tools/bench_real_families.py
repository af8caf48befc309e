#!/usr/bin/env python3
"""Throughput, traffic and end-to-end numbers on the real-genome, hit-heavy database of tools/family_db.py:
21 000 columns (600 E. coli strains + 15 species x 100 strains, 10 chunks each; ~1.3 GB of index, column densities uneven,
relatives share sectors), 150-bp reads sampled from the strains.  For uniform_sigs 0 (the reference's per-block NumSigs) and 1
(blocks of a tier share NumSigs => grouped rows):

  * K1/K2 kernel times, bytes fetched (the kernel's own count) vs algorithmic bytes, hits per read, pruning on/off;
  * the host boundary (kmcpg_search_batch: host buffers in, finalized matches out);
  * `kmcp-search` end to end on a FASTQ file, TSV written (tens to hundreds of rows per read: the writer under load);
  * the database and reads stay under OUT_DIR for tests/test_gpu_real_families.py::test_full_size_sample (oracle diff).

usage: bench_real_families.py OUT_DIR [--ecoli-strains 600] [--small-strains 100] [--batch 131072] [--cli-reads 100000] [--modes 0,1]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import family_db  # noqa: E402
from kmcp_amd import Database, default_params, lib  # noqa: E402

CLI = os.path.join(ROOT, "kmcp_amd", "kmcp-search")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--ecoli-strains", type=int, default=600)
    ap.add_argument("--small-strains", type=int, default=100)
    ap.add_argument("--batch", type=int, default=131072)
    ap.add_argument("--cli-reads", type=int, default=100000)
    ap.add_argument("--modes", default="0,1")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    os.makedirs(a.out_dir, exist_ok=True)
    n_reads = max(a.batch, a.cli_reads)
    cols, reads, info = family_db.generate(a.ecoli_strains, a.small_strains, n_reads, log=lambda m: print(m, file=sys.stderr))
    out = {"database": info}
    fq = os.path.join(a.out_dir, "reads.fq")
    family_db.write_fastq(fq, reads[:a.cli_reads])
    B = a.batch
    h_reads = np.ascontiguousarray(reads[:B]).reshape(-1)
    h_offs = np.arange(B + 1, dtype=np.uint64) * family_db.READ_LEN
    params = default_params()
    for mode in [int(x) for x in a.modes.split(",")]:
        d = os.path.join(a.out_dir, f"mode{mode}")
        shutil.rmtree(d, ignore_errors=True)
        t0 = time.time()
        db_dir = lib.build_db(d, cols, k=family_db.K, threads=32, uniform_sigs=mode, alias="family-db")
        build_s = time.time() - t0
        r = {"build_s": build_s}
        t0 = time.time()
        with Database.open(db_dir, device=0) as db:
            r["load_s"] = time.time() - t0
            nb = int(db.info.n_blocks)
            bi = [db.block_info(b) for b in range(nb)]
            r.update(index_bytes=int(db.info.matrix_bytes), blocks=nb, columns=int(db.info.n_cols), distinct_num_sigs=len({b["num_sigs"] for b in bi}),
                     row_bytes=sorted({b["row_bytes"] for b in bi}), strides=sorted({b["stride"] for b in bi}),
                     num_sigs_min=min(b["num_sigs"] for b in bi), num_sigs_max=max(b["num_sigs"] for b in bi))
            d_reads = torch.from_numpy(h_reads).to(dev)
            d_offs = torch.from_numpy(h_offs.view(np.int64)).to(dev)
            cap = 1024 * B
            d_hits = torch.empty((cap, 3), dtype=torch.int32, device=dev)
            d_cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            d_qk = torch.zeros(B, dtype=torch.int32, device=dev)
            d_ql = torch.zeros(B, dtype=torch.int32, device=dev)

            def kernels(env=None, level=1, reps=3):
                old = {k_: os.environ.get(k_) for k_ in (env or {})}
                os.environ.update(env or {})
                try:
                    db.set_profiling(level)
                    ms = []
                    for _ in range(reps):
                        db.query_device(d_reads.data_ptr(), d_offs.data_ptr(), B, B * family_db.READ_LEN, family_db.READ_LEN, d_hits.data_ptr(), cap, d_cnt.data_ptr(),
                                        d_qk.data_ptr(), d_ql.data_ptr(), params=params)
                        torch.cuda.synchronize()
                        ms.append(db.last_timing())
                    g = db.last_gathered_bytes() if level == 2 else None
                finally:
                    for k_, v_ in old.items():
                        if v_ is None:
                            os.environ.pop(k_, None)
                        else:
                            os.environ[k_] = v_
                return min(m[0] for m in ms), min(m[1] for m in ms), g
            k1, k2, _ = kernels()
            n_hits = int(d_cnt[0].item())
            assert n_hits <= cap, "hit buffer too small"
            _, _, gathered = kernels(level=2, reps=1)
            _, k2_np, _ = kernels({"KMCPG_PRUNE": "0"})
            _, _, gathered_np = kernels({"KMCPG_PRUNE": "0"}, level=2, reps=1)
            kmers = int(d_qk.sum().item())
            alg = kmers * sum(b["row_bytes"] for b in bi) + B * family_db.READ_LEN + 12 * n_hits
            r.update(batch_reads=B, k1_ms=k1, k2_ms=k2, reads_per_s_kernels=B / ((k1 + k2) * 1e-3), hits=n_hits, hits_per_read=n_hits / B,
                     algorithmic_bytes=alg, gathered_bytes=gathered, gathered_over_algorithmic=gathered / alg, effective_gbps=alg / (k2 * 1e-3) / 1e9,
                     achieved_gbps=gathered / (k2 * 1e-3) / 1e9, k2_ms_prune_off=k2_np, gathered_bytes_prune_off=gathered_np,
                     gathered_over_algorithmic_prune_off=gathered_np / alg, achieved_gbps_prune_off=gathered_np / (k2_np * 1e-3) / 1e9)
            # K3 alone: the hit list of the last launch grouped by read, filtered, ordered on the device
            d_pairs = torch.empty((n_hits + 16, 2), dtype=torch.int32, device=dev)
            d_roffs = torch.zeros(B + 2, dtype=torch.int64, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k3 = []
            for _ in range(3):
                e0.record()
                db.group_device(d_hits.data_ptr(), d_cnt.data_ptr(), cap, d_qk.data_ptr(), B, d_pairs.data_ptr(), d_roffs.data_ptr(), params=params,
                                stream=torch.cuda.current_stream(dev).cuda_stream)
                e1.record()
                torch.cuda.synchronize()
                k3.append(e0.elapsed_time(e1))
            r.update(k3_ms=min(k3), k3_kept=int(d_roffs[B].item()))
            del d_hits, d_pairs
            torch.cuda.empty_cache()
            db.search_packed_count(h_reads, h_offs, params=params)  # sizes staging and hit buffers
            ts = []
            for _ in range(3):
                t0 = time.time()
                nm = db.search_packed_count(h_reads, h_offs, params=params)
                ts.append(time.time() - t0)
            dt = min(ts)
            r.update(search_batch_s=dt, search_batch_reads_per_s=B / dt, search_batch_s_all=ts, matches=nm, matches_per_read=nm / B)
            # compact results (round 5): the final matches as 8-byte pairs, no 56-byte records written
            db.search_packed_pairs(h_reads, h_offs, params=params, count_only=True)
            ts = []
            for _ in range(3):
                t0 = time.time()
                nmp = db.search_packed_pairs(h_reads, h_offs, params=params, count_only=True)
                ts.append(time.time() - t0)
            assert nmp == nm, (nmp, nm)
            r.update(search_batch_pairs_s=min(ts), search_batch_pairs_reads_per_s=B / min(ts), search_batch_pairs_s_all=ts)
            # ... and pipelined: batches through kmcpg_submit / kmcpg_wait from two host threads (what kmcp-search does)
            import threading
            NB = 8

            def pump(t_):
                tk = []
                for i in range(t_, NB, 2):
                    if len(tk) == 2:
                        db.wait(tk.pop(0), count_only=True)
                    tk.append(db.submit(h_reads, h_offs, params=params))
                while tk:
                    db.wait(tk.pop(0), count_only=True)
            th = [threading.Thread(target=pump, args=(t_,)) for t_ in range(2)]
            t0 = time.time()
            [x.start() for x in th]
            [x.join() for x in th]
            dtp = (time.time() - t0) / NB
            r.update(pipelined_s_per_batch=dtp, pipelined_reads_per_s=B / dtp)

            def pump_pairs(t_):
                tk = []
                for i in range(t_, NB, 2):
                    if len(tk) == 2:
                        db.wait_pairs(tk.pop(0), count_only=True)
                    tk.append(db.submit(h_reads, h_offs, params=params))
                while tk:
                    db.wait_pairs(tk.pop(0), count_only=True)
            th = [threading.Thread(target=pump_pairs, args=(t_,)) for t_ in range(2)]
            t0 = time.time()
            [x.start() for x in th]
            [x.join() for x in th]
            dtp = (time.time() - t0) / NB
            r.update(pipelined_pairs_s_per_batch=dtp, pipelined_pairs_reads_per_s=B / dtp)
        # the round-3 host half on the same batch (KMCPG_DEVICE_FINALIZE=0 is read when a handle makes its first search)
        os.environ["KMCPG_DEVICE_FINALIZE"] = "0"
        try:
            with Database.open(db_dir, device=0) as db2:
                db2.search_packed_count(h_reads, h_offs, params=params)
                ts = []
                for _ in range(3):
                    t0 = time.time()
                    nm2 = db2.search_packed_count(h_reads, h_offs, params=params)
                    ts.append(time.time() - t0)
                assert nm2 == nm
                r.update(search_batch_host_finalize_s=min(ts), search_batch_host_finalize_reads_per_s=B / min(ts))
        finally:
            os.environ.pop("KMCPG_DEVICE_FINALIZE", None)
        # end to end through the CLI: FASTQ in, TSV out
        tsv = os.path.join(a.out_dir, f"mode{mode}.tsv")
        t0 = time.time()
        p = subprocess.run([CLI, "-d", d, fq, "-o", tsv], capture_output=True, text=True)
        dt = time.time() - t0
        assert p.returncode == 0, p.stderr
        rows = sum(1 for _ in open(tsv)) - 4
        pipe = [ln for ln in p.stderr.split("\n") if "pipeline:" in ln]
        # the same without a file system behind the writer: formatting + pipeline alone
        t0 = time.time()
        p2 = subprocess.run([CLI, "-d", d, fq, "-o", "/dev/null"], capture_output=True, text=True)
        dt2 = time.time() - t0
        assert p2.returncode == 0, p2.stderr
        pipe2 = [ln for ln in p2.stderr.split("\n") if "pipeline:" in ln]
        r["cli"] = dict(reads=a.cli_reads, wall_s=dt, reads_per_s=a.cli_reads / dt, tsv_bytes=os.path.getsize(tsv), rows=rows, rows_per_read=rows / a.cli_reads,
                        rows_per_s=rows / dt, pipeline=pipe[-1] if pipe else "", to_dev_null=dict(wall_s=dt2, rows_per_s=rows / dt2, pipeline=pipe2[-1] if pipe2 else ""))
        out[f"uniform_sigs={mode}"] = r
        print(json.dumps({f"uniform_sigs={mode}": r}), file=sys.stderr)
    info["db_dirs"] = {m: os.path.join(a.out_dir, f"mode{m}", "R001") for m in a.modes.split(",")}
    json.dump(info, open(os.path.join(a.out_dir, "family_db.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
