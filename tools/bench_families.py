#!/usr/bin/env python3
"""Hit-heavy variant of the headline workload: every planted fragment sits in F columns (a "species" of F close relatives,
as GTDB has for E. coli and friends) instead of one, so a read produces ~F hits, F sectors per read never die and the host half /
hit read-back carry F times the load.  GTDB-scale synthetic index, 150-bp reads; prints kernel time, device-resident rate and
the host-boundary rate (kmcpg_submit / kmcpg_wait, two host threads) for a few F.

usage: bench_families.py [F ...]   (default 1 8 64)
"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from kmcp_amd import Database, default_params, lib  # noqa: E402


def main():
    fams = [int(x) for x in sys.argv[1:]] or [1, 8, 64]
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS["gtdb"]
    B = 262144
    out = {}
    for F in fams:
        spec = lib.SynthSpec(k=wl["k"], num_hashes=1, fpr=wl["fpr"], n_blocks=wl["n_blocks"], cols_per_block=wl["cols_per_block"], num_sigs=wl["num_sigs"],
                             kmers_per_col=wl["kmers_per_col"], seed=42, sigs_step=wl["sigs_step"])
        db = Database.open_synthetic(spec, device=0)
        n_cols = int(db.info.n_cols)
        frag, cols, reads, offs = bench.make_batch(dev, B, n_cols, seed=77)
        for j in range(F):  # the relatives of a species sit in neighbouring chunks / genomes: columns c, c+1, ...
            cj = torch.where(cols >= 0, (cols + j) % n_cols, cols).contiguous()
            db.plant_reads_device(frag.data_ptr(), offs.data_ptr(), B, B * bench.READ_LEN, bench.READ_LEN, cj.data_ptr())
        torch.cuda.synchronize()
        params = default_params()
        db.set_profiling(True)
        cap = (F + 4) * B + 4096
        d_hits = torch.empty((cap, 3), dtype=torch.int32, device=dev)
        d_cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        d_qk = torch.zeros(B, dtype=torch.int32, device=dev)
        d_ql = torch.zeros(B, dtype=torch.int32, device=dev)
        ms = []
        for _ in range(3):
            db.query_device(reads.data_ptr(), offs.data_ptr(), B, B * bench.READ_LEN, bench.READ_LEN, d_hits.data_ptr(), cap, d_cnt.data_ptr(), d_qk.data_ptr(),
                            d_ql.data_ptr(), params=params)
            torch.cuda.synchronize()
            ms.append(db.last_timing()[1])
        n_hits = int(d_cnt[0].item())
        # what the kernel fetched (it counts its own row loads at profiling level 2) beside the algorithmic bytes
        db.set_profiling(2)
        db.query_device(reads.data_ptr(), offs.data_ptr(), B, B * bench.READ_LEN, bench.READ_LEN, d_hits.data_ptr(), cap, d_cnt.data_ptr(), d_qk.data_ptr(),
                        d_ql.data_ptr(), params=params)
        torch.cuda.synchronize()
        gathered = db.last_gathered_bytes()
        db.set_profiling(True)
        alg = int(d_qk.sum().item()) * int(db.info.row_bytes_sum_local) + B * bench.READ_LEN + 12 * n_hits
        h_reads = reads.cpu().numpy()
        h_offs = offs.cpu().numpy().astype(np.uint64)
        db.search_packed_count(h_reads, h_offs, params=params)
        NB, HT = int(os.environ.get("NB", "6")), 2

        def pump(t):
            tk = []
            for _ in range(t, NB, HT):
                if len(tk) == 2:
                    db.wait(tk.pop(0), count_only=True)
                tk.append(db.submit(h_reads, h_offs, params=params))
            while tk:
                db.wait(tk.pop(0), count_only=True)
        th = [threading.Thread(target=pump, args=(t,)) for t in range(HT)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        dt = (time.perf_counter() - t0) / NB
        out[f"F={F}"] = dict(hits_per_read=n_hits / B, k2_ms=min(ms), device_reads_per_s=B / (min(ms) * 1e-3), host_boundary_reads_per_s=B / dt,
                             host_boundary_ms_per_batch=dt * 1e3, gathered_bytes=gathered, algorithmic_bytes=alg, gathered_over_algorithmic=gathered / alg,
                             achieved_gbps=gathered / (min(ms) * 1e-3) / 1e9)
        db.close()
        del d_hits
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
