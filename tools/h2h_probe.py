#!/usr/bin/env python3
"""Host-to-host pump of one bench workload through kmcpg_submit / kmcpg_submit_packed, alone (for rocprofv3 traces and A/B runs):
   python tools/h2h_probe.py config2_genome_search [--packed] [--threads 2] [--depth 2] [--batches 16]"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kmcp_amd import Database, default_params, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--packed", action="store_true")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--batch-reads", type=int, default=0)
    a = ap.parse_args()
    wl = dict(bench.WORKLOADS[a.workload])
    B = a.batch_reads or wl["batch_reads"]
    dev = torch.device("cuda", 0)
    spec = lib.SynthSpec(k=wl["k"], num_hashes=wl["num_hashes"], fpr=wl["fpr"], n_blocks=wl["n_blocks"], cols_per_block=wl["cols_per_block"],
                         num_sigs=wl["num_sigs"], kmers_per_col=wl["kmers_per_col"], seed=42, sigs_step=wl.get("sigs_step", 0), scale=wl.get("scale", 0),
                         syncmer_s=wl.get("syncmer_s", 0), minimizer_w=wl.get("minimizer_w", 0))
    db = Database.open_synthetic(spec, device=0)
    n_cols = int(db.info.n_cols)
    params = default_params()
    params.min_qcov = wl.get("min_qcov", params.min_qcov)
    params.sort_by = wl.get("sort_by", 0)

    def plant(frag, offs, n, total, maxlen, cols):
        db.plant_reads_device(frag.data_ptr(), offs.data_ptr(), n, total, maxlen, cols.data_ptr())

    batches = [bench.make_batch(dev, wl, B, n_cols, 1000 + i, plant) for i in range(2)]
    hb = [(b.reads.cpu().numpy(), b.offs.cpu().numpy().astype(np.uint64)) for b in batches]
    del batches
    torch.cuda.empty_cache()
    if a.packed:
        hbp = []
        for r_, o_ in hb:
            if os.environ.get("H2H_PINNED", "1") == "1":
                pin = lib.PinnedBytes((len(r_) + 3) // 4 + 8)
                pin.a[:] = 0
                c_, e_, _ = lib.pack2([r_], codes=pin.a)
            else:
                c_, e_, _ = lib.pack2([r_])
            hbp.append((c_, o_, e_))
        submit = lambda i: db.submit_packed(*hbp[i % 2], params=params)  # noqa: E731
    else:
        submit = lambda i: db.submit(*hb[i % 2], params=params)  # noqa: E731
    stamps = []

    def pump(t_, nb):
        tk = []
        for i in range(t_, nb, a.threads):
            if len(tk) == a.depth:
                db.wait(tk.pop(0), count_only=True)
                stamps.append(time.perf_counter())
            t0 = time.perf_counter()
            tk.append(submit(i))
            if os.environ.get("H2H_VERBOSE"):
                print(f"thread {t_}: submit of batch {i} took {1e3*(time.perf_counter()-t0):.2f} ms", file=sys.stderr)
        while tk:
            db.wait(tk.pop(0), count_only=True)
            stamps.append(time.perf_counter())

    def pumped(nb):
        th = [threading.Thread(target=pump, args=(t_, nb)) for t_ in range(a.threads)]
        t1 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        return (time.perf_counter() - t1) / nb

    pumped(2 * a.threads * a.depth)
    stamps.clear()
    dt = pumped(a.batches)
    print(f"{a.workload} {'packed' if a.packed else 'text'}: {1e3*dt:.2f} ms per batch of {B} = {B/dt:.0f} queries/s ({a.threads} threads x {a.depth} in flight)")
    db.close()


if __name__ == "__main__":
    main()
