"""K2 time / traffic / tail-mode counters of a long-query workload under different KMCPG_TAIL_* settings (one process, one index)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from kmcp_amd import Database, default_params, lib  # noqa: E402


def main(name, settings):
    dev = torch.device("cuda:0")
    wl = dict(bench.WORKLOADS[name])
    B = wl["batch_reads"]
    spec = lib.SynthSpec(k=wl["k"], num_hashes=wl["num_hashes"], fpr=wl["fpr"], n_blocks=wl["n_blocks"], cols_per_block=wl["cols_per_block"],
                         num_sigs=wl["num_sigs"], kmers_per_col=wl["kmers_per_col"], seed=42, sigs_step=wl.get("sigs_step", 0), scale=wl.get("scale", 0),
                         syncmer_s=wl.get("syncmer_s", 0), minimizer_w=wl.get("minimizer_w", 0))
    db = Database.open_synthetic(spec, device=0)
    params = default_params()
    params.min_qcov = wl.get("min_qcov", params.min_qcov)
    params.sort_by = wl.get("sort_by", 0)

    def plant(frag, offs, n, total, maxlen, cols):
        db.plant_reads_device(frag.data_ptr(), offs.data_ptr(), n, total, maxlen, cols.data_ptr())

    bt = bench.make_batch(dev, wl, B, int(db.info.n_cols), 1000, plant)
    cap = (4 + 2 * int(wl.get("relatives", 1))) * B + 4096
    hits = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    qk = torch.zeros(B, dtype=torch.int32, device=dev)
    ql = torch.zeros(B, dtype=torch.int32, device=dev)
    db.set_profiling(2)
    ref = None
    for st in settings:
        env = dict(kv.split("=") for kv in st.split(",") if kv)
        for k, v in env.items():
            os.environ[k] = v
        ms = []
        for it in range(6):
            cnt.zero_()
            db.query_device(bt.reads.data_ptr(), bt.offs.data_ptr(), B, bt.total, bt.maxlen, hits.data_ptr(), cap, cnt.data_ptr(), qk.data_ptr(), ql.data_ptr(), params=params)
            torch.cuda.synchronize()
            ms.append(db.last_timing()[1])
        m = int(cnt[0].item())
        h = hits[:m].cpu().numpy()
        import numpy as np
        h = h[np.lexsort((h[:, 1], h[:, 0]))]
        if ref is None:
            ref = h
        print("%-14s %-44s k2 %.3f ms (min %.3f)  rows %.4g B  hashes %.4g B  tail waves %d  hits %d same %s" % (
            name[:14], st, sorted(ms)[len(ms) // 2], min(ms), db.last_gathered_bytes(), db.last_hash_bytes(), db.last_tail_waves(), m,
            bool(h.shape == ref.shape and (h == ref).all())), flush=True)
        for k in env:
            os.environ.pop(k, None)
    db.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
