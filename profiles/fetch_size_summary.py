#!/usr/bin/env python3
"""FETCH_SIZE per launch of the k2_cobs kernels from a *_pmc_pmc.txt written by extract_rocprof.py: the median over the launches of every
template instance (synthetic-index set-up launches are not kmcpg kernels and are not in the file), in KB and in bytes for both
multipliers (x 1024 x 2: 16-B-per-lane streaming rows; x 1024 x 1: one 64-byte request per narrow row).  usage: fetch_size_summary.py <file> ..."""
import collections
import statistics
import sys

for f in sys.argv[1:]:
    acc = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    hdr = None
    for ln in open(f):
        if ln.startswith("#"):
            continue
        p = ln.rstrip("\n").split("\t")
        if hdr is None:
            hdr = p
            continue
        d = dict(zip(hdr, p))
        kn = d.get("kernel_name") or d.get("name")
        if "k2_cobs" not in kn or d["counter_name"] != "FETCH_SIZE":
            continue
        acc[kn].append(float(d["value"]))
        if "start" in d:
            dur[kn].append(float(d["end"]) - float(d["start"]))
    print(f)
    for kn, v in acc.items():
        # the timed launches of a bench run are the large ones: drop launches below half of the maximum (sanity / parity launches on small batches)
        big = [x for x in v if x >= 0.5 * max(v)]
        m = statistics.median(big)
        dm = statistics.median(dur[kn][-len(big):]) / 1e6 if dur[kn] else float("nan")
        print(f"  {kn[:60]:60s} launches {len(big):2d}/{len(v):2d}  FETCH_SIZE median {m:14.1f} KB  x1024x2 = {m*2048/1e9:9.2f} GB  x1024x1 = {m*1024/1e9:9.2f} GB  (duration under the profiler {dm:.3f} ms)")
