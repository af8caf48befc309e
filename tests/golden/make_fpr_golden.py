#!/usr/bin/env python3
"""Lifts the FPR column of the reference's own demo result into a fixture (data only: numbers from a table).

Source: /root/reference/docs/tutorial/profiling/index.md:199-211 — nine `kmcp search` rows of the v0.9-era tutorial
(150-bp reads, k = 21 => qKmers = 130, database built with the default `kmcp index -f 0.3 -n 1`), the only place in the
reference tree that prints Theorem-2 query FPRs (util-fpr.go:32-71) together with the (qKmers, mKmers) they belong to.
(docs/tutorial/searching/index.md:113-119 is older: v0.8.0's Chernoff bound)

Run in the build container (the reference does not exist on the GPU box):
    python tests/golden/make_fpr_golden.py  ->  tests/golden/tutorial_profiling_fpr.json
"""
import json
import os

SRC = "/root/reference/docs/tutorial/profiling/index.md"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    lines = open(SRC).read().splitlines()
    head = next(i for i, l in enumerate(lines) if l.startswith("|#query") and "FPR" in l)
    cols = [c.strip() for c in lines[head].strip("|").split("|")]
    rows = []
    for l in lines[head + 2:]:
        if not l.startswith("|"):
            break
        f = dict(zip(cols, [c.strip() for c in l.strip("|").split("|")]))
        rows.append({"query": f["#query"], "qLen": int(f["qLen"]), "qKmers": int(f["qKmers"]), "mKmers": int(f["mKmers"]), "FPR": f["FPR"],
                     "qCov": f["qCov"], "kSize": int(f["kSize"])})
    out = {"source": "docs/tutorial/profiling/index.md:%d-%d (kmcp v0.9.x demo result)" % (head + 3, head + 2 + len(rows)),
           "db_fpr": 0.3, "format": "%.4e (strconv.FormatFloat(fpr, 'e', 4, 64), search.go:539)", "rows": rows}
    json.dump(out, open(os.path.join(HERE, "tutorial_profiling_fpr.json"), "w"), indent=1)
    print(len(rows), "rows")


if __name__ == "__main__":
    main()
