#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the reference tree (run in the build container only).

Inputs : /root/reference/demo-searching/refs/*.fasta.gz (the reference's demo genomes) and the two result tables
         published in /root/reference/demo-searching/README.md:61-68 (Closed Syncmer) and :102-109 (FracMinHash).
Outputs: demo_searching_sketches.npz  sorted-unique sketch hashes of the 9 genomes for both sketch modes, produced
                                      by the oracle's restatement of `kmcp compute` (data, not source)
         demo_searching_tables.json   the published tables (verbatim values) + accession mapping + intermediate KATs
         NC_018658.1.fasta.gz         the query genome of the demo (data file of the reference's worked example)
"""
import glob
import gzip
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

REF = "/root/reference/demo-searching"


def read_fasta_gz(path):
    seqs, cur, name = [], [], None
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if line[0] == ">":
                if name is not None:
                    seqs.append((name, "".join(cur).encode()))
                name, cur = line[1:].strip(), []
            else:
                cur.append(line.strip())
    seqs.append((name, "".join(cur).encode()))
    return seqs


TABLES = {
    # target accession: (qCov, tCov, jacc) as printed in the README (4 decimals)
    "minhash": {"cmd": "compute -k 31 --scale 1000 -B plasmid; index -n 3 -f 0.01; search -g -t 0.5 -n 0 -s jacc",
                "k": 31, "scale": 1000, "syncmer_s": 0,
                "rows": [["NC_018658.1", "1.0000", "1.0000", "1.0000"], ["NZ_CP028116.1", "0.7499", "0.7234", "0.5828"],
                         ["NC_000913.3", "0.6064", "0.6833", "0.4734"], ["NC_012971.2", "0.5965", "0.6893", "0.4701"],
                         ["NZ_CP007592.1", "0.5852", "0.5958", "0.4189"], ["NC_002695.2", "0.5527", "0.5383", "0.3750"]]},
    "syncmer": {"cmd": "compute -k 31 --syncmer-s 15 --scale 62 -B plasmid; index -n 3 -f 0.01; search -g -t 0.5 -n 0 -s jacc",
                "k": 31, "scale": 62, "syncmer_s": 15,
                "rows": [["NC_018658.1", "1.0000", "1.0000", "1.0000"], ["NZ_CP028116.1", "0.7439", "0.7189", "0.5763"],
                         ["NC_000913.3", "0.6041", "0.6768", "0.4688"], ["NC_012971.2", "0.5972", "0.6807", "0.4665"],
                         ["NZ_CP007592.1", "0.5782", "0.5868", "0.4109"], ["NC_002695.2", "0.5482", "0.5322", "0.3699"]]},
}


def main():
    files = sorted(glob.glob(os.path.join(REF, "refs", "*.fasta.gz")))
    arrays, meta = {}, {"genomes": {}, "tables": TABLES}
    for f in files:
        acc = os.path.basename(f)[:-len(".fasta.gz")]
        recs = [(n, s) for n, s in read_fasta_gz(f) if "plasmid" not in n]  # --seq-name-filter plasmid
        meta["genomes"][acc] = {"gsize": sum(len(s) for _, s in recs), "header": recs[0][0]}
        for mode, t in TABLES.items():
            cfg = O.sketch_cfg(k=t["k"], scale=t["scale"], syncmer_s=t["syncmer_s"])
            h = O.sort_unique(np.concatenate([O.generate_kmers(s, cfg) for _, s in recs]))
            arrays[f"{mode}:{acc}"] = h
            meta["genomes"][acc][f"{mode}_kmers"] = int(len(h))
    q = read_fasta_gz(os.path.join(REF, "refs", "NC_018658.1.fasta.gz"))[0][1]
    meta["kats"] = {
        "nthash_k21_ACGTx": ["ACGTACGTACGTACGTACGTA", "0x6f6b1ff54c38ed32"],
        "nthash_k21_A21": ["A" * 21, "0xc6573ed1093f7306"],
        "nthash_k31_GATTACA": ["GATTACAGATTACAGATTACAGATTACAGAT", "0x1e29c43bd1173d27"],
        "max_hash": {"1000": 18446744073709552, "62": 297528130221121792},
        "calc_signature_size": [[10339, 3, 0.01, 127834], [10439, 3, 0.01, 129070], [400000, 1, 0.3, 1121470]],
        "query_first150_k21_xor": hex(int(np.bitwise_xor.reduce(O.nthash_all(q[:150], 21)))),
    }
    np.savez_compressed(os.path.join(HERE, "demo_searching_sketches.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "demo_searching_tables.json"), "w"), indent=1)
    shutil.copyfile(os.path.join(REF, "refs", "NC_018658.1.fasta.gz"), os.path.join(HERE, "NC_018658.1.fasta.gz"))
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
