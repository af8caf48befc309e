#!/usr/bin/env python3
"""Builds tests/golden/demo_profiling_refs_300k.fa.gz — the input of the BASELINE.json configs[0] test
(tests/test_gpu_config0.py) — from the 15 reference genomes the reference ships under demo-profiling/refs/.

Run in the build container (reads /root/reference; the GPU box only sees the committed fixture):
    python tests/golden/make_demo_profiling.py

A fixture is data: for every genome the records of its FASTA file in file order, cut after 300 kb of sequence in
total (`>GCF_xxx.N|<record id> <original description>`); if the genome has a record named "... plasmid ..." that fell
behind the cut, its first 20 kb are kept too, so that `--seq-name-filter plasmid` of the demo recipe
(demo-profiling/README.md:232-241) has something to drop.  Sequence letters are copied as they are (case, N's).
"""
import glob
import gzip
import os

REFS = "/root/reference/demo-profiling/refs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo_profiling_refs_300k.fa.gz")
KEEP = 300_000
PLASMID_KEEP = 20_000


def records(path):
    name, seq = None, []
    with gzip.open(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(seq)
                name, seq = line[1:], []
            elif line:
                seq.append(line)
    if name is not None:
        yield name, "".join(seq)


def main():
    files = sorted(glob.glob(os.path.join(REFS, "*.fa.gz")))
    assert len(files) == 15, files
    n_rec = 0
    with gzip.GzipFile(OUT, "wb", compresslevel=9, mtime=0) as gz:
        for f in files:
            acc = os.path.basename(f)[:-len(".fa.gz")]
            left, plasmid_done = KEEP, False
            for name, s in records(f):
                is_plasmid = "plasmid" in name
                if left > 0:
                    part = s[:left]
                    left -= len(part)
                elif is_plasmid and not plasmid_done:
                    part = s[:PLASMID_KEEP]
                else:
                    continue
                plasmid_done = plasmid_done or is_plasmid
                gz.write(f">{acc}|{name}\n".encode())
                for i in range(0, len(part), 80):
                    gz.write(part[i:i + 80].encode() + b"\n")
                n_rec += 1
    print(OUT, os.path.getsize(OUT), "bytes,", n_rec, "records")


if __name__ == "__main__":
    main()
