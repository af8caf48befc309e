import os, subprocess, sys, gzip
import numpy as np
sys.path.insert(0, '.')
from kmcp_amd.dist_search import read_fastx
BIN = sys.argv[1]
N = int(sys.argv[2])
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
d = 'scratch/fuzzfmt/rd'; os.makedirs(d, exist_ok=True)
def fnv(recs):
    total = 0
    for idx, (i, s) in enumerate(recs):
        h = 1469598103934665603
        for b in i + b"\t" + s + b"\n":
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        total = (total + h * (2 * idx + 1)) & 0xFFFFFFFFFFFFFFFF
    return total
def rand_seq(n):
    return bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), n))
bad = 0
for it in range(N):
    fastq = rng.random() < 0.6
    nrec = int(rng.integers(0, 40))
    nl = b"\r\n" if rng.random() < 0.3 else b"\n"
    out = bytearray()
    for r in range(nrec):
        L = int(rng.choice([0, 1, 5, 30, 100, 150, 151, 400, 3000]))
        s = rand_seq(L)
        name = b"r%d" % r + (b" desc text" if rng.random() < 0.5 else b"") 
        wrap = int(rng.choice([0, 0, 7, 60])) 
        def lines(x):
            if wrap == 0 or len(x) == 0: return x + nl
            return b"".join(x[p:p+wrap] + nl for p in range(0, len(x), wrap))
        if fastq:
            q = bytes(rng.choice(np.frombuffer(b"@+I#5>", dtype=np.uint8), L))
            out += b"@" + name + nl + lines(s) + b"+" + (name if rng.random() < 0.3 else b"") + nl + lines(q)
        else:
            out += b">" + name + nl + lines(s)
        if rng.random() < 0.1: out += nl
    if out and rng.random() < 0.3:
        while out and out[-1:] in (b"\n", b"\r"): out = out[:-1]
    p = f"{d}/f{it}." + ("fq" if fastq else "fa")
    gz = rng.random() < 0.2
    if gz:
        p += ".gz"
        with gzip.open(p, "wb") as fh: fh.write(bytes(out))
    else:
        open(p, "wb").write(bytes(out))
    want = list(read_fastx(p))
    env = dict(os.environ, KMCP_READER_BUF=str(int(rng.choice([16, 17, 31, 64, 100, 257, 4096, 1 << 20]))))
    r = subprocess.run([BIN, "--parse-only", p], capture_output=True, text=True, env=env, timeout=60)
    if r.returncode != 0:
        print("CRASH", p, env["KMCP_READER_BUF"], r.stderr[-800:]); bad += 1; break
    f = dict(x.split("=") for x in r.stdout.strip().split("\t")[1:])
    exp = dict(records=str(len(want)), bases=str(sum(len(s) for _, s in want)), id_bytes=str(sum(len(i) for i, _ in want)), fnv1a="%016x" % fnv(want))
    if f != exp:
        print("MISMATCH", p, env["KMCP_READER_BUF"], f, exp); bad += 1; break
    os.remove(p)
print("done", N, "bad", bad)
