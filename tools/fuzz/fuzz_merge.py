import os, subprocess, sys
import numpy as np
BIN = sys.argv[1]; N = int(sys.argv[2]); rng = np.random.default_rng(5)
d = 'scratch/fuzzfmt/mg'; os.makedirs(d, exist_ok=True)
def row(q, idx, t, sc): return f"{q}\t150\t130\t1.0e-05\t9\t{t}\t0\t1\t1000\t21\t80\t{sc}\t0.1000\t0.0900\t{idx}\n"
H = "#query\tqLen\tqKmers\tFPR\thits\ttarget\tchunkIdx\tchunks\ttLen\tkSize\tmKmers\tqCov\ttCov\tjacc\tqueryIdx\n"
def mk():
    rows = "".join(row(f"r{i}", i, f"T{j}", "%.4f" % rng.random()) for i in sorted(rng.choice(60, 20, replace=False)) for j in range(int(rng.integers(1, 4))))
    return (H + rows + "# input queries: 60\n# matched queries: 20\n# matched percentage: 33.3333%\n").encode()
bad = 0
for it in range(N):
    files = []
    for k in range(int(rng.integers(2, 4))):
        b = bytearray(mk())
        for _ in range(int(rng.integers(0, 12))):
            m = rng.integers(0, 3)
            if m == 0: b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif m == 1:
                p = int(rng.integers(0, len(b))); del b[p:p + int(rng.integers(1, 40))]
            else:
                p = int(rng.integers(0, len(b))); b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 10)), dtype=np.uint8))
            if not b: b = bytearray(b"x")
        p = f"{d}/m{it}_{k}.tsv"; open(p, "wb").write(bytes(b)); files.append(p)
    r = subprocess.run([BIN, "-o", f"{d}/out.tsv"] + files, capture_output=True, timeout=60); r.stderr = r.stderr.decode(errors="replace")
    if r.returncode not in (0, 255) or "Sanitizer" in r.stderr or "runtime error" in r.stderr:
        print("CRASH", files, r.returncode, r.stderr[-1500:]); bad += 1; break
    for p in files: os.remove(p)
print("merge fuzz done", N, "bad", bad)
