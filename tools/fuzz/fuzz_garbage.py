import os, subprocess, sys
import numpy as np
BIN = sys.argv[1]; N = int(sys.argv[2]); rng = np.random.default_rng(11)
d = 'scratch/fuzzfmt/gb'; os.makedirs(d, exist_ok=True)
base = b"".join(b"@r%d d\nACGTACGTNNACGT\n+\nIIIIIIIIIIIIII\n" % i for i in range(30)) + b"".join(b">s%d\nACGTAC\nGGT\n" % i for i in range(10))
bad = 0
for it in range(N):
    b = bytearray(base)
    for _ in range(int(rng.integers(1, 30))):
        m = rng.integers(0, 4)
        if m == 0: b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif m == 1:
            p = int(rng.integers(0, len(b))); del b[p:p + int(rng.integers(1, 50))]
        elif m == 2:
            p = int(rng.integers(0, len(b))); b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))
        else:
            p = int(rng.integers(0, len(b))); b[p:p] = b"\n" * int(rng.integers(1, 4))
        if not b: b = bytearray(b"@")
    p = f"{d}/g{it}"
    open(p, "wb").write(bytes(b))
    env = dict(os.environ, KMCP_READER_BUF=str(int(rng.choice([16, 33, 100, 4096]))))
    r = subprocess.run([BIN, "--parse-only", p], capture_output=True, text=True, env=env, timeout=60)
    if r.returncode != 0:
        print("CRASH", p, r.stderr[-1200:]); bad += 1; break
    os.remove(p)
print("garbage done", N, "bad", bad)
