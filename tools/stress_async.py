#!/usr/bin/env python3
"""Stress of the asynchronous boundary: several host threads hammer ONE handle with kmcpg_submit / kmcpg_wait /
kmcpg_search_batch in random patterns (batch sizes 0 .. 3000, single-end, paired-end with --try-se, up to three tickets held per
thread, waits out of order, KMCPG_EBUSY handled by waiting), on a single-GPU handle, on an in-process multi-device handle
(two shards on GPU 0: host merge), on one whose hit lists go through the RCCL exchange, and on a paged handle (3 passes per batch).
Every result must equal the one the same batch gives when it is searched alone.

usage: stress_async.py [seconds=30] [threads=6]
"""
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmcp_amd import Database, default_params, lib  # noqa: E402
from tests import synth  # noqa: E402


def key(res):
    return (res.qlen.tobytes(), res.qkmers.tobytes(), res.ksize.tobytes(), res.offs.tobytes(), res.matches.tobytes())


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rng = np.random.default_rng(1)
    with tempfile.TemporaryDirectory() as tmp:
        genomes = synth.random_genomes(40, 12000, seed=3)
        db_dir = synth.make_db(tmp, genomes, k=21, n_chunks=2, threads=8)
        # a pool of batches with their expected results (searched alone, nothing else in flight)
        pool = []
        with Database.open(db_dir) as db:
            for i in range(40):
                n = int(rng.choice([0, 1, 7, 200, 1000, 3000]))
                paired = i % 3 == 0
                r1 = synth.sample_reads(genomes, n, 150, seed=100 + i, frac_random=0.3) if n else []
                r2 = None
                p = default_params()
                if paired:
                    r2 = [synth.sample_reads(genomes, 1, 150, seed=5000 + 7 * i + j, frac_random=0.6)[0] if j % 4 else b"ACGT" for j in range(n)]
                    p = default_params(try_se=1)
                s1, o1 = lib.pack_reads(r1)
                s2, o2 = lib.pack_reads(r2) if paired else (None, None)
                pool.append((s1, o1, s2, o2, p, key(db.search_packed(s1, o1, s2, o2, p))))
        def open_rccl():
            os.environ["KMCPG_RCCL"] = "force"  # one GPU here: the RCCL exchange with a one-rank communicator
            try:
                db = Database.open_devices(db_dir, [0])
            finally:
                os.environ.pop("KMCPG_RCCL", None)
            assert db.exchange_info().startswith("RCCL gather"), db.exchange_info()
            return db

        openers = (("one GPU", lambda: Database.open(db_dir)), ("two shards in one process", lambda: Database.open_devices(db_dir, [0, 0])),
                   ("RCCL exchange of an in-process multi-GPU handle (one rank)", open_rccl), ("paged in 3 passes", lambda: Database.open_paged(db_dir, 0, 3)))
        for label, opener in openers:
            db = opener()
            stop = time.time() + seconds / len(openers)
            errors, done, busy = [], [0] * n_threads, [0] * n_threads

            def worker(t):
                r = np.random.default_rng(100 + t)
                held = []
                try:
                    while time.time() < stop and not errors:
                        act = r.random()
                        if act < 0.5 and len(held) < 3:
                            b = pool[int(r.integers(0, len(pool)))]
                            try:
                                held.append((db.submit(b[0], b[1], b[2], b[3], params=b[4]), b[5]))
                            except lib.KmcpGpuError as e:
                                if e.code != -7:
                                    raise
                                busy[t] += 1
                                if held:
                                    tk, want = held.pop(int(r.integers(0, len(held))))
                                    assert key(db.wait(tk)) == want
                                    done[t] += 1
                        elif act < 0.8 and held:
                            tk, want = held.pop(int(r.integers(0, len(held))))
                            assert key(db.wait(tk)) == want
                            done[t] += 1
                        elif not held:  # search_batch waits for a lane: only safe while this thread holds none
                            b = pool[int(r.integers(0, len(pool)))]
                            assert key(db.search_packed(b[0], b[1], b[2], b[3], b[4])) == b[5]
                            done[t] += 1
                    for tk, want in held:
                        assert key(db.wait(tk)) == want
                        done[t] += 1
                except Exception as e:  # noqa: BLE001
                    errors.append(repr(e))
                    for tk, _ in held:
                        try:
                            db.wait(tk)
                        except Exception:  # noqa: BLE001
                            pass

            th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
            [x.start() for x in th]
            [x.join() for x in th]
            db.close()
            print(f"{label}: {sum(done)} batches checked on {n_threads} threads, {sum(busy)} x EBUSY, errors: {errors[:2]}")
            if errors:
                sys.exit(1)
    print("stress ok")


if __name__ == "__main__":
    main()
