"""The N > 1 code paths, executed where only one GPU is visible: two ranks share GPU 0 (each holds its shard of the blocks),
the exchange runs over gloo because RCCL refuses two ranks on one device.  When the box shows two or more GPUs the same
launches run one rank per GPU over nccl (= RCCL over xGMI), the path the driver's scaling bench takes.

  * bench.py --gpus 2 at the headline (GTDB-scale) shape: the merged hit list of two half-index ranks is the one-rank hit list;
  * kmcp_amd.dist.ShardedSearcher's overflow loop (all-reduce of the largest per-rank hit count, rerun with room for every hit)
    through python -m kmcp_amd.dist_search, against the oracle's TSV."""
import json
import os
import subprocess
import sys

import pytest

from tests import synth
from tests.test_gpu_cli import compare, oracle_tsv, write_fastq

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(same_gpu_var):
    import torch
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    multi = torch.cuda.device_count() >= 2
    if not multi:
        env[same_gpu_var] = "1"
    return env, multi


def _free_port():
    """A TCP port nobody listens on right now (tests of this file may run side by side under pytest-xdist)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _launch(n, port, args, env):
    port = port or _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    if r.returncode != 0:  # the ranks' tracebacks sit in the middle of torchrun's stderr: keep all of it where gpurun brings it back
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "launch_failure.txt"), "w") as fh:
            fh.write(" ".join(cmd) + "\n" + r.stdout + "\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return r


def _bench_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_equal_one_rank():
    import torch
    free_b, _ = torch.cuda.mem_get_info(0)
    if free_b < 140e9:
        pytest.skip("needs 140 GB of free HBM (58 GB index once for the one-rank run, once split over the two ranks)")
    env, multi = _env("KMCP_BENCH_SAME_GPU")
    args = ["--steps", "2", "--warmup", "1", "--batch-reads", "16384", "--no-cpu-baseline", "--no-secondary"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    j1 = _bench_line(one.stdout)
    # N = 2 started exactly as N = 1 is: plain `python bench.py --gpus 2` re-executes itself under torch.distributed.run
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, (two.stdout[-2000:], two.stderr[-4000:])
    assert len(one.stdout) < 8000 and len(two.stdout) < 8000  # the driver keeps an 8 KB tail of stdout
    j2 = _bench_line(two.stdout)
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    assert j2["config"]["parallelism"] == "block-shard x2"
    assert j2["config"]["index_bytes"] == j1["config"]["index_bytes"]
    assert abs(j2["config"]["index_bytes_this_rank"] * 2 - j1["config"]["index_bytes"]) < 0.02 * j1["config"]["index_bytes"]
    # the union of the two ranks' hit lists, gathered on rank 0 and finalized there, is the one-rank result
    assert j2["hits_per_step"] == j1["hits_per_step"] > 10000
    assert j2["matches_per_step"] == j1["matches_per_step"] > 10000
    assert j2["planted_recall"] == j1["planted_recall"] > 0.99
    # the N-invariant of the bench line: the merged hit list of one batch, as an order-independent checksum
    assert j2["sanity_batch"]["hits_checksum"] == j1["sanity_batch"]["hits_checksum"] and j2["sanity_batch"]["hits"] == j1["sanity_batch"]["hits"] > 10000
    assert j2["ranks"]["world_size"] == 2 and j2["ranks"]["ranks_reporting"] == [0, 1]
    assert j2["ranks"]["backend"] == ("nccl" if multi else "gloo")
    assert j2["roofline"]["algorithmic_bytes_per_launch"] * 2 == pytest.approx(j1["roofline"]["algorithmic_bytes_per_launch"], rel=0.02)
    assert j2["value"] > 0 and "value_host_to_host" not in j2  # per-rank extras ride along only at N = 1
    assert ("nccl" if multi else "gloo")  # which exchange ran is decided by the GPUs visible; both go through gather_hits
    # the same N > 1 code path of bench.py over RCCL itself: a one-rank nccl group (all_gather_into_tensor, gather, barrier,
    # all_reduce on device tensors) must reproduce the plain one-rank numbers
    env_f = dict(env, KMCP_BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()))
    env_f.pop("KMCP_BENCH_SAME_GPU", None)
    forced = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + args, capture_output=True, text=True, timeout=900, env=env_f, cwd=ROOT)
    assert forced.returncode == 0, forced.stderr[-3000:]
    jf = _bench_line(forced.stdout)
    assert jf["hits_per_step"] == j1["hits_per_step"] and jf["matches_per_step"] == j1["matches_per_step"] and jf["planted_recall"] == j1["planted_recall"]
    assert jf["sanity_batch"]["hits_checksum"] == j1["sanity_batch"]["hits_checksum"] and jf["ranks"]["backend"] == "nccl"


def test_bench_two_ranks_hit_list_equals_the_cpu_oracle():
    """At N > 1 rank 0 fetches the rows of every rank's shard (owner reads them back, send/recv to rank 0), runs the CPU oracle
    on a sample of the batch and compares the MERGED multi-GPU hit list with it (bench.py `parity_at_n`): the first run on real
    multi-GPU hardware checks itself.  Here: BASELINE configs[1] (32 blocks, 1.4 GB) on two ranks."""
    env, multi = _env("KMCP_BENCH_SAME_GPU")
    args = ["--steps", "2", "--warmup", "1", "--workload", "config1", "--batch-reads", "32768", "--no-secondary", "--no-extras"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    j1 = _bench_line(one.stdout)
    j2 = _bench_line(_launch(2, None, ["bench.py", "--gpus", "2"] + args, env).stdout)
    assert j1["cpu_baseline"]["parity_on_sample"] is True and "parity_at_n" not in j1
    assert j2["parity_at_n"]["parity_on_sample"] is True and j2["parity_at_n"]["sample_hits"] > 1000 and "cpu_baseline" not in j2
    assert j2["sanity_batch"] == j1["sanity_batch"]


def test_bench_eight_ranks_dry_run():
    """`python bench.py --gpus 8` end to end before the driver's scaling run does it for real: eight ranks (all on GPU 0 over gloo
    when the box has fewer than eight GPUs, one per GPU over nccl otherwise), four blocks of the GTDB-scale index each, ragged gather
    sizes, K3 on rank 0 over eight hit lists, the merged list checked against the CPU oracle over rows fetched from every rank
    (`parity_at_n`), and the same order-independent checksum as the one-rank run of the same batches."""
    import torch
    free_b, _ = torch.cuda.mem_get_info(0)
    n_dev = torch.cuda.device_count()
    if n_dev < 8 and free_b < 80e9:
        pytest.skip("needs 80 GB of free HBM for the 58 GB index + eight ranks' workspaces on one GPU")
    env, _ = _env("KMCP_BENCH_SAME_GPU")
    multi = n_dev >= 8
    if not multi:
        env["KMCP_BENCH_SAME_GPU"] = "1"
    args = ["--batch-reads", "65536", "--steps", "2", "--warmup", "1"]
    eight = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + args, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    if eight.returncode != 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "eight_ranks_failure.txt"), "w") as fh:
            fh.write(eight.stdout + "\n" + eight.stderr)
    assert eight.returncode == 0, (eight.stdout[-2000:], eight.stderr[-4000:])
    j8 = _bench_line(eight.stdout)
    assert j8["n_gpus"] == 8 and j8["scaling"] == "strong" and j8["config"]["parallelism"] == "block-shard x8"
    assert j8["ranks"]["world_size"] == 8 and j8["ranks"]["ranks_reporting"] == list(range(8))
    assert j8["ranks"]["backend"] == ("nccl" if multi else "gloo")
    assert j8["parity_at_n"]["parity_on_sample"] is True and j8["parity_at_n"]["sample_hits"] > 100 and "parity_ok" not in j8
    assert abs(j8["config"]["index_bytes_this_rank"] * 8 - j8["config"]["index_bytes"]) < 0.02 * j8["config"]["index_bytes"]
    assert j8["planted_recall"] > 0.99 and j8["value"] > 0
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--no-cpu-baseline", "--no-secondary", "--no-extras"] + args, capture_output=True, text=True,
                         timeout=900, env={k: v for k, v in env.items() if k != "KMCP_BENCH_SAME_GPU"}, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    j1 = _bench_line(one.stdout)
    assert j8["sanity_batch"] == j1["sanity_batch"] and j1["sanity_batch"]["hits"] > 10000
    assert j8["hits_per_step"] == j1["hits_per_step"] and j8["matches_per_step"] == j1["matches_per_step"]


def test_bench_reports_a_parity_failure_on_the_line_and_in_the_exit_status():
    """If the merged multi-GPU hit list ever differs from the oracle, the run must not die in a traceback with the other ranks at a
    barrier: the line comes out with `parity_ok: false` + what differed, and the exit status is 3 (KMCP_BENCH_FAULT=parity drops
    one GPU hit of the sample before the comparison)."""
    env, _ = _env("KMCP_BENCH_SAME_GPU")
    env["KMCP_BENCH_FAULT"] = "parity"
    args = ["--steps", "2", "--warmup", "1", "--workload", "config1", "--batch-reads", "32768", "--no-secondary", "--no-extras"]
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode != 0, two.stdout[-2000:]
    j2 = _bench_line(two.stdout)
    assert j2["parity_ok"] is False and j2["parity_at_n"]["parity_on_sample"] is False
    assert j2["parity_failure"] == {"gpu_only": 0, "oracle_only": 1, "qkmers_differ": 0}
    assert j2["value"] > 0 and j2["ranks"]["ranks_reporting"] == [0, 1]


def test_sharded_searcher_overflow_loop(oracle_lib, tmp_path):
    """Hundreds of chance hits per read: the per-rank hit buffers (8 per read) overflow on every rank; the ranks agree on the
    largest count with an all-reduce and rerun (kmcp_amd/dist.py)."""
    O = oracle_lib
    genomes = synth.random_genomes(400, 3000, seed=195)
    db_dir = synth.make_db(tmp_path / "db", genomes, k=21, fpr=0.3, block_size=104)  # 4 blocks: two per rank
    reads = synth.sample_reads(genomes, 120, 150, seed=196, frac_random=0.5)
    ids = [f"q{i}" for i in range(len(reads))]
    fq = str(tmp_path / "reads.fq")
    write_fastq(fq, ids, reads)
    odb = O.OracleDB(db_dir)
    p = O.default_params(min_qcov=0.31, max_fpr=1.0, min_matched=1)
    want, trailer = oracle_tsv(O, odb, ids, reads, params=p)
    odb.close()
    assert len(want) > 50 * len(reads)
    env, _ = _env("KMCP_DIST_SAME_GPU")
    out = str(tmp_path / "o.tsv")
    r = _launch(2, None, ["-m", "kmcp_amd.dist_search", "-d", os.path.dirname(db_dir), fq, "-o", out, "-t", "0.31", "-f", "1", "-c", "1"], env)
    compare(open(out).read().split("\n"), want, trailer)
    # both hosts log the same order-independent checksum of the (queryIdx, column, mKmers) tuples: two ranks here, one GPU there
    import re
    from tests.test_gpu_cli import CLI
    two = re.search(r"2 rank\(s\), backend (\w+): matches: (\d+), checksum ([0-9a-f]{16})", r.stderr)
    assert two and int(two.group(2)) == len(want), r.stderr[-2000:]
    one = subprocess.run([CLI, "-d", os.path.dirname(db_dir), fq, "-o", str(tmp_path / "cli.tsv"), "-t", "0.31", "-f", "1", "-c", "1"], capture_output=True, text=True,
                         timeout=300)
    assert one.returncode == 0, one.stderr
    m1 = re.search(r"matches: (\d+), checksum ([0-9a-f]{16})", one.stderr)
    assert m1 and (m1.group(1), m1.group(2)) == (two.group(2), two.group(3)), (m1.groups(), two.groups())


def test_sharded_searcher_walks_the_k_sizes_of_a_multi_k_database(oracle_lib, tmp_path):
    """One process per GPU against a database with `ks: [21, 31]`: queries that match nothing with k=31 go again with k=21 on
    every rank (rank 0 broadcasts which), as kmcpg_search_batch does inside one process and the reference in handleQuery
    (util-db-search.go:764, :1016-1022).  Two ranks through python -m kmcp_amd.dist_search against the oracle's TSV (kSize
    column included), and the in-process ShardedSearcher against Database.search."""
    import re
    import numpy as np
    from kmcp_amd import Database, default_params, lib
    from kmcp_amd.dist import ShardedSearcher
    O = oracle_lib
    genomes = synth.random_genomes(12, 8000, seed=277)
    cols = []
    for gi, g in enumerate(genomes):
        h = np.concatenate([O.generate_kmers(g, O.sketch_cfg(k=k)) for k in (21, 31)])
        cols.append((f"g{gi}", len(g), 0, 1, O.sort_unique(h)))
    db_dir = O.build_db(str(tmp_path / "db"), O.sketch_cfg(k=31), cols, num_hashes=1, fpr=0.1, threads=4, block_size=8)  # 2 blocks: one per rank; -t 0.2 must exceed the FPR
    yml = open(db_dir + "/__db.yml").read()
    yml2 = re.sub(r"ks:\n- 31\n", "ks:\n- 21\n- 31\n", yml)
    assert yml2 != yml
    open(db_dir + "/__db.yml", "w").write(yml2)
    rng = np.random.default_rng(3)
    reads = []
    for i in range(240):
        g = genomes[i % len(genomes)]
        pos = int(rng.integers(0, len(g) - 150))
        r = bytearray(g[pos:pos + 150])
        if i % 3 == 2:  # a substitution every 28 bases: no intact 31-mer, runs of 27 leave 7 intact 21-mers each
            for j in range(5, 150, 28):
                r[j] = ord("A") if r[j] != ord("A") else ord("C")
        elif i % 7 == 0:
            r = bytearray(rng.choice(list(b"ACGT"), 150).astype(np.uint8))  # matches with neither size
        reads.append(bytes(r))
    ids = [f"q{i}" for i in range(len(reads))]
    fq = str(tmp_path / "reads.fq")
    write_fastq(fq, ids, reads)
    odb = O.OracleDB(db_dir)
    p = O.default_params(min_qcov=0.2)
    want, trailer = oracle_tsv(O, odb, ids, reads, params=p)
    ks = {odb.search(r, None, params=p)["k"] for r in reads}
    odb.close()
    assert ks == {21, 31}
    env, _ = _env("KMCP_DIST_SAME_GPU")
    out = str(tmp_path / "o.tsv")
    _launch(2, None, ["-m", "kmcp_amd.dist_search", "-d", os.path.dirname(db_dir), fq, "-o", out, "-t", "0.2", "--gpu-batch", "100"], env)
    compare(open(out).read().split("\n"), want, trailer)
    # in process, one rank: the same walk through the per-shard pair == the library's own
    seqs, offs = lib.pack_reads(reads)
    srch = ShardedSearcher(db_dir, device=0)
    try:
        assert srch.db.ks == [31, 21]
        a = srch.search(seqs, offs, params=default_params(min_qcov=0.2))
        # pairs with --try-se: mates of unmatched pairs searched on their own before the next k (:831-850, :1001-1014)
        reads2 = [genomes[(i + 5) % len(genomes)][40:190] if i % 4 else bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)) for i in range(len(reads))]
        seqs2, offs2 = lib.pack_reads(reads2)
        pe = default_params(min_qcov=0.4, try_se=1, fpr_buf_size=499)
        a2 = srch.search(seqs, offs, params=pe, seqs2=seqs2, offs2=offs2)
    finally:
        srch.close()
    with Database.open(db_dir) as db:
        b = db.search(reads, params=default_params(min_qcov=0.2))
        b2 = db.search(reads, reads2, params=pe)
    for x, y in ((a, b), (a2, b2)):
        for f in ("qlen", "qkmers", "ksize", "offs", "matches"):
            assert np.array_equal(getattr(x, f), getattr(y, f)), f
    assert set(a.ksize.tolist()) == {21, 31} and len(a2.matches) > 50 and len(set(a2.qlen.tolist())) > 1  # some pairs answered by one mate (qLen 150)


def test_gather_hits_over_rccl_with_one_rank(tmp_path):
    """The nccl (= RCCL) flavour of the exchange — all_gather_into_tensor of the counts, gather of the padded hit buffers, on
    device tensors — cannot meet a second GPU on this box, but it can run as a one-rank group: the calls, dtypes and shapes are
    the ones the N-GPU run makes."""
    script = tmp_path / "one_rank_rccl.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from kmcp_amd.dist import gather_hits
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29747", world_size=1, rank=0, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
hits = torch.arange(3000, dtype=torch.int32, device=dev).reshape(1000, 3).contiguous()
for n in (0, 1, 417, 1000):
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    parts = gather_hits(hits, cnt, dst=0, force_collectives=True)
    assert len(parts) == 1 and parts[0].shape == (n, 3) and torch.equal(parts[0], hits[:n]), n
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("rccl one-rank ok")
''' % ROOT)
    env, _ = _env("KMCP_UNUSED")
    env.pop("KMCP_UNUSED", None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
