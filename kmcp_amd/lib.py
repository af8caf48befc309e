"""ctypes binding of libkmcpgpu.so (include/kmcp_gpu.h).

Plumbing only: the product is the shared library.  There is no CPU fallback — if the library is missing
or no GPU is present, every compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkmcpgpu.so")

EXPORTS = [
    "kmcpg_open", "kmcpg_close", "kmcpg_last_error", "kmcpg_db_info", "kmcpg_col_info", "kmcpg_search_batch",
    "kmcpg_result_free", "kmcpg_query_device", "kmcpg_finalize", "kmcpg_open_synthetic", "kmcpg_plant",
    "kmcpg_read_rows", "kmcpg_block_info", "kmcpg_kmers_device", "kmcpg_plant_reads_device", "kmcpg_set_profiling",
    "kmcpg_last_timing", "kmcpg_open_devices", "kmcpg_build_db", "kmcpg_submit", "kmcpg_wait", "kmcpg_read_row_range", "kmcpg_timing_at", "kmcpg_last_gathered_bytes", "kmcpg_last_hash_bytes", "kmcpg_last_tail_waves",
    "kmcpg_db_ks", "kmcpg_open_paged", "kmcpg_paged_info", "kmcpg_exchange_info", "kmcpg_batch_hint", "kmcpg_group_device", "kmcpg_finalize_grouped",
    "kmcpg_search_batch_pairs", "kmcpg_wait_pairs", "kmcpg_result_pairs_free", "kmcpg_expand_pairs", "kmcpg_save_db",
    "kmcpg_pack2", "kmcpg_unpack2", "kmcpg_submit_packed", "kmcpg_host_alloc", "kmcpg_host_free",
    "kmcpg_kmers_device_packed", "kmcpg_k1_codes_batches",
]


def pack2(pieces, codes=None, exc=None, pos=0):
    """kmcpg_pack2 over a list of byte strings / uint8 arrays appended one after the other from base position `pos`:
    returns (codes uint8[(total + 3) // 4 + 8], exc EXC_DTYPE[n_exc], total bases)."""
    arrs = [np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else np.ascontiguousarray(x, dtype=np.uint8) for x in pieces]
    total = pos + sum(len(a) for a in arrs)
    if codes is None:
        codes = np.zeros((total + 3) // 4 + 8, dtype=np.uint8)
    cap = 1024 if exc is None else max(1024, 2 * len(exc))
    runs = np.zeros(cap, dtype=EXC_DTYPE)
    n_exc = C.c_uint64(0)
    if exc is not None and len(exc):
        runs[:len(exc)] = exc
        n_exc.value = len(exc)
    at = pos
    for a in arrs:
        while True:
            before = n_exc.value
            rc = load().kmcpg_pack2(a.ctypes.data, len(a), at, codes.ctypes.data, runs.ctypes.data, len(runs), C.byref(n_exc))
            if rc == 0:
                break
            if rc != -5:
                _check(rc)
            grown = np.zeros(max(2 * len(runs), int(n_exc.value) + 1024), dtype=EXC_DTYPE)
            grown[:before] = runs[:before]
            runs = grown
            n_exc.value = before
        at += len(a)
    return codes, runs[:n_exc.value].copy(), total


class PinnedBytes:
    """uint8 array in page-locked memory from kmcpg_host_alloc (`.a`); freed by close() / the context manager."""

    def __init__(self, n):
        self._p = C.c_void_p()
        _check(load().kmcpg_host_alloc(n, C.byref(self._p)))
        self.a = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(max(int(n), 64),))

    def close(self):
        if self._p:
            self.a = None
            _check(load().kmcpg_host_free(self._p))
            self._p = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def unpack2(codes, n_bases, exc):
    out = np.empty(n_bases, dtype=np.uint8)
    _check(load().kmcpg_unpack2(codes.ctypes.data, n_bases, exc.ctypes.data if len(exc) else None, len(exc), out.ctypes.data))
    return out


class KmcpGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libkmcpgpu error {code}: {msg}")
        self.code = code


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("reserved", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("k", C.c_int32), ("canonical", C.c_int32), ("num_hashes", C.c_int32), ("scaled", C.c_int32),
                ("scale", C.c_uint32), ("minimizer", C.c_int32), ("minimizer_w", C.c_uint32), ("syncmer", C.c_int32),
                ("syncmer_s", C.c_uint32), ("fpr", C.c_double), ("n_blocks", C.c_int32), ("n_blocks_local", C.c_int32),
                ("n_cols", C.c_uint64), ("matrix_bytes", C.c_uint64), ("matrix_bytes_local", C.c_uint64),
                ("row_bytes_sum_local", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("min_qlen", C.c_int32), ("min_matched", C.c_int32), ("min_qcov", C.c_double), ("min_tcov", C.c_double),
                ("max_fpr", C.c_double), ("dedup_threshold", C.c_int32), ("try_se", C.c_int32), ("sort_by", C.c_int32),
                ("do_not_sort", C.c_int32), ("top_n_scores", C.c_int32), ("fpr_buf_size", C.c_int32), ("k", C.c_int32),
                ("reserved", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("read", C.c_uint32), ("col", C.c_uint32), ("count", C.c_uint32)]


class Match(C.Structure):
    _fields_ = [("col", C.c_uint32), ("target_idx", C.c_uint32), ("gsize", C.c_uint64), ("mkmers", C.c_int32),
                ("reserved", C.c_int32), ("fpr", C.c_double), ("qcov", C.c_double), ("tcov", C.c_double), ("jacc", C.c_double)]


class Result(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("k", C.c_int32), ("qlen", C.POINTER(C.c_int32)), ("qkmers", C.POINTER(C.c_int32)),
                ("ksize", C.POINTER(C.c_int32)), ("match_offs", C.POINTER(C.c_uint64)), ("matches", C.POINTER(Match)), ("owner", C.c_void_p)]


class SynthSpec(C.Structure):
    _fields_ = [("k", C.c_int32), ("num_hashes", C.c_int32), ("fpr", C.c_double), ("n_blocks", C.c_uint32),
                ("cols_per_block", C.c_uint32), ("num_sigs", C.c_uint64), ("kmers_per_col", C.c_uint64), ("seed", C.c_uint64),
                ("scale", C.c_uint32), ("syncmer_s", C.c_uint32), ("minimizer_w", C.c_uint32), ("sigs_step", C.c_uint32)]


class BuildCfg(C.Structure):
    _fields_ = [("k", C.c_int32), ("canonical", C.c_int32), ("num_hashes", C.c_int32), ("fpr", C.c_double), ("threads", C.c_int32),
                ("block_size", C.c_int32), ("scale", C.c_uint32), ("minimizer_w", C.c_uint32), ("syncmer_s", C.c_uint32),
                ("split_seq", C.c_int32), ("split_size", C.c_int32), ("split_num", C.c_int32), ("split_overlap", C.c_int32),
                ("alias", C.c_char_p), ("kmers_x", C.c_uint64), ("block_size_x", C.c_int32), ("uniform_sigs", C.c_int32), ("kmers_8", C.c_uint64),
                ("kmers_1", C.c_uint64)]


class BuildCol(C.Structure):
    _fields_ = [("name", C.c_char_p), ("gsize", C.c_uint64), ("chunk_idx", C.c_uint32), ("chunks", C.c_uint32),
                ("hashes", C.c_void_p), ("n_hashes", C.c_uint64)]


HIT_DTYPE = np.dtype([("read", np.uint32), ("col", np.uint32), ("count", np.uint32)])
PAIR_DTYPE = np.dtype([("col", np.uint32), ("count", np.uint32)])
EXC_DTYPE = np.dtype([("pos", np.uint64), ("len", np.uint32), ("byte", np.uint32)])  # kmcpg_exc_run
MATCH_DTYPE = np.dtype([("col", np.uint32), ("target_idx", np.uint32), ("gsize", np.uint64), ("mkmers", np.int32),
                        ("reserved", np.int32), ("fpr", np.float64), ("qcov", np.float64), ("tcov", np.float64),
                        ("jacc", np.float64)])
assert HIT_DTYPE.itemsize == C.sizeof(Hit) and MATCH_DTYPE.itemsize == C.sizeof(Match)


def default_params(**kw):
    """Defaults of `kmcp search` (kmcp/cmd/search.go:1052-1102)."""
    p = Params(min_qlen=30, min_matched=10, min_qcov=0.55, min_tcov=0.0, max_fpr=0.01, dedup_threshold=256, try_se=0,
               sort_by=0, do_not_sort=0, top_n_scores=0, fpr_buf_size=0, k=0, reserved=0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


_lib = None


def load():
    """Load libkmcpgpu.so; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). kmcp_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP/HSA runtime under the soname libamdhip64.so.7.  Two HIP runtimes in
    # one process cannot both own the GPU, so torch's copy must be mapped first; libkmcpgpu.so's NEEDED entry
    # then binds to it.  (Stand-alone C/C++/Go hosts simply get /opt/rocm's runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64p, i32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
    L.kmcpg_last_error.restype = C.c_char_p
    L.kmcpg_open.argtypes = [C.c_char_p, C.POINTER(Opts), C.POINTER(vp)]
    L.kmcpg_open_synthetic.argtypes = [C.POINTER(SynthSpec), C.POINTER(Opts), C.POINTER(vp)]
    L.kmcpg_open_devices.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
    L.kmcpg_open_paged.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.kmcpg_paged_info.argtypes = [vp, i32p, u64p]
    L.kmcpg_exchange_info.argtypes = [vp]
    L.kmcpg_batch_hint.argtypes = [vp, u64p]
    L.kmcpg_exchange_info.restype = C.c_char_p
    L.kmcpg_close.argtypes = [vp]
    L.kmcpg_db_info.argtypes = [vp, C.POINTER(Info)]
    L.kmcpg_db_ks.argtypes = [vp, i32p, C.c_int32, i32p]
    L.kmcpg_col_info.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), u64p, u64p]
    L.kmcpg_block_info.argtypes = [vp, C.c_uint32, u64p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), i32p, C.POINTER(C.c_uint32)]
    L.kmcpg_search_batch.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(Params), C.POINTER(Result)]
    L.kmcpg_result_free.argtypes = [C.POINTER(Result)]
    L.kmcpg_result_free.restype = None
    L.kmcpg_submit.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(Params), C.POINTER(vp)]
    L.kmcpg_wait.argtypes = [vp, C.POINTER(Result)]
    L.kmcpg_query_device.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(Params), vp,
                                     C.c_uint64, vp, vp, vp, vp]
    L.kmcpg_finalize.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint32, C.POINTER(Params), C.POINTER(Result)]
    L.kmcpg_group_device.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_uint32, C.POINTER(Params), vp, vp, vp]
    L.kmcpg_finalize_grouped.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(Params), C.POINTER(Result)]
    L.kmcpg_search_batch_pairs.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(Params), C.POINTER(ResultPairs)]
    L.kmcpg_wait_pairs.argtypes = [vp, C.POINTER(ResultPairs)]
    L.kmcpg_result_pairs_free.argtypes = [C.POINTER(ResultPairs)]
    L.kmcpg_result_pairs_free.restype = None
    L.kmcpg_expand_pairs.argtypes = [vp, C.c_int32, vp, C.c_uint64, vp]
    L.kmcpg_plant.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
    L.kmcpg_read_rows.argtypes = [vp, C.c_uint32, vp, C.c_uint64, vp]
    L.kmcpg_read_row_range.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint64, vp]
    L.kmcpg_kmers_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(Params), vp, C.c_uint64,
                                     vp, vp, vp]
    L.kmcpg_kmers_device_packed.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(Params), vp, C.c_uint64,
                                            vp, vp, vp]
    L.kmcpg_k1_codes_batches.argtypes = [vp, u64p, u64p]
    L.kmcpg_plant_reads_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp]
    L.kmcpg_set_profiling.argtypes = [vp, C.c_int]
    L.kmcpg_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.kmcpg_last_gathered_bytes.argtypes = [vp, u64p]
    L.kmcpg_last_tail_waves.argtypes = [vp, u64p]
    L.kmcpg_last_hash_bytes.argtypes = [vp, u64p]
    L.kmcpg_timing_at.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.kmcpg_build_db.argtypes = [C.c_char_p, C.POINTER(BuildCfg), C.POINTER(BuildCol), C.c_uint32, C.c_int32]
    L.kmcpg_save_db.argtypes = [vp, C.c_char_p]
    L.kmcpg_pack2.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint64, u64p]
    L.kmcpg_unpack2.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp]
    L.kmcpg_host_alloc.argtypes = [C.c_uint64, C.POINTER(vp)]
    L.kmcpg_host_free.argtypes = [vp]
    L.kmcpg_submit_packed.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint32, C.POINTER(Params), C.POINTER(vp)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise KmcpGpuError(rc, load().kmcpg_last_error().decode(errors="replace"))


class Pair(C.Structure):
    _fields_ = [("col", C.c_uint32), ("count", C.c_uint32)]


class ResultPairs(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("k", C.c_int32), ("qlen", C.POINTER(C.c_int32)), ("qkmers", C.POINTER(C.c_int32)),
                ("ksize", C.POINTER(C.c_int32)), ("match_offs", C.POINTER(C.c_uint64)), ("pairs", C.POINTER(Pair)), ("owner", C.c_void_p)]


def pack_reads(reads):
    """list of bytes -> (uint8 array, uint64 offsets[n+1])."""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, dtype=np.uint8)
    return seqs, offs


def build_db(out_dir, columns, k=21, num_hashes=1, fpr=0.3, threads=32, block_size=0, scale=1, minimizer_w=0, syncmer_s=0, device=0,
             alias="kmcp-gpu-db", kmers_x=0, block_size_x=0, kmers_8=0, kmers_1=0, uniform_sigs=0):
    """`kmcp index` on the GPU.  columns: list of (name, gsize, chunk_idx, chunks, sorted-unique uint64 hashes).
    uniform_sigs: 0 = the reference's per-block NumSigs; 1 / 2 = blocks share NumSigs (groupable in HBM), see kmcp_gpu.h."""
    cfg = BuildCfg(k=k, canonical=1, num_hashes=num_hashes, fpr=fpr, threads=threads, block_size=block_size, scale=scale,
                   minimizer_w=minimizer_w, syncmer_s=syncmer_s, alias=alias.encode(), kmers_x=kmers_x, block_size_x=block_size_x, kmers_8=kmers_8,
                   kmers_1=kmers_1, uniform_sigs=uniform_sigs)
    arr = (BuildCol * len(columns))()
    keep = []
    for i, (name, gsize, ci, nch, h) in enumerate(columns):
        h = np.ascontiguousarray(h, dtype=np.uint64)
        keep.append(h)
        arr[i] = BuildCol(name.encode(), gsize, ci, nch, h.ctypes.data, len(h))
    _check(load().kmcpg_build_db(os.fsencode(out_dir), C.byref(cfg), arr, len(columns), device))
    return os.path.join(out_dir, "R001")


class BatchResult:
    """QueryResult for a batch (numpy copies of kmcpg_result)."""

    def __init__(self, qlen, qkmers, offs, matches, k, ksize=None):
        self.qlen, self.qkmers, self.offs, self.matches, self.k, self.ksize = qlen, qkmers, offs, matches, k, ksize

    def __len__(self):
        return len(self.qlen)

    def read(self, i):
        return self.matches[int(self.offs[i]):int(self.offs[i + 1])]


def _copy_result(r):
    n = r.n_reads
    qlen = np.ctypeslib.as_array(r.qlen, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    qk = np.ctypeslib.as_array(r.qkmers, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    offs = np.ctypeslib.as_array(r.match_offs, shape=(n + 1,)).copy()
    m = int(offs[-1])
    if m:
        buf = C.string_at(r.matches, m * C.sizeof(Match))
        matches = np.frombuffer(buf, dtype=MATCH_DTYPE).copy()
    else:
        matches = np.zeros(0, dtype=MATCH_DTYPE)
    ks = np.ctypeslib.as_array(r.ksize, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    out = BatchResult(qlen, qk, offs, matches, r.k, ks)
    load().kmcpg_result_free(C.byref(r))
    return out


class PairsResult:
    """kmcpg_result_pairs for a batch (numpy copies): the final matches of every query as (column, mKmers) pairs."""

    def __init__(self, qlen, qkmers, offs, pairs, k, ksize):
        self.qlen, self.qkmers, self.offs, self.pairs, self.k, self.ksize = qlen, qkmers, offs, pairs, k, ksize

    def __len__(self):
        return len(self.qlen)

    def read(self, i):
        return self.pairs[int(self.offs[i]):int(self.offs[i + 1])]


def _copy_pairs(r):
    n = r.n_reads
    qlen = np.ctypeslib.as_array(r.qlen, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    qk = np.ctypeslib.as_array(r.qkmers, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    offs = np.ctypeslib.as_array(r.match_offs, shape=(n + 1,)).copy() if n else np.zeros(1, np.uint64)
    ks = np.ctypeslib.as_array(r.ksize, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    m = int(offs[-1])
    pairs = np.frombuffer(C.string_at(r.pairs, m * 8), dtype=np.uint32).reshape(m, 2).copy() if m else np.zeros((0, 2), np.uint32)
    out = PairsResult(qlen, qk, offs, pairs, r.k, ks)
    load().kmcpg_result_pairs_free(C.byref(r))
    return out


class Database:
    """A kmcp database resident in the HBM of one GPU (or one shard of it)."""

    def __init__(self, handle):
        self._h = handle
        info = Info()
        _check(load().kmcpg_db_info(self._h, C.byref(info)))
        self.info = info
        self._names = {}
        n = C.c_int32(0)
        _check(load().kmcpg_db_ks(self._h, None, 0, C.byref(n)))
        buf = (C.c_int32 * max(1, n.value))()
        _check(load().kmcpg_db_ks(self._h, buf, n.value, C.byref(n)))
        self.ks = [int(buf[i]) for i in range(n.value)]  # the database's k-mer sizes, largest first

    @classmethod
    def open(cls, db_dir, device=0, shard_rank=0, shard_count=1):
        h = C.c_void_p()
        o = Opts(device, shard_rank, shard_count, 0)
        _check(load().kmcpg_open(os.fsencode(db_dir), C.byref(o), C.byref(h)))
        return cls(h)

    @classmethod
    def open_devices(cls, db_dir, devices):
        """One process, several GPUs: blocks partitioned over `devices`, search() fans out and merges on the host."""
        h = C.c_void_p()
        arr = (C.c_int32 * len(devices))(*devices)
        _check(load().kmcpg_open_devices(os.fsencode(db_dir), arr, len(devices), C.byref(h)))
        return cls(h)

    @classmethod
    def open_paged(cls, db_dir, device=0, passes=0):
        """A database larger than the free HBM of one GPU: searched in `passes` shards per batch (0 = as few as fit)."""
        h = C.c_void_p()
        _check(load().kmcpg_open_paged(os.fsencode(db_dir), device, passes, C.byref(h)))
        return cls(h)

    def save(self, out_dir):
        """kmcpg_save_db: the resident database as <out_dir>/R001 in the reference's on-disk format; returns that directory."""
        _check(load().kmcpg_save_db(self._h, os.fsencode(out_dir)))
        return os.path.join(out_dir, "R001")

    def exchange_info(self):
        return load().kmcpg_exchange_info(self._h).decode()

    def paged_info(self):
        """(passes, shard uploads so far); passes == 0 for resident handles"""
        p, u = C.c_int32(0), C.c_uint64(0)
        _check(load().kmcpg_paged_info(self._h, C.byref(p), C.byref(u)))
        return int(p.value), int(u.value)

    def batch_hint(self):
        """bases one batch may hold so that its device workspace fits beside the resident index (0 = unknown)"""
        n = C.c_uint64(0)
        _check(load().kmcpg_batch_hint(self._h, C.byref(n)))
        return int(n.value)

    @classmethod
    def open_synthetic(cls, spec: SynthSpec, device=0, shard_rank=0, shard_count=1):
        h = C.c_void_p()
        o = Opts(device, shard_rank, shard_count, 0)
        _check(load().kmcpg_open_synthetic(C.byref(spec), C.byref(o), C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            load().kmcpg_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def col_info(self, col):
        if col not in self._names:
            name, ti, gs, sz = C.c_char_p(), C.c_uint32(), C.c_uint64(), C.c_uint64()
            _check(load().kmcpg_col_info(self._h, col, C.byref(name), C.byref(ti), C.byref(gs), C.byref(sz)))
            self._names[col] = (name.value.decode(), ti.value, gs.value, sz.value)
        return self._names[col]

    def block_info(self, b):
        ns, nc, rb, st, loc, cb = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_uint32()
        _check(load().kmcpg_block_info(self._h, b, C.byref(ns), C.byref(nc), C.byref(rb), C.byref(st), C.byref(loc), C.byref(cb)))
        return dict(num_sigs=ns.value, n_cols=nc.value, row_bytes=rb.value, stride=st.value, local=bool(loc.value), col_base=cb.value)

    # ---- whole pipeline, host buffers -------------------------------------------------------------
    def search(self, reads, reads2=None, params=None):
        seqs, offs = pack_reads(reads)
        s2 = o2 = None
        if reads2 is not None:
            s2, o2 = pack_reads(reads2)
        return self.search_packed(seqs, offs, s2, o2, params)

    def search_packed(self, seqs, offs, seqs2=None, offs2=None, params=None):
        p = params or default_params()
        r = Result()
        n = len(offs) - 1
        _check(load().kmcpg_search_batch(self._h, seqs.ctypes.data, offs.ctypes.data,
                                         seqs2.ctypes.data if seqs2 is not None else None,
                                         offs2.ctypes.data if offs2 is not None else None, n, C.byref(p), C.byref(r)))
        return _copy_result(r)

    def submit(self, seqs, offs, seqs2=None, offs2=None, params=None):
        """kmcpg_submit: returns a ticket; the arrays may be reused at once."""
        p = params or default_params()
        t = C.c_void_p()
        n = len(offs) - 1
        _check(load().kmcpg_submit(self._h, seqs.ctypes.data, offs.ctypes.data, seqs2.ctypes.data if seqs2 is not None else None,
                                   offs2.ctypes.data if offs2 is not None else None, n, C.byref(p), C.byref(t)))
        return t

    def submit_packed(self, codes, offs, exc, params=None):
        """kmcpg_submit_packed: a batch as 2-bit codes (pack2) + exception runs; offs in bases."""
        p = params or default_params()
        t = C.c_void_p()
        n = len(offs) - 1
        _check(load().kmcpg_submit_packed(self._h, codes.ctypes.data, offs.ctypes.data, exc.ctypes.data if len(exc) else None, len(exc), n, C.byref(p), C.byref(t)))
        return t

    def wait(self, ticket, count_only=False):
        """kmcpg_wait: the finalized matches of a submitted batch (or just their number)."""
        r = Result()
        _check(load().kmcpg_wait(ticket, C.byref(r)))
        if count_only:
            m = int(r.match_offs[r.n_reads])
            load().kmcpg_result_free(C.byref(r))
            return m
        return _copy_result(r)

    # ---- compact results: (column, mKmers) pairs instead of Match records ------------------------
    def search_pairs(self, reads, reads2=None, params=None):
        seqs, offs = pack_reads(reads)
        s2 = o2 = None
        if reads2 is not None:
            s2, o2 = pack_reads(reads2)
        return self.search_packed_pairs(seqs, offs, s2, o2, params)

    def search_packed_pairs(self, seqs, offs, seqs2=None, offs2=None, params=None, count_only=False):
        p = params or default_params()
        r = ResultPairs()
        n = len(offs) - 1
        _check(load().kmcpg_search_batch_pairs(self._h, seqs.ctypes.data, offs.ctypes.data, seqs2.ctypes.data if seqs2 is not None else None,
                                               offs2.ctypes.data if offs2 is not None else None, n, C.byref(p), C.byref(r)))
        if count_only:
            m = int(r.match_offs[n]) if n else 0
            load().kmcpg_result_pairs_free(C.byref(r))
            return m
        return _copy_pairs(r)

    def wait_pairs(self, ticket, count_only=False):
        r = ResultPairs()
        _check(load().kmcpg_wait_pairs(ticket, C.byref(r)))
        if count_only:
            m = int(r.match_offs[r.n_reads]) if r.n_reads else 0
            load().kmcpg_result_pairs_free(C.byref(r))
            return m
        return _copy_pairs(r)

    def expand_pairs(self, qkmers, pairs):
        """kmcpg_expand_pairs: the Match records (MATCH_DTYPE) of one query's pairs"""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros(len(pairs), dtype=MATCH_DTYPE)
        _check(load().kmcpg_expand_pairs(self._h, int(qkmers), pairs.ctypes.data, len(pairs), out.ctypes.data))
        return out

    def search_packed_count(self, seqs, offs, params=None):
        """kmcpg_search_batch without copying the result into numpy: returns the number of matches (timing of the C boundary)."""
        p = params or default_params()
        r = Result()
        n = len(offs) - 1
        _check(load().kmcpg_search_batch(self._h, seqs.ctypes.data, offs.ctypes.data, None, None, n, C.byref(p), C.byref(r)))
        m = int(r.match_offs[n])
        load().kmcpg_result_free(C.byref(r))
        return m

    # ---- GPU half on device pointers (torch tensors' data_ptr()) ----------------------------------
    def query_device(self, d_seqs, d_offs, n_reads, total_bases, max_read_len, d_hits, hit_cap, d_counters, d_qkmers,
                     d_qlen, params=None, d_seqs2=None, d_offs2=None, stream=None):
        p = params or default_params()
        _check(load().kmcpg_query_device(self._h, d_seqs, d_offs, d_seqs2, d_offs2, n_reads, total_bases, max_read_len,
                                         C.byref(p), d_hits, hit_cap, d_counters, d_qkmers, d_qlen, stream))

    def kmers_device(self, d_seqs, d_offs, n_reads, total_bases, max_read_len, d_hashes, hashes_cap, d_koff, d_nk,
                     params=None, stream=None):
        p = params or default_params()
        _check(load().kmcpg_kmers_device(self._h, d_seqs, d_offs, n_reads, total_bases, max_read_len, C.byref(p),
                                         d_hashes, hashes_cap, d_koff, d_nk, stream))

    def kmers_device_packed(self, d_codes, d_exc, n_exc, d_text, d_offs, n_reads, total_bases, max_read_len, d_hashes, hashes_cap, d_koff, d_nk,
                            params=None, stream=None):
        """kmers_device on a batch that is on the device as 2-bit codes + runs of foreign bytes (kmcpg_kmers_device_packed)"""
        p = params or default_params()
        _check(load().kmcpg_kmers_device_packed(self._h, d_codes, d_exc, n_exc, d_text, d_offs, n_reads, total_bases, max_read_len, C.byref(p),
                                                d_hashes, hashes_cap, d_koff, d_nk, stream))

    def k1_codes_batches(self):
        """(packed batches whose k-mer kernels read the codes directly, packed batches expanded to text first)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(load().kmcpg_k1_codes_batches(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def finalize(self, hits, qkmers, qlen, params=None):
        """hits: structured array HIT_DTYPE (any order, may be the concatenation of all shards)."""
        p = params or default_params()
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        qkmers = np.ascontiguousarray(qkmers, dtype=np.int32)
        qlen = np.ascontiguousarray(qlen, dtype=np.int32)
        r = Result()
        _check(load().kmcpg_finalize(self._h, hits.ctypes.data, len(hits), qkmers.ctypes.data, qlen.ctypes.data,
                                     len(qkmers), C.byref(p), C.byref(r)))
        return _copy_result(r)

    def finalize_count(self, hits_u32, qkmers, qlen, params=None):
        """kmcpg_finalize on raw buffers (hits: C-contiguous int32/uint32 [n, 3]); returns the number of matches, copies nothing."""
        p = params or default_params()
        assert hits_u32.flags["C_CONTIGUOUS"] and hits_u32.dtype.itemsize == 4 and qkmers.dtype == np.int32 and qlen.dtype == np.int32
        r = Result()
        n = len(qkmers)
        _check(load().kmcpg_finalize(self._h, hits_u32.ctypes.data, hits_u32.shape[0], qkmers.ctypes.data, qlen.ctypes.data, n, C.byref(p), C.byref(r)))
        m = int(r.match_offs[n]) if n else 0
        load().kmcpg_result_free(C.byref(r))
        return m

    # ---- the host half split in two: K3 on the device (group, -T, order), expansion to Match records on the host ----------
    def group_device(self, d_hits, d_n_hits, hit_cap, d_qkmers, n_reads, d_pairs, d_read_offs, params=None, stream=None):
        """kmcpg_group_device on device pointers: hits as query_device left them -> (column, count) pairs grouped by read, filtered by
        -T, ordered per read; d_read_offs needs n_reads + 2 uint64 words."""
        p = params or default_params()
        _check(load().kmcpg_group_device(self._h, d_hits, d_n_hits, hit_cap, d_qkmers, n_reads, C.byref(p), d_pairs, d_read_offs, stream))

    def finalize_grouped(self, pairs, read_offs, qkmers, qlen, params=None, count_only=False):
        """kmcpg_finalize_grouped: pairs uint32 [m, 2] (or PAIR_DTYPE [m]), read_offs uint64 [n + 2] as group_device wrote them."""
        p = params or default_params()
        pairs = np.ascontiguousarray(pairs)
        read_offs = np.ascontiguousarray(read_offs, dtype=np.uint64)
        qkmers = np.ascontiguousarray(qkmers, dtype=np.int32)
        qlen = np.ascontiguousarray(qlen, dtype=np.int32)
        assert pairs.dtype.itemsize in (4, 8) and len(read_offs) == len(qkmers) + 2
        r = Result()
        n = len(qkmers)
        _check(load().kmcpg_finalize_grouped(self._h, pairs.ctypes.data, read_offs.ctypes.data, qkmers.ctypes.data, qlen.ctypes.data, n, C.byref(p), C.byref(r)))
        if count_only:
            m = int(r.match_offs[n]) if n else 0
            load().kmcpg_result_free(C.byref(r))
            return m
        return _copy_result(r)

    # ---- bench / parity support -----------------------------------------------------------------------
    def read_row_range(self, block, first_row, out):
        """rows first_row .. first_row+len(out)-1 of a resident block into `out` (uint8 [n, NumRowBytes], C-contiguous)."""
        assert out.flags["C_CONTIGUOUS"] and out.dtype == np.uint8
        _check(load().kmcpg_read_row_range(self._h, block, first_row, out.shape[0], out.ctypes.data))

    def plant(self, col, hashes):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        _check(load().kmcpg_plant(self._h, col, hashes.ctypes.data, len(hashes)))

    def plant_reads_device(self, d_seqs, d_offs, n_reads, total_bases, max_read_len, d_cols, stream=None):
        _check(load().kmcpg_plant_reads_device(self._h, d_seqs, d_offs, n_reads, total_bases, max_read_len, d_cols, stream))

    def set_profiling(self, on=True):
        """False/0 off, True/1 kernel timing, 2 timing + count the row loads of the COBS kernel."""
        _check(load().kmcpg_set_profiling(self._h, int(on)))

    def last_gathered_bytes(self):
        n = C.c_uint64()
        _check(load().kmcpg_last_gathered_bytes(self._h, C.byref(n)))
        return n.value

    def last_tail_waves(self):
        """Waves of the last query_device call's COBS kernels that finished in tail mode (profiling level 2)."""
        n = C.c_uint64()
        _check(load().kmcpg_last_tail_waves(self._h, C.byref(n)))
        return n.value

    def last_hash_bytes(self):
        """Bytes of k-mer hashes the COBS kernel(s) of the last query_device call read (profiling level 2)."""
        n = C.c_uint64()
        _check(load().kmcpg_last_hash_bytes(self._h, C.byref(n)))
        return n.value

    def last_timing(self, age=0):
        """(k-mer kernels ms, COBS kernel ms) of the last query_device call (age 1: the one before, ... up to 3)."""
        a, b = C.c_float(), C.c_float()
        _check(load().kmcpg_timing_at(self._h, age, C.byref(a), C.byref(b)))
        return a.value, b.value

    def read_rows(self, block, row_idx):
        row_idx = np.ascontiguousarray(row_idx, dtype=np.uint64)
        rb = self.block_info(block)["row_bytes"]
        out = np.zeros((len(row_idx), rb), dtype=np.uint8)
        _check(load().kmcpg_read_rows(self._h, block, row_idx.ctypes.data, len(row_idx), out.ctypes.data))
        return out
