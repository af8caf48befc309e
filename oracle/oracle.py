"""ctypes binding of the CPU parity oracle (oracle/kmcp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (kmcp_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkmcp_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "kmcp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "kmcp_oracle.h"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _LIB_PATH


class SketchCfg(C.Structure):
    _fields_ = [("k", C.c_int32), ("canonical", C.c_int32), ("scaled", C.c_int32), ("scale", C.c_uint32),
                ("minimizer", C.c_int32), ("minimizer_w", C.c_uint32), ("syncmer", C.c_int32),
                ("syncmer_s", C.c_uint32)]


class SearchParams(C.Structure):
    _fields_ = [("min_qlen", C.c_int32), ("min_matched", C.c_int32), ("min_qcov", C.c_double),
                ("min_tcov", C.c_double), ("max_fpr", C.c_double), ("dedup_threshold", C.c_int32),
                ("try_se", C.c_int32), ("fpr_buf_size", C.c_int32), ("sort_by", C.c_int32),
                ("do_not_sort", C.c_int32), ("top_n_scores", C.c_int32)]


class Match(C.Structure):
    _fields_ = [("block", C.c_uint32), ("col", C.c_uint32), ("col_global", C.c_uint32),
                ("target_idx", C.c_uint32), ("gsize", C.c_uint64), ("size", C.c_uint64),
                ("mkmers", C.c_int32), ("fpr", C.c_double), ("qcov", C.c_double), ("tcov", C.c_double),
                ("jacc", C.c_double), ("target", C.c_char_p)]


class Result(C.Structure):
    _fields_ = [("qlen", C.c_int32), ("qkmers", C.c_int32), ("k", C.c_int32), ("nmatches", C.c_int32),
                ("matches", C.POINTER(Match))]


class Column(C.Structure):
    _fields_ = [("name", C.c_char_p), ("gsize", C.c_uint64), ("chunk_idx", C.c_uint32), ("chunks", C.c_uint32),
                ("hashes", C.POINTER(C.c_uint64)), ("n_hashes", C.c_uint64)]


class BlockRules(C.Structure):
    """index.go:1453-1463 flags -x / -X / -8 / -1 (0 = default)."""
    _fields_ = [("kmers_x", C.c_uint64), ("block_size_x", C.c_int32), ("kmers_8", C.c_uint64), ("kmers_1", C.c_uint64)]


def default_params(**kw):
    """search.go:1052-1102 defaults."""
    p = SearchParams(min_qlen=30, min_matched=10, min_qcov=0.55, min_tcov=0.0, max_fpr=0.01,
                     dedup_threshold=256, try_se=0, fpr_buf_size=249, sort_by=0, do_not_sort=0, top_n_scores=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def sketch_cfg(k=21, canonical=True, scale=1, minimizer_w=0, syncmer_s=0):
    return SketchCfg(k=k, canonical=int(canonical), scaled=int(scale > 1), scale=scale,
                     minimizer=int(minimizer_w > 0), minimizer_w=minimizer_w, syncmer=int(syncmer_s > 0),
                     syncmer_s=syncmer_s)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u8p = C.POINTER(C.c_uint8)
    u64p = C.POINTER(C.c_uint64)
    L.ko_last_error.restype = C.c_char_p
    L.ko_nthash_kmer.restype = C.c_uint64
    L.ko_nthash_kmer.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.ko_nthash_all.restype = C.c_size_t
    L.ko_nthash_all.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, u64p]
    L.ko_max_hash.restype = C.c_uint64
    L.ko_max_hash.argtypes = [C.c_uint32]
    L.ko_generate_kmers.restype = C.c_size_t
    L.ko_generate_kmers.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(SketchCfg), u64p]
    L.ko_sort_unique.restype = C.c_size_t
    L.ko_sort_unique.argtypes = [u64p, C.c_size_t]
    L.ko_hash_values.argtypes = [C.c_uint64, C.c_int, u64p]
    L.ko_calc_signature_size.restype = C.c_uint64
    L.ko_calc_signature_size.argtypes = [C.c_uint64, C.c_int, C.c_double]
    L.ko_query_fpr.restype = C.c_double
    L.ko_query_fpr.argtypes = [C.c_int, C.c_int, C.c_double]
    L.ko_go_pow.restype = C.c_double
    L.ko_go_pow.argtypes = [C.c_double, C.c_double]
    L.ko_binomial_coeff.restype = C.c_double
    L.ko_binomial_coeff.argtypes = [C.c_int, C.c_int]
    L.ko_write_block.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, C.POINTER(Column), C.c_uint32]
    L.ko_build_db.argtypes = [C.c_char_p, C.POINTER(SketchCfg), C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(Column), C.c_uint32]
    L.ko_build_db2.argtypes = [C.c_char_p, C.POINTER(SketchCfg), C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(BlockRules), C.POINTER(Column),
                               C.c_uint32]
    L.ko_block_layout.argtypes = [C.POINTER(C.c_uint64), C.c_uint32, C.c_int, C.POINTER(BlockRules), C.POINTER(C.c_int)]
    L.ko_db_open.restype = C.c_void_p
    L.ko_db_open.argtypes = [C.c_char_p]
    L.ko_db_close.argtypes = [C.c_void_p]
    L.ko_db_create_mem.restype = C.c_void_p
    L.ko_db_create_mem.argtypes = [C.POINTER(SketchCfg), C.c_int, C.c_double, C.c_int, u64p, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_uint64]
    L.ko_db_info.argtypes = [C.c_void_p, C.POINTER(SketchCfg), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), u64p]
    L.ko_db_block_info.argtypes = [C.c_void_p, C.c_int, u64p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ko_db_block_rows.restype = C.c_void_p
    L.ko_db_block_rows.argtypes = [C.c_void_p, C.c_int]
    L.ko_db_col_name.restype = C.c_char_p
    L.ko_db_col_name.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), u64p, u64p]
    L.ko_search.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(SearchParams), C.POINTER(Result)]
    L.ko_result_free.argtypes = [C.POINTER(Result)]
    L.ko_block_counts.argtypes = [C.c_void_p, C.c_int, u64p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.ko_search_batch.restype = C.c_int64
    L.ko_search_batch.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint32, C.POINTER(SearchParams), C.c_int,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_int64]
    L.ko_search_batch_refshape.restype = C.c_int64
    L.ko_search_batch_refshape.argtypes = L.ko_search_batch.argtypes
    L.ko_format_match.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(Result), C.POINTER(Match), C.c_uint64]
    L.ko_set_seed_mode.argtypes = [C.c_int]
    _ = u8p
    _lib = L
    return L


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def nthash_all(seq: bytes, k: int, canonical=True) -> np.ndarray:
    n = max(0, len(seq) - k + 1)
    out = np.zeros(max(n, 1), dtype=np.uint64)
    m = lib().ko_nthash_all(seq, len(seq), k, int(canonical), _u64p(out))
    return out[:m]


def generate_kmers(seq: bytes, cfg: SketchCfg) -> np.ndarray:
    out = np.zeros(max(len(seq), 1), dtype=np.uint64)
    m = lib().ko_generate_kmers(seq, len(seq), C.byref(cfg), _u64p(out))
    return out[:m].copy()


def sort_unique(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    n = lib().ko_sort_unique(_u64p(a), len(a))
    return a[:n]


def block_layout(kmers, sblock, rules=None):
    """kmers ascending -> (number of blocks, 1-based block of every column; 0 = skipped)."""
    km = np.ascontiguousarray(kmers, dtype=np.uint64)
    out = np.zeros(len(km), dtype=np.int32)
    nb = lib().ko_block_layout(_u64p(km), len(km), sblock, C.byref(rules) if rules is not None else None, out.ctypes.data_as(C.POINTER(C.c_int)))
    return nb, out


def build_db(out_dir, cfg, columns, num_hashes=1, fpr=0.3, threads=32, block_size=0, rules=None):
    """columns: list of (name, gsize, chunk_idx, chunks, np.uint64 sorted-unique hashes)."""
    arr = (Column * len(columns))()
    keep = []
    for i, (name, gsize, ci, nchunks, h) in enumerate(columns):
        h = np.ascontiguousarray(h, dtype=np.uint64)
        keep.append(h)
        arr[i] = Column(name.encode(), gsize, ci, nchunks, _u64p(h), len(h))
    rc = lib().ko_build_db2(out_dir.encode(), C.byref(cfg), num_hashes, fpr, threads, block_size, C.byref(rules) if rules is not None else None, arr,
                            len(columns))
    if rc != 0:
        raise RuntimeError(lib().ko_last_error().decode())
    return os.path.join(out_dir, "R001")


class OracleDB:
    def __init__(self, db_dir=None, _handle=None, _keep=None):
        self._keep = _keep
        self.h = _handle if _handle is not None else lib().ko_db_open(db_dir.encode())
        if not self.h:
            raise RuntimeError(lib().ko_last_error().decode())
        cfg = SketchCfg()
        nh, fpr, nb, nc = C.c_int(), C.c_double(), C.c_int(), C.c_uint64()
        lib().ko_db_info(self.h, C.byref(cfg), C.byref(nh), C.byref(fpr), C.byref(nb), C.byref(nc))
        self.cfg, self.num_hashes, self.fpr, self.nblocks, self.ncols = cfg, nh.value, fpr.value, nb.value, nc.value

    @classmethod
    def from_memory(cls, cfg, num_hashes, fpr, blocks, size_all):
        """blocks: list of (num_sigs, ncols, col_base, rows uint8[num_sigs, row_bytes])."""
        n = len(blocks)
        ns = np.array([b[0] for b in blocks], dtype=np.uint64)
        nc = np.array([b[1] for b in blocks], dtype=np.uint32)
        cb = np.array([b[2] for b in blocks], dtype=np.uint32)
        mats = [np.ascontiguousarray(b[3], dtype=np.uint8) for b in blocks]
        ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in mats])
        h = lib().ko_db_create_mem(C.byref(cfg), num_hashes, fpr, n, _u64p(ns), nc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   cb.ctypes.data_as(C.POINTER(C.c_uint32)), ptrs, size_all)
        return cls(_handle=h, _keep=mats)

    def close(self):
        if self.h:
            lib().ko_db_close(self.h)
            self.h = None

    def block_info(self, b):
        ns, nc, rb = C.c_uint64(), C.c_uint32(), C.c_uint32()
        lib().ko_db_block_info(self.h, b, C.byref(ns), C.byref(nc), C.byref(rb))
        return ns.value, nc.value, rb.value

    def block_rows(self, b):
        ns, nc, rb = self.block_info(b)
        ptr = lib().ko_db_block_rows(self.h, b)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(ns, rb))

    def col_info(self, cg):
        ti, gs, sz = C.c_uint32(), C.c_uint64(), C.c_uint64()
        name = lib().ko_db_col_name(self.h, cg, C.byref(ti), C.byref(gs), C.byref(sz))
        return name.decode(), ti.value, gs.value, sz.value

    def block_counts(self, b, kmers):
        ns, nc, rb = self.block_info(b)
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(nc, dtype=np.uint32)
        lib().ko_block_counts(self.h, b, _u64p(kmers), len(kmers), out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def search(self, seq1: bytes, seq2: bytes = None, params=None):
        """Returns dict(qlen, qkmers, k, matches=None|[dict...])."""
        p = params or default_params()
        r = Result()
        rc = lib().ko_search(self.h, seq1, len(seq1), seq2, len(seq2) if seq2 is not None else 0, C.byref(p), C.byref(r))
        if rc != 0:
            raise RuntimeError(lib().ko_last_error().decode())
        res = dict(qlen=r.qlen, qkmers=r.qkmers, k=r.k, matches=None)
        if r.nmatches >= 0:
            ms = []
            for i in range(r.nmatches):
                m = r.matches[i]
                ms.append(dict(block=m.block, col=m.col, col_global=m.col_global, target=m.target.decode(),
                               chunk_idx=m.target_idx & 0xFFFF, chunks=m.target_idx >> 16, tlen=m.gsize,
                               size=m.size, mkmers=m.mkmers, fpr=m.fpr, qcov=m.qcov, tcov=m.tcov, jacc=m.jacc))
            res["matches"] = ms
        lib().ko_result_free(C.byref(r))
        return res

    def search_batch(self, seqs: np.ndarray, offs: np.ndarray, params=None, threads=0, refshape=False):
        """Threaded single-end batch (cpu_baseline).  Returns (qkmers[n], hits[m,3] sorted by (read,col)).
        refshape: the reference's own loop shape (one worker per block, 64-row byte transposition + Count8)."""
        p = params or default_params()
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        qk = np.zeros(n, dtype=np.int32)
        cap = max(1024, 64 * n)
        while True:
            hits = np.zeros((cap, 3), dtype=np.uint32)
            fn = lib().ko_search_batch_refshape if refshape else lib().ko_search_batch
            m = fn(self.h, seqs.ctypes.data, _u64p(offs), n, C.byref(p), threads,
                                      qk.ctypes.data_as(C.POINTER(C.c_int32)),
                                      hits.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
            if m <= cap:
                break
            cap = int(m)
        hits = hits[:m]
        order = np.lexsort((hits[:, 1], hits[:, 0]))
        return qk, hits[order]
