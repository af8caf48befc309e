/*
 * kmcp_gpu.h — C ABI of libkmcpgpu.so: the MI355X-native replacement of the `kmcp search` hot path.
 *
 * The reference (shenwei356/kmcp v0.9.5, pure Go) has no FFI seam; the seam is a pair of Go channel
 * protocols inside package cmd.  Each entry point below names the reference code it
 * replaces (paths relative to kmcp/cmd/).  The cgo binding a maintainer would add is shim/kmcp_gpu.go
 * (see INTEGRATION.md).
 *
 * Conventions: every function returns 0 on success or a negative KMCPG_E* code; the message is
 * available from kmcpg_last_error() (thread-local).  No exception or abort crosses the ABI.  Input
 * buffers are borrowed for the duration of the call only; buffers returned inside kmcpg_result are
 * owned by the library and released by kmcpg_result_free().  A kmcpg_db may be used from several OS
 * threads: the enqueueing of GPU work on one handle is serialised internally, and consecutive kmcpg_query_device calls
 * on different streams are ordered by an event (they share the handle's k-mer workspace), so they never overlap on the GPU.
 */
#ifndef KMCP_GPU_H
#define KMCP_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMCPG_OK 0
#define KMCPG_EINVAL (-1)   /* bad argument */
#define KMCPG_EIO (-2)      /* file missing / unreadable */
#define KMCPG_EFORMAT (-3)  /* invalid or incompatible .uniki / __db.yml (serialization.go:41-57) */
#define KMCPG_EDEVICE (-4)  /* HIP error, no GPU */
#define KMCPG_ENOMEM (-5)
#define KMCPG_EUNSUPPORTED (-6)
#define KMCPG_EBUSY (-7)    /* kmcpg_submit: every lane of the handle is in flight */

typedef struct kmcpg_db kmcpg_db;

/* How the database is placed on the GPU(s).  One process drives one GPU (one rank); the index's
 * independent .uniki blocks are partitioned over `shard_count` ranks by bytes. */
typedef struct {
  int32_t device;      /* HIP device ordinal of this process; -1 = metadata only (headers parsed, nothing resident:
                          kmcpg_db_info/col_info/block_info/finalize work, every GPU entry point fails) */
  int32_t shard_rank;  /* 0..shard_count-1 */
  int32_t shard_count; /* >=1 */
  int32_t reserved;
} kmcpg_opts;

/* What `search` needs to know about the database: UnikIndexDBInfo (util-db-info.go:46-79) + Header
 * (index/serialization.go:66-82). */
typedef struct {
  int32_t k;
  int32_t canonical;
  int32_t num_hashes;
  int32_t scaled;
  uint32_t scale;
  int32_t minimizer;
  uint32_t minimizer_w;
  int32_t syncmer;
  uint32_t syncmer_s;
  double fpr;             /* DB-wide false-positive rate of one Bloom filter (`fpr` in __db.yml) */
  int32_t n_blocks;       /* all .uniki files of the DB */
  int32_t n_blocks_local; /* blocks resident on this rank's GPU */
  uint64_t n_cols;        /* reference chunks (columns) over all blocks */
  uint64_t matrix_bytes;       /* on-disk bit-matrix bytes over all blocks */
  uint64_t matrix_bytes_local; /* on-disk bit-matrix bytes of the local blocks */
  uint64_t row_bytes_sum_local; /* sum over local blocks of NumRowBytes: algorithmic bytes per (k-mer, hash) */
} kmcpg_info;

/* SearchOptions (util-db-search.go:162-189); defaults are those of search.go:1052-1102. */
typedef struct {
  int32_t min_qlen;        /* -m/--min-query-len 30 */
  int32_t min_matched;     /* -c/--min-kmers 10 */
  double min_qcov;         /* -t/--min-query-cov 0.55 */
  double min_tcov;         /* -T/--min-target-cov 0 */
  double max_fpr;          /* -f/--max-fpr 0.01 */
  int32_t dedup_threshold; /* -u/--kmer-dedup-threshold 256 */
  int32_t try_se;          /* --try-se */
  int32_t sort_by;         /* -s: 0 qcov, 1 tcov, 2 jacc */
  int32_t do_not_sort;     /* -S */
  int32_t top_n_scores;    /* -n/--keep-top-scores */
  int32_t fpr_buf_size;    /* 249 single-end / 499 paired-end (search.go:250-255); 0 = pick */
  int32_t k;               /* 0 = the database's k-mer sizes, largest first, smaller ones for queries that matched nothing
                              (util-db-search.go:764, :1016-1022); > 0 = this size only (must be one of the database's) */
  int32_t reserved;
} kmcpg_params;

/* One (read, reference chunk) pair that passed the integer thresholds on the GPU:
 * count >= max(min_matched, smallest c with float64(c) > n*min_qcov)   (util-db-search.go:7468-7470),
 * and, for queries of up to 1024 k-mers, count >= the smallest c whose FPR(n, c) <= max_fpr (:7474-7478): a pair that
 * kmcpg_finalize would drop for its FPR anyway is not reported (KMCPG_FPR_BOUND=0 reports it). */
typedef struct {
  uint32_t read;  /* index of the read in the batch */
  uint32_t col;   /* global column: columns numbered over the blocks in __db.yml `files` order */
  uint32_t count; /* matched k-mers (mKmers) */
} kmcpg_hit;

/* Match (util-db-search.go:83-93) without the strings; names come from kmcpg_col_info(). */
typedef struct {
  uint32_t col;
  uint32_t target_idx; /* chunkIdx | chunks<<16 (index.go:1096) */
  uint64_t gsize;
  int32_t mkmers;
  int32_t reserved;
  double fpr, qcov, tcov, jacc;
} kmcpg_match;

/* QueryResult (util-db-search.go:60-74) for a batch, CSR over reads. */
typedef struct {
  uint32_t n_reads;
  int32_t k;
  int32_t* qlen;         /* [n_reads] QueryLen (read1+read2 for paired-end; the searched mate after --try-se) */
  int32_t* qkmers;       /* [n_reads] NumKmers (0 when the query was not searched) */
  int32_t* ksize;        /* [n_reads] QueryResult.K: the k-mer size the query was last searched with (differs from `k` only in
                            databases with several k-mer sizes) */
  uint64_t* match_offs;  /* [n_reads+1] */
  kmcpg_match* matches;  /* [match_offs[n_reads]] sorted as handleQuerySingleDB does (:273-282) */
  void* owner;           /* internal */
} kmcpg_result;

/* -- lifecycle: replaces NewUnikIndexDB / NewUnikIndex (util-db-search.go:648-743, 1196-1280) and
 *    UnikIndexDB.Close (:1119-1150).  db_dir is the directory holding __db.yml (e.g. <db>/R001). */
int kmcpg_open(const char* db_dir, const kmcpg_opts* opts, kmcpg_db** out);
/* One host process, several GPUs (what a cgo host needs): the blocks are partitioned over `devices`, kmcpg_search_batch fans
 * every batch out to all of them from one host thread per GPU; the per-read hit lists of the shards are gathered on the first
 * GPU with RCCL (grouped ncclSend/ncclRecv over xGMI of exactly the bytes each shard produced; librccl is bound at run time)
 * and reach the host in one copy — the reference's concatenation of its per-block workers' replies (util-db-search.go:939-964).
 * When RCCL cannot serve (ordinals repeat — RCCL wants one rank per device —, no librccl, KMCPG_RCCL=0, its self-test at open
 * fails) the shards' lists are copied to the host one by one and merged there; kmcpg_exchange_info says which it is.
 * kmcpg_query_device is not available on such a handle. */
int kmcpg_open_devices(const char* db_dir, const int32_t* devices, int32_t n_devices, kmcpg_db** out);
/* "RCCL gather over N device(s)" / "host merge of the shards' hit lists (<reason>)" / "single device: no exchange step";
 * the string lives until the calling thread's next call of this function */
const char* kmcpg_exchange_info(const kmcpg_db* db);
/* A database LARGER than the HBM at hand, on one GPU (the reference searches any size through mmap / --low-mem,
 * util-db-search.go:1238-1280, :6975-7335; search.go:80): the index is cut into `passes` shards (the byte-balanced partition of
 * kmcpg_open with shard_count = passes; 0 = as few as fit the free HBM of `device`), and kmcpg_search_batch / kmcpg_submit
 * search every batch against one resident shard after the other — upload shard, K1 + K2, keep the hit tuples, next shard —
 * then finalize the concatenated hit lists once: results are those of a resident database.  The search of a batch runs inside
 * kmcpg_submit on such a handle (it blocks); the shard searched last stays resident for the next batch, so a batch costs
 * passes - 1 uploads (11-44 GB/s from the page cache): use large batches.  If the index fits (1 pass) this is kmcpg_open.
 * kmcpg_open itself reports an index that does not fit with KMCPG_ENOMEM and the bytes needed / free in the message. */
int kmcpg_open_paged(const char* db_dir, int32_t device, int32_t passes, kmcpg_db** out);
/* passes of a paged handle (0 for every other handle) and how many shard uploads it has done so far */
int kmcpg_paged_info(const kmcpg_db* db, int32_t* passes, uint64_t* uploads);
/* How many bases (read 1 + read 2) one batch may hold on this handle so that its device workspace fits beside the resident index
 * (K1 and the sort + unique of long queries take up to 24 B per base, query.cpp): a paged handle answers from the HBM it kept
 * free beside its largest shard, every other handle from what is free now; 0 = unknown.  A batch that does not fit is not an
 * error of kmcpg_search_batch either: it is searched as two halves (recursively) and answered as one result. */
int kmcpg_batch_hint(const kmcpg_db* db, uint64_t* max_bases);
int kmcpg_close(kmcpg_db* db);
const char* kmcpg_last_error(void);
int kmcpg_db_info(const kmcpg_db* db, kmcpg_info* info);
/* The k-mer sizes of the database, largest first (`ks` of __db.yml, util-db-info.go:50; one entry for most databases):
 * *n = how many there are, the first min(*n, cap) are written to ks.  kmcpg_search_batch walks them by itself
 * (util-db-search.go:764, :1016-1022); a host that drives kmcpg_query_device + kmcpg_finalize per shard repeats the
 * search of its unmatched queries with kmcpg_params.k set to the next entry. */
int kmcpg_db_ks(const kmcpg_db* db, int32_t* ks, int32_t cap, int32_t* n);
/* Header.Names/GSizes/Indices/Sizes of one column (serialization.go:73-79) */
int kmcpg_col_info(const kmcpg_db* db, uint32_t col, const char** name, uint32_t* target_idx, uint64_t* gsize,
                   uint64_t* size);

/* -- the whole per-query pipeline for a batch of queries: replaces UnikIndexDB.handleQuery
 *    (util-db-search.go:763-1025) and the sorting/top-N of handleQuerySingleDB (:260-345).
 *    seqs/offs: read i is seqs[offs[i]..offs[i+1]) (ASCII, as fastx delivers it);
 *    seqs2/offs2: mates for paired-end input or NULL. */
int kmcpg_search_batch(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2,
                       const uint64_t* offs2, uint32_t n_reads, const kmcpg_params* params, kmcpg_result* out);
void kmcpg_result_free(kmcpg_result* r);

/* -- the same pipeline, asynchronous: the reference keeps 30 x threads queries in flight (util-db-search.go:243, :347-351);
 *    here a few BATCHES are.  kmcpg_submit copies the batch into pinned staging (the caller's buffers are free again when it
 *    returns), enqueues H2D copy, kernels and D2H copy on the handle's private stream and returns.  It never blocks: with all
 *    lanes (KMCPG_INFLIGHT, default 4) in flight it fails with KMCPG_EBUSY and the caller waits for one of its tickets first
 *    (kmcpg_search_batch waits for a lane instead, so a thread must not call it while it holds every lane itself).
 *    kmcpg_wait blocks until that batch's GPU work is done and runs the host half (float64 thresholds, FPR, sorting;
 *    --try-se / smaller-k retries) on the calling thread while later batches occupy the GPU.
 *    Tickets may be waited for in any order and from any thread; kmcpg_wait consumes the ticket, also when it fails.
 *    kmcpg_close refuses (KMCPG_EBUSY) while tickets are outstanding.
 *    kmcpg_search_batch(...) == kmcpg_submit(...) + kmcpg_wait(...). */
typedef struct kmcpg_ticket kmcpg_ticket;
int kmcpg_submit(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2,
                 uint32_t n_reads, const kmcpg_params* params, kmcpg_ticket** out);
int kmcpg_wait(kmcpg_ticket* ticket, kmcpg_result* out);

/* -- the GPU half only (k-mer generation + COBS query on the local blocks), device-resident in and
 *    out: generateKmers (util-db-search.go:1037-1107) + dedup (:874-908) + the UnikIndex workers
 *    (:6611-7742, integer thresholds only).  All pointers are DEVICE pointers; `stream` is a
 *    hipStream_t (NULL = default stream).  d_counters points at TWO 64-bit words: [0] receives the number of hits
 *    produced (which may exceed hit_cap: then only hit_cap were stored and the caller retries with a larger buffer), [1] the
 *    largest NumKmers of the batch — if it exceeds what max_read_len allows (max_read_len - k + 1, twice that for pairs),
 *    max_read_len was under-reported, the match counters may have wrapped and the hits must be discarded.
 *    d_qkmers[i] receives NumKmers of read i (0 if not searched), d_qlen[i] its QueryLen.
 *    max_read_len must be >= the longest read (mate) of the batch: it sizes the counters.  The call only enqueues work on
 *    `stream` for short-read batches; when a query may exceed 32 768 k-mers — or 2048, in a batch too small to fill the GPU
 *    by itself — it reads 8 bytes back (which queries are long is known only on the device; they may take the chunked form of
 *    the kernel) and therefore synchronises the stream once or twice.
 *    The hit list depends on params->min_qcov, min_matched AND max_fpr (see kmcpg_hit: counts that cannot pass -f are left out
 *    on the device): finalize it with the SAME params — a looser max_fpr/min_qcov in kmcpg_finalize cannot bring back what
 *    the GPU already dropped (a stricter one is fine, kmcpg_finalize applies every threshold again). */
int kmcpg_query_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, const uint8_t* d_seqs2,
                       const uint64_t* d_offs2, uint32_t n_reads, uint64_t total_bases, uint32_t max_read_len,
                       const kmcpg_params* params, kmcpg_hit* d_hits, uint64_t hit_cap, uint64_t* d_counters,
                       int32_t* d_qkmers, int32_t* d_qlen, void* stream);

/* -- the host half: thresholds that need float64/FPR (util-db-search.go:7471-7489), Match values,
 *    sorting, --keep-top-scores.  `hits` may be the concatenation of the hit lists of all shards
 *    (any order).  Every rank knows every column's metadata, so this can run on the gathering rank.
 *    `params` must not be looser (max_fpr, min_qcov, min_matched) than the ones the hits were produced with by
 *    kmcpg_query_device, which already filters on them. */
int kmcpg_finalize(const kmcpg_db* db, const kmcpg_hit* hits, uint64_t n_hits, const int32_t* qkmers,
                   const int32_t* qlen, uint32_t n_reads, const kmcpg_params* params, kmcpg_result* out);

/* -- the host half split in two (round 4): its grouping, -T filter and per-query ordering run on the GPU, the host only expands
 *    (column, count) pairs to Match records.  kmcpg_search_batch / kmcpg_submit use this by themselves (KMCPG_DEVICE_FINALIZE=0
 *    keeps the hits on the round-3 path through kmcpg_finalize); a host that drives kmcpg_query_device per shard gathers the
 *    shards' hit lists on one GPU (RCCL) and calls kmcpg_group_device there.
 *    kmcpg_group_device: d_hits[0 .. min(*d_n_hits, hit_cap)) as kmcpg_query_device left them (any order, any number of shards
 *    concatenated) -> d_pairs: the matches of read i at [d_read_offs[i], d_read_offs[i+1]), those failing -T (count / size <
 *    min_tcov in float64, util-db-search.go:7471-7473) dropped, ordered as handleQuerySingleDB orders them (:260-283, :105-145:
 *    -s qcov/tcov/jacc, ties by column; -S: column order).  Segments longer than 4096 matches come back grouped but unordered
 *    (kmcpg_finalize_grouped sorts those).  d_pairs needs room for hit_cap pairs, d_read_offs for n_reads + 2 words: the last
 *    one receives the number of hits that named a read or column that does not exist (must be 0).  Only enqueues on `stream`.
 *    kmcpg_finalize_grouped: the float64 Match values (qCov, tCov, jacc :7487-7489), the FPR column and its -f test (:7474-7478),
 *    --keep-top-scores (:285-311) and the name-independent metadata; same result as kmcpg_finalize on the same hits.  Segments
 *    that are not in that order (longer than 4096 matches, or a list that did not come from kmcpg_group_device) are sorted here. */
typedef struct {
  uint32_t col;   /* global column */
  uint32_t count; /* matched k-mers */
} kmcpg_pair;
/* -- compact results (round 5): on a database full of close relatives a read has hundreds of matches, and writing a 56-byte Match
 *    record for each (1.5 GB per 131 072 reads) was what bounded kmcpg_search_batch (6.8 M reads/s against 12.7 M for the kernels).
 *    The *_pairs forms return the FINAL matches of every query — every threshold, -f and --keep-top-scores applied, in the order
 *    kmcp search prints them — as 8-byte (column, mKmers) pairs; match_offs[i+1] - match_offs[i] is the query's `hits` column.
 *    Reference counterpart: the Match structs a worker appends (util-db-search.go:7479-7489) and the row loop that prints them
 *    (search.go:517-575) — here the second is what needs the first, one query at a time.
 *    kmcpg_expand_pairs derives the Match records of one query's pairs (qCov, tCov, jacc util-db-search.go:7487-7489, the FPR column
 *    util-fpr.go:32-50, column metadata) on the caller's thread, typically into a small scratch array right before the rows are formatted: the same bits as
 *    kmcpg_search_batch / kmcpg_wait would have written.  Everything else (arguments, errors, ownership: kmcpg_result_pairs_free)
 *    is as for the record forms; a ticket is consumed by EITHER kmcpg_wait or kmcpg_wait_pairs. */
typedef struct {
  uint32_t n_reads;
  int32_t k;
  int32_t* qlen;
  int32_t* qkmers;
  int32_t* ksize;
  uint64_t* match_offs;  /* [n_reads+1] */
  kmcpg_pair* pairs;     /* [match_offs[n_reads]] */
  void* owner;           /* internal */
} kmcpg_result_pairs;
/* -- packed queries (round 6): long queries (genomes, contigs, HiFi reads) as 2-bit codes.  A batch of 256 assemblies is 1 GB of ASCII;
 *    kmcpg_submit reads it once more to pack it into pinned staging (at 29 k genomes/s the host reads 117 GB/s of text for that).  A reader
 *    that packs where it first touches the bases (kmcp-search -g does: cli/kmcp_search.cpp) hands the library a quarter of the bytes and no
 *    second pass.  Reference counterpart of the input: the sequence a Query carries (util-db-search.go:50-57, search.go:885-915 for -g).
 *    Layout: base j of the batch (all queries back to back, offs in BASES as for kmcpg_submit) sits in bits 2*(j % 4) of codes[j / 4];
 *    code = (ascii >> 1) & 3, i.e. A/a 0, C/c 1, T/t/U/u 2, G/g 3.  Every other byte (N, IUPAC codes, gaps ...) is listed as a run
 *    {first base, length, the byte}: it reaches the k-mer kernels verbatim (its code bits are ignored).  Runs must lie inside the batch and
 *    must not overlap.  Exactness: the k-mer kernels see a base only through the ntHash seed tables, where all spellings of a base share an
 *    entry (kmcp_amd/csrc/pack2.hpp) — results are those of kmcpg_submit on the text, bit for bit (tests/test_gpu_pack.py).
 *    Single-end only.  Handles that must re-read the text (several k-mer sizes, paged indexes) unpack it on the host first: correct, slower.
 *    kmcpg_pack2 appends n bases of text at base position `pos` of a codes array (any alignment; codes needs (pos + n + 3) / 4 + 8 bytes)
 *    and writes the runs it met to exc[*n_exc ...] (positions absolute); when exc_cap is too small it returns KMCPG_ENOMEM with *n_exc =
 *    the number needed in total (call again with room: packing the same bases twice is harmless).  kmcpg_unpack2 is its inverse
 *    (canonical spelling A C G T for the coded bases). */
typedef struct {
  uint64_t pos;  /* first base of the run (position in the batch) */
  uint32_t len;
  uint32_t byte; /* the ASCII byte of every base of the run */
} kmcpg_exc_run;
int kmcpg_pack2(const uint8_t* seq, uint64_t n, uint64_t pos, uint8_t* codes, kmcpg_exc_run* exc, uint64_t exc_cap, uint64_t* n_exc);
/* Page-locked host memory for a reader's batches.  Codes that live in memory from kmcpg_host_alloc are uploaded from where they are —
 * kmcpg_submit_packed makes no staging copy of them (a quarter of a gigabyte per batch of 256 assemblies: the copy took longer than the GPU
 * needs for the batch) — and must then stay untouched until kmcpg_wait / kmcpg_wait_pairs has returned for the ticket.  Codes in any other
 * memory are copied inside the call as every other input is.  (offs and exc are small and always copied.) */
int kmcpg_host_alloc(uint64_t bytes, void** out);
int kmcpg_host_free(void* p);
int kmcpg_unpack2(const uint8_t* codes, uint64_t n_bases, const kmcpg_exc_run* exc, uint64_t n_exc, uint8_t* out);
int kmcpg_submit_packed(kmcpg_db* db, const uint8_t* codes, const uint64_t* offs, const kmcpg_exc_run* exc, uint64_t n_exc, uint32_t n_reads,
                        const kmcpg_params* params, kmcpg_ticket** out);

int kmcpg_search_batch_pairs(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2,
                             uint32_t n_reads, const kmcpg_params* params, kmcpg_result_pairs* out);
int kmcpg_wait_pairs(kmcpg_ticket* ticket, kmcpg_result_pairs* out);
void kmcpg_result_pairs_free(kmcpg_result_pairs* r);
int kmcpg_expand_pairs(const kmcpg_db* db, int32_t qkmers, const kmcpg_pair* pairs, uint64_t n, kmcpg_match* out);

int kmcpg_group_device(kmcpg_db* db, const kmcpg_hit* d_hits, const uint64_t* d_n_hits, uint64_t hit_cap, const int32_t* d_qkmers,
                       uint32_t n_reads, const kmcpg_params* params, kmcpg_pair* d_pairs, uint64_t* d_read_offs, void* stream);
int kmcpg_finalize_grouped(const kmcpg_db* db, const kmcpg_pair* pairs, const uint64_t* read_offs, const int32_t* qkmers,
                           const int32_t* qlen, uint32_t n_reads, const kmcpg_params* params, kmcpg_result* out);

/* -- benchmarking / full-size parity support (no counterpart in the reference) ------------------ */
typedef struct {
  int32_t k;            /* 21 */
  int32_t num_hashes;   /* 1 */
  double fpr;           /* 0.3: the fullest column has bit density 1-exp(-1/ratio) = fpr */
  uint32_t n_blocks;    /* e.g. 32 */
  uint32_t cols_per_block; /* e.g. 14976 */
  uint64_t num_sigs;    /* rows per block, e.g. 970000 */
  uint64_t kmers_per_col; /* Sizes[] of every column, e.g. 346000 */
  uint64_t seed;
  uint32_t scale;       /* 0/1 = all k-mers; >1 = FracMinHash database (`scaled: true`, `scale`) */
  uint32_t syncmer_s;   /* >0 = Closed-Syncmer database */
  uint32_t minimizer_w; /* >0 = Minimizer database */
  uint32_t sigs_step;   /* block i has num_sigs + i*sigs_step rows.  Real databases have a different NumSigs in (almost) every
                           block; blocks with equal NumSigs are laid side by side in HBM and served by one gather */
} kmcpg_synth_spec;
/* Builds a synthetic database directly in HBM: every bit i.i.d. Bernoulli(density) from a
 * counter-based generator keyed by (seed, block, row, word).  Column names are "syn<global col>". */
int kmcpg_open_synthetic(const kmcpg_synth_spec* spec, const kmcpg_opts* opts, kmcpg_db** out);
/* ORs the Bloom bits of the given k-mer hashes into column `col` (global id) if it is local. */
int kmcpg_plant(kmcpg_db* db, uint32_t col, const uint64_t* hashes, uint64_t n);
/* Device-resident variant for whole batches: ORs the Bloom bits of every k-mer of read i (hashed exactly as a
 * query would be) into global column d_cols[i]; 0xFFFFFFFF = do not plant; non-local columns are skipped. */
int kmcpg_plant_reads_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, uint32_t n_reads,
                             uint64_t total_bases, uint32_t max_read_len, const uint32_t* d_cols, void* stream);
/* HIP-event timing of the kernels inside kmcpg_query_device (events on the caller's stream).
 * kmcpg_last_timing waits for the last call to finish; times are milliseconds. */
int kmcpg_set_profiling(kmcpg_db* db, int enable);  /* 0 off, 1 timing, 2 timing + count the row loads of the COBS kernel */
/* Bytes the COBS kernel(s) of the last kmcpg_query_device call asked the memory system for (16 B per lane and row actually
 * loaded, row padding included, pruned rows not): a live cross-check of the FETCH_SIZE counter passes.  Level 2 only. */
int kmcpg_last_gathered_bytes(kmcpg_db* db, uint64_t* bytes);
/* ... and the bytes of k-mer hashes the same launches read (8 B per k-mer, once per (read, slot): 0.8 % of the row bytes for
 * 1-KB row tiles, 6 % for 128-byte rows).  Rows + hashes is what FETCH_SIZE sees.  Level 2 only. */
int kmcpg_last_hash_bytes(kmcpg_db* db, uint64_t* bytes);
/* ... and how many waves of those launches finished in tail mode (long queries on 1-KiB row tiles whose sectors had died down
 * to at most four: the idle lanes take shares of the remaining rows, k2_cobs.hip; KMCPG_TAIL_SECTORS=0 switches it off).  Level 2 only. */
int kmcpg_last_tail_waves(kmcpg_db* db, uint64_t* waves);
int kmcpg_last_timing(kmcpg_db* db, float* kmers_ms, float* cobs_ms);
/* The same for an earlier call: age 0 = the last one, 1 = the one before ... (the last 4 are kept), so that a caller with
 * several batches in flight can read the times of a finished one without waiting for the newest. */
int kmcpg_timing_at(kmcpg_db* db, uint32_t age, float* kmers_ms, float* cobs_ms);
/* Copies rows (on-disk width NumRowBytes each) of a local block back to the host. */
int kmcpg_read_rows(kmcpg_db* db, uint32_t block, const uint64_t* row_idx, uint64_t n_rows, uint8_t* out);
/* The same for rows first_row .. first_row + n_rows - 1 (bench: the index goes back to the host for the CPU baseline). */
int kmcpg_read_row_range(kmcpg_db* db, uint32_t block, uint64_t first_row, uint64_t n_rows, uint8_t* out);
/* Geometry of block b (global index): NumSigs, columns, NumRowBytes, device row stride, is-local. */
int kmcpg_block_info(const kmcpg_db* db, uint32_t block, uint64_t* num_sigs, uint32_t* n_cols, uint32_t* row_bytes,
                     uint32_t* dev_stride, int32_t* is_local, uint32_t* col_base);
/* k-mer generation only (K1): returns hashes of read i at d_hashes[d_koff[i] .. +d_nk[i]). Debug/tests. */
int kmcpg_kmers_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, uint32_t n_reads,
                       uint64_t total_bases, uint32_t max_read_len, const kmcpg_params* params,
                       uint64_t* d_hashes, uint64_t hashes_cap, uint64_t* d_koff, int32_t* d_nk, void* stream);
/* The same on a batch that is on the device as 2-bit codes (the layout of kmcpg_submit_packed: base j of the batch in bits 2 (j % 4)
 * of d_codes[j / 4], 16 readable bytes behind the last one, 4-byte aligned; d_exc = its n_exc runs of foreign bytes on the device)
 * and d_text = total_bases + 16 writable bytes for whatever has to be expanded: whole genomes (k <= 128, plain or FracMinHash k-mers)
 * are hashed from the codes directly and only the 65 536-position segments a foreign byte reaches become text (k1_kmers.hip);
 * every other shape of batch is expanded whole first.  Debug/tests. */
int kmcpg_kmers_device_packed(kmcpg_db* db, const uint8_t* d_codes, const kmcpg_exc_run* d_exc, uint32_t n_exc, uint8_t* d_text,
                              const uint64_t* d_offs, uint32_t n_reads, uint64_t total_bases, uint32_t max_read_len,
                              const kmcpg_params* params, uint64_t* d_hashes, uint64_t hashes_cap, uint64_t* d_koff, int32_t* d_nk,
                              void* stream);
/* Batches of this handle so far whose k-mer kernels read 2-bit codes directly (packed whole-genome batches), and packed batches that
 * were expanded to text first. */
int kmcpg_k1_codes_batches(kmcpg_db* db, uint64_t* direct, uint64_t* expanded);

/* -- index building on the GPU ("next" row of SURVEY.md §8f): the Bloom-column scatter of `kmcp index`
 *    (kmcp/cmd/index.go:657-682 block layout, :1023 signature size, :1107-1309 scatter, index/serialization.go:159-300 file,
 *    util-db-info.go:46-79 __db.yml) from lists of k-mer hashes — what the .unik files of `kmcp compute` hold.  Writes
 *    <out_dir>/R001/{_blockNNN.uniki, __db.yml, __name_mapping.tsv} byte-compatible with the reference's reader.  The block
 *    layout includes the big-genome rules (index.go:787-894: smaller blocks above -x/-8, single-column blocks above -1). */
typedef struct {
  int32_t k;
  int32_t canonical;
  int32_t num_hashes;   /* -n */
  double fpr;           /* -f */
  int32_t threads;      /* -j: block size = (int(#cols/threads)+7)/8*8 clamped to [8, #cols] when block_size == 0 */
  int32_t block_size;   /* -b */
  uint32_t scale;       /* recorded in __db.yml (scaled: scale > 1) */
  uint32_t minimizer_w;
  uint32_t syncmer_s;
  int32_t split_seq, split_size, split_num, split_overlap; /* recorded in __db.yml */
  const char* alias;
  /* big-genome block rules, flags -x / -X / -8 / -1 of `kmcp index` (index.go:1453-1463); 0 = the reference's default */
  uint64_t kmers_x;     /* -x 10M (M = 2^20): columns with more k-mers go to blocks of block_size_x columns */
  int32_t block_size_x; /* -X 256 */
  int32_t uniform_sigs; /* 0 = NumSigs per block as `kmcp index` sizes it (index.go:936-946, :1023: files byte-identical to the
                           reference's).  Not in the reference: 1 = every block of a size tier gets the tier's largest NumSigs,
                           2 = NumSigs rounded up to a 5/4 ladder (< 25 % larger files).  Blocks with equal NumSigs are served
                           by ONE gather when resident (grouped rows): a `-j 32` index of 39-byte rows runs at the speed of
                           1248-byte rows.  A larger filter only lowers a block's FPR; any kmcp reader accepts the files. */
  uint64_t kmers_8;     /* -8 20M: ... to blocks of 8 columns */
  uint64_t kmers_1;     /* -1 200M: ... to a block of their own */
} kmcpg_build_cfg;
typedef struct {
  const char* name;       /* reference name */
  uint64_t gsize;         /* genome size */
  uint32_t chunk_idx;     /* index of this chunk */
  uint32_t chunks;        /* number of chunks of the genome */
  const uint64_t* hashes; /* host pointer: sorted-unique k-mer hashes of the chunk */
  uint64_t n_hashes;
} kmcpg_build_col;
int kmcpg_build_db(const char* out_dir, const kmcpg_build_cfg* cfg, const kmcpg_build_col* cols, uint32_t n_cols, int32_t device);
/* The resident database written back as <out_dir>/R001/{_blockNNN.uniki, __db.yml, __name_mapping.tsv} in the reference's format
 * (index/serialization.go:159-300, util-db-info.go:46-79): bench.py puts its synthetic configs[1] index (planted reads included) on
 * /dev/shm this way and searches it with kmcp-search, FASTQ in, TSV out.  Every block must be resident on the handle. */
int kmcpg_save_db(kmcpg_db* db, const char* out_dir);

#ifdef __cplusplus
}
#endif
#endif
