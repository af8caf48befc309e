/*
 * kmcp_oracle.h — CPU restatement of the `kmcp search` hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X build.  It restates, in plain C, the algorithm of
 * shenwei356/kmcp v0.9.5 for the path named in BASELINE.json:north_star.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (kmcp_amd/, libkmcpgpu.so, kmcp-search) never links or calls anything in oracle/.
 *
 * Pinning status:
 *   - The reference is Go and cannot be built here (no Go toolchain): there is no oracle/_ref.
 *   - FracMinHash + plain ntHash path and the Closed-Syncmer path are pinned by the reference's
 *     two documentation goldens, demo-searching/README.md:61-68 and :102-109 (36 values,
 *     tests/test_oracle_golden.py).
 *   - Minimizer mode: PARITY UNPINNED (no golden in the reference).
 *   - Emission multiplicity of syncmers/minimizers for queries with <=256 k-mers: UNPINNED.
 *   - FPR column: Go's math.Pow / big.Float restated from the Go stdlib algorithm; compared with
 *     tolerance, not bit-exact (outside the north-star's bit-exact tuple).
 *
 * Third-party arithmetic that is NOT under /root/reference (restated from the published algorithms):
 *   github.com/will-rowe/nthash v0.4.0 (go.mod:43)   ntHash1 canonical rolling hash
 *   github.com/shenwei356/bio v0.9.0   (go.mod:13)   sketches.{HashIterator,MinimizerSketch,SyncmerSketch}
 *   github.com/shenwei356/pand v0.0.7, pospop v1.2.3, bmkessler/fastdiv (go.mod:16,17,27)
 */
#ifndef KMCP_ORACLE_H
#define KMCP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sketch configuration: the __db.yml fields that drive generateKmers (util-db-search.go:1037-1107) */
typedef struct {
  int32_t k;
  int32_t canonical;   /* index is always canonical (index.go:350-353) */
  int32_t scaled;      /* FracMinHash: keep code <= maxHash */
  uint32_t scale;
  int32_t minimizer;
  uint32_t minimizer_w;
  int32_t syncmer;
  uint32_t syncmer_s;
} ko_sketch_cfg;

/* ---- search options: SearchOptions (util-db-search.go:162-189), defaults search.go:1052-1102 */
typedef struct {
  int32_t min_qlen;        /* -m 30  */
  int32_t min_matched;     /* -c 10  */
  double min_qcov;         /* -t 0.55 */
  double min_tcov;         /* -T 0   */
  double max_fpr;          /* -f 0.01 */
  int32_t dedup_threshold; /* -u 256 */
  int32_t try_se;          /* --try-se */
  int32_t fpr_buf_size;    /* 249 SE / 499 PE (search.go:250-255) */
  int32_t sort_by;         /* 0 qcov, 1 tcov, 2 jacc */
  int32_t do_not_sort;     /* -S */
  int32_t top_n_scores;    /* -n */
} ko_search_params;

typedef struct {
  uint32_t block;      /* index of the .uniki file in __db.yml `files` order */
  uint32_t col;        /* column inside the block */
  uint32_t col_global; /* column numbered over all blocks in `files` order */
  uint32_t target_idx; /* chunkIdx | chunks<<16 (index.go:1096) */
  uint64_t gsize;
  uint64_t size;       /* #k-mers of the column */
  int32_t mkmers;
  double fpr, qcov, tcov, jacc;
  const char* target;  /* owned by the db */
} ko_match;

typedef struct {
  int32_t qlen;
  int32_t qkmers;   /* QueryResult.NumKmers */
  int32_t k;
  int32_t nmatches; /* -1: Matches == nil (unmatched) */
  ko_match* matches; /* malloc'd; free with ko_result_free */
} ko_result;

typedef struct ko_db ko_db;

/* ---- hashing / sketching ------------------------------------------------------------------ */
void ko_set_seed_mode(int iupac_rc_quirk); /* 1 (default): rc seed = seedTab[b&7] as ntHash v1 */
uint64_t ko_nthash_kmer(const uint8_t* kmer, int k, int canonical);
/* all hashes of seq by the rolling recurrence; returns count (len-k+1) or 0 if len<k */
size_t ko_nthash_all(const uint8_t* seq, size_t len, int k, int canonical, uint64_t* out);
uint64_t ko_max_hash(uint32_t scale);
/* generateKmers restatement; appends to out (capacity must be >= len); returns #appended */
size_t ko_generate_kmers(const uint8_t* seq, size_t len, const ko_sketch_cfg* cfg, uint64_t* out);
/* sort ascending + in-place unique as handleQuery :874-908; returns new length */
size_t ko_sort_unique(uint64_t* a, size_t n);
void ko_hash_values(uint64_t h, int num_hashes, uint64_t* out);

/* ---- index arithmetic ------------------------------------------------------------------------ */
uint64_t ko_calc_signature_size(uint64_t n_elements, int num_hashes, double fpr);
double ko_query_fpr(int n, int k, double p);
double ko_go_pow(double x, double y);
double ko_binomial_coeff(int n, int k);

/* ---- index writer (restates index.go block building; fixtures / synthetic DBs only) ----------- */
/* one column (= one reference chunk) of a block */
typedef struct {
  const char* name;
  uint64_t gsize;
  uint32_t chunk_idx;
  uint32_t chunks;
  const uint64_t* hashes; /* sorted-unique k-mer hashes of the chunk */
  uint64_t n_hashes;
} ko_column;

/* writes <path> as a .uniki block; num_sigs==0 => CalcSignatureSize(max n_hashes, h, fpr) */
int ko_write_block(const char* path, int k, int canonical, int num_hashes, double fpr,
                   uint64_t num_sigs, const ko_column* cols, uint32_t ncols);
/* builds <out_dir>/R001/{__db.yml,_blockNNN.uniki,__name_mapping.tsv} from columns, using the
 * block layout of index.go:657-682 (sort by #k-mers ascending, blocks of sBlock). */
int ko_build_db(const char* out_dir, const ko_sketch_cfg* cfg, int num_hashes, double fpr,
                int threads_for_block_size, int block_size, const ko_column* cols, uint32_t ncols);
/* the big-genome rules of `kmcp index` (index.go:787-894; flags -x/-X/-8/-1 at :1453-1463): 0 = the flag's default
 * (10M / 256 / 20M / 200M, "M" = 2^20 as bytesize.ParseByteSize reads it) */
typedef struct {
  uint64_t kmers_x; /* -x: above it blocks shrink to block_size_x columns */
  int32_t block_size_x; /* -X */
  uint64_t kmers_8; /* -8: above it blocks shrink to 8 columns */
  uint64_t kmers_1; /* -1: above it every column gets its own block */
} ko_block_rules;
/* kmers[i]: #k-mers of column i in ascending order.  block_of[i] = 1-based block of column i (0: empty, skipped).
 * sblock = the clamped -b value (index.go:671-682).  Returns the number of blocks. */
int ko_block_layout(const uint64_t* kmers, uint32_t n, int sblock, const ko_block_rules* rules, int* block_of);
int ko_build_db2(const char* out_dir, const ko_sketch_cfg* cfg, int num_hashes, double fpr, int threads_for_block_size,
                 int block_size, const ko_block_rules* rules, const ko_column* cols, uint32_t ncols);

/* ---- database + search ------------------------------------------------------------------------- */
ko_db* ko_db_open(const char* db_dir); /* db_dir is the R001-style dir holding __db.yml */
void ko_db_close(ko_db* db);
/* in-memory database over caller-owned row-major bit matrices (bench: a sample of the blocks of a synthetic
 * GTDB-scale index copied back from HBM).  Column c of block b is named "syn<col_base[b]+c>", Sizes = size_all. */
ko_db* ko_db_create_mem(const ko_sketch_cfg* cfg, int num_hashes, double fpr, int nblocks, const uint64_t* num_sigs,
                        const uint32_t* ncols, const uint32_t* col_base, const uint8_t* const* rows, uint64_t size_all);
const char* ko_last_error(void);
int ko_db_info(const ko_db* db, ko_sketch_cfg* cfg, int* num_hashes, double* fpr, int* nblocks,
               uint64_t* ncols_total);
/* per-block geometry */
int ko_db_block_info(const ko_db* db, int block, uint64_t* num_sigs, uint32_t* ncols, uint32_t* row_bytes);
const uint8_t* ko_db_block_rows(const ko_db* db, int block);
const char* ko_db_col_name(const ko_db* db, uint32_t col_global, uint32_t* target_idx, uint64_t* gsize,
                           uint64_t* size);

/* one query through handleQuery (util-db-search.go:763-1025) + handleQuerySingleDB sort (:260-345) */
int ko_search(ko_db* db, const uint8_t* seq1, size_t len1, const uint8_t* seq2, size_t len2,
              const ko_search_params* p, ko_result* out);
void ko_result_free(ko_result* r);

/* raw per-column counts of one block for a hash list (the a9-a11 inner kernel, :6804-6974) */
int ko_block_counts(const ko_db* db, int block, const uint64_t* kmers, size_t n, uint32_t* counts /*ncols*/);

/* batch search used by the bench's cpu_baseline leg: OpenMP over reads, word-wise vertical counters.
 * Writes for each read qkmers[i] and appends (read, col_global, count) triples that pass the integer
 * thresholds and the tcov/FPR filters.  Returns number of hits or <0. */
int64_t ko_search_batch(ko_db* db, const uint8_t* seqs, const uint64_t* offs, uint32_t n_reads,
                        const ko_search_params* p, int threads, int32_t* qkmers, uint32_t* hits_out,
                        int64_t hits_cap);

/* the same batch through the reference's own loop shape: one worker per block, 64 buffered row pointers, byte transposition +
 * Count8 per column byte (util-db-search.go:6811-6972, :213-219).  Single-hash databases. */
int64_t ko_search_batch_refshape(ko_db* db, const uint8_t* seqs, const uint64_t* offs, uint32_t n_reads,
                                 const ko_search_params* p, int threads, int32_t* qkmers, uint32_t* hits_out,
                                 int64_t hits_cap);

/* TSV line formatting exactly as search.go:517-575 (FormatFloat 'f',4 / 'e',4) */
int ko_format_match(char* buf, size_t cap, const char* query_id, const ko_result* r, const ko_match* m,
                    uint64_t query_idx);

#ifdef __cplusplus
}
#endif
#endif
