/*
 * kmcp_oracle.c — CPU restatement of the `kmcp search` hot path of shenwei356/kmcp v0.9.5.
 *
 * TEST INFRASTRUCTURE ONLY (see kmcp_oracle.h).  Every function cites the reference file:line it
 * follows; paths are relative to /root/reference/kmcp/cmd/ unless stated otherwise.
 * Compile with -ffp-contract=off: Go on amd64 never fuses a*b+c.
 */
#define _GNU_SOURCE
#include "kmcp_oracle.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[512];
const char* ko_last_error(void) { return g_err; }
#define FAIL(...)                                  \
  do {                                             \
    snprintf(g_err, sizeof g_err, __VA_ARGS__);    \
    return -1;                                     \
  } while (0)

/* ================================================================================================
 * ntHash1 — github.com/will-rowe/nthash v0.4.0 (go.mod:43), used through
 * bio/sketches.NewHashIterator / NextHash (call sites util-db-search.go:1057,1094; compute.go:754).
 * Seeds and the `& 0x07` complement trick are those of the original ntHash v1 tables.
 * ============================================================================================== */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL

static uint64_t g_fwd[256];
static uint64_t g_rev[256];
static int g_seed_ready = 0;
static int g_iupac_quirk = 1;

static void seed_init(void) {
  uint64_t tab[256];
  memset(tab, 0, sizeof tab);
  /* rows 0..7 of the ntHash v1 seed table: N,T,N,G,A,A,N,C — the complement of base b is tab[b&7] */
  tab[1] = SEED_T; tab[3] = SEED_G; tab[4] = SEED_A; tab[5] = SEED_A; tab[7] = SEED_C;
  tab['A'] = tab['a'] = SEED_A;
  tab['C'] = tab['c'] = SEED_C;
  tab['G'] = tab['g'] = SEED_G;
  tab['T'] = tab['t'] = SEED_T;
  tab['U'] = tab['u'] = SEED_T;
  for (int b = 0; b < 256; b++) {
    g_fwd[b] = tab[b];
    if (g_iupac_quirk) {
      g_rev[b] = tab[b & 7];
    } else { /* "everything that is not ACGTU hashes as 0 on both strands" variant */
      uint64_t f = tab[b];
      g_rev[b] = (b < 8) ? 0 : (f == SEED_A ? SEED_T : f == SEED_C ? SEED_G : f == SEED_G ? SEED_C : f == SEED_T ? SEED_A : 0);
    }
  }
  g_seed_ready = 1;
}
void ko_set_seed_mode(int q) { g_iupac_quirk = q; seed_init(); }

static inline uint64_t rol64(uint64_t v, unsigned s) { s &= 63; return s ? (v << s) | (v >> (64 - s)) : v; }
static inline uint64_t ror64(uint64_t v, unsigned s) { s &= 63; return s ? (v >> s) | (v << (64 - s)) : v; }

/* closed form: fh = XOR_j rol(seed[b_j], k-1-j); rh = XOR_j rol(seed[comp b_j], j) (ntf64/ntr64) */
uint64_t ko_nthash_kmer(const uint8_t* kmer, int k, int canonical) {
  if (!g_seed_ready) seed_init();
  uint64_t fh = 0, rh = 0;
  for (int j = 0; j < k; j++) {
    fh = rol64(fh, 1) ^ g_fwd[kmer[j]];
    rh = rol64(rh, 1) ^ g_rev[kmer[k - 1 - j]];
  }
  if (!canonical) return fh;
  return fh <= rh ? fh : rh;
}

/* rolling recurrence of NTHi.Next (alg. 3 of the ntHash paper):
 *   fh = rol(fh,1) ^ rol(seed[out],k) ^ seed[in]
 *   rh = ror(rh,1) ^ ror(seed[comp out],1) ^ rol(seed[comp in],k-1)                     */
size_t ko_nthash_all(const uint8_t* seq, size_t len, int k, int canonical, uint64_t* out) {
  if (!g_seed_ready) seed_init();
  if (k < 1 || len < (size_t)k) return 0;
  uint64_t fh = 0, rh = 0;
  for (int j = 0; j < k; j++) {
    fh = rol64(fh, 1) ^ g_fwd[seq[j]];
    rh = rol64(rh, 1) ^ g_rev[seq[k - 1 - j]];
  }
  size_t n = len - (size_t)k + 1;
  out[0] = canonical ? (fh <= rh ? fh : rh) : fh;
  for (size_t i = 1; i < n; i++) {
    uint8_t prev = seq[i - 1], end = seq[i + (size_t)k - 1];
    fh = rol64(fh, 1) ^ rol64(g_fwd[prev], (unsigned)k) ^ g_fwd[end];
    rh = ror64(rh, 1) ^ ror64(g_rev[prev], 1) ^ rol64(g_rev[end], (unsigned)(k - 1));
    out[i] = canonical ? (fh <= rh ? fh : rh) : fh;
  }
  return n;
}

/* util-db-search.go:1040-1043 / compute.go:316: maxHash = uint64(float64(^uint64(0)) / float64(scale)) */
uint64_t ko_max_hash(uint32_t scale) {
  double d = 18446744073709551615.0 /* float64(^uint64(0)) == 2^64 */ / (double)scale;
  if (d >= 18446744073709551616.0) return ~0ULL; /* Go's conversion of 2^64 is implementation-defined; scale==1 is never "scaled" */
  return (uint64_t)d;
}

/* ------------------------------------------------------------------------------------------------
 * sketches.NewMinimizerSketch / NextMinimizer (bio v0.9.0, call sites util-db-search.go:1055,1081).
 * Window of w consecutive canonical k-mer hashes, leftmost minimum; a minimizer is emitted when
 * its position differs from the previously emitted one.  PARITY UNPINNED (no golden).
 * ---------------------------------------------------------------------------------------------- */
static size_t minimizer_sketch(const uint8_t* seq, size_t len, int k, uint32_t w, uint64_t** out_codes) {
  *out_codes = NULL;
  if (w < 1 || len < (size_t)k + w - 1) return 0;
  size_t nk = len - (size_t)k + 1;
  uint64_t* hk = (uint64_t*)malloc(nk * sizeof(uint64_t));
  ko_nthash_all(seq, len, k, 1, hk);
  uint64_t* codes = (uint64_t*)malloc(nk * sizeof(uint64_t));
  size_t* dq = (size_t*)malloc(nk * sizeof(size_t));
  size_t head = 0, tail = 0, n = 0;
  size_t prev = (size_t)-1;
  for (size_t i = 0; i < nk; i++) {
    while (tail > head && hk[dq[tail - 1]] > hk[i]) tail--; /* strict: leftmost of equal values stays */
    dq[tail++] = i;
    if (i + 1 < w) continue;
    size_t w0 = i + 1 - w;
    while (dq[head] < w0) head++;
    size_t m = dq[head];
    if (m != prev) { codes[n++] = hk[m]; prev = m; }
  }
  free(dq); free(hk);
  *out_codes = codes;
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * sketches.NewSyncmerSketch / NextSyncmer (bio v0.9.0, call sites util-db-search.go:1053,1068).
 * Semantics verified against demo-searching/README.md:61-68:
 *   window of L = 2k-s-1 bases (2(k-s) s-mers); m = leftmost minimal canonical s-mer hash;
 *   emit the canonical hash of the k-mer STARTING at m if m-win_start < k-s, else of the k-mer
 *   ENDING at m+s; one emission per window.
 * ---------------------------------------------------------------------------------------------- */
static size_t syncmer_sketch(const uint8_t* seq, size_t len, int k, uint32_t s, uint64_t** out_codes) {
  *out_codes = NULL;
  if (s < 1 || (int)s > k) return 0;
  size_t L = 2 * (size_t)k - s - 1;
  if (len < L || len < (size_t)k) return 0;
  size_t nk = len - (size_t)k + 1, ns = len - s + 1;
  size_t nw = len - L + 1;
  size_t wsz = 2 * ((size_t)k - s); /* s-mers per window */
  uint64_t* hk = (uint64_t*)malloc(nk * sizeof(uint64_t));
  uint64_t* hs = (uint64_t*)malloc(ns * sizeof(uint64_t));
  ko_nthash_all(seq, len, k, 1, hk);
  ko_nthash_all(seq, len, (int)s, 1, hs);
  uint64_t* codes = (uint64_t*)malloc(nw * sizeof(uint64_t));
  size_t n = 0;
  if (wsz == 0) { /* s == k: every k-mer is its own window */
    for (size_t i = 0; i < nw && i < nk; i++) codes[n++] = hk[i];
  } else {
    size_t* dq = (size_t*)malloc(ns * sizeof(size_t));
    size_t head = 0, tail = 0;
    for (size_t i = 0; i < ns; i++) {
      while (tail > head && hs[dq[tail - 1]] > hs[i]) tail--;
      dq[tail++] = i;
      if (i + 1 < wsz) continue;
      size_t w0 = i + 1 - wsz;
      if (w0 >= nw) break;
      while (dq[head] < w0) head++;
      size_t m = dq[head];
      size_t pos = (m - w0 < (size_t)k - s) ? m : m + s - (size_t)k;
      codes[n++] = hk[pos];
    }
    free(dq);
  }
  free(hk); free(hs);
  *out_codes = codes;
  return n;
}

/* UnikIndexDB.generateKmers, util-db-search.go:1037-1107 (and the same loop in compute.go:746-801):
 * syncmer > minimizer > plain; keep code iff !(scaled && code > maxHash) && code > 0;
 * ErrShortSeq => no k-mers. */
size_t ko_generate_kmers(const uint8_t* seq, size_t len, const ko_sketch_cfg* cfg, uint64_t* out) {
  if (!g_seed_ready) seed_init();
  int k = cfg->k;
  uint64_t max_hash = ~0ULL;
  if (cfg->scaled) max_hash = ko_max_hash(cfg->scale);
  uint64_t* codes = NULL;
  size_t nc = 0;
  int own = 1;
  if (cfg->syncmer) {
    nc = syncmer_sketch(seq, len, k, cfg->syncmer_s, &codes);
  } else if (cfg->minimizer) {
    nc = minimizer_sketch(seq, len, k, cfg->minimizer_w, &codes);
  } else {
    if (k < 1 || len < (size_t)k) return 0;
    codes = (uint64_t*)malloc((len - (size_t)k + 1) * sizeof(uint64_t));
    nc = ko_nthash_all(seq, len, k, cfg->canonical, codes);
  }
  size_t n = 0;
  for (size_t i = 0; i < nc; i++) {
    uint64_t code = codes[i];
    if (cfg->scaled && code > max_hash) continue;
    if (code > 0) out[n++] = code;
  }
  if (own) free(codes);
  return n;
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : x > y;
}
/* util-db-search.go:874-908: sort.Sort + in-place unique */
size_t ko_sort_unique(uint64_t* a, size_t n) {
  if (n == 0) return 0;
  qsort(a, n, sizeof(uint64_t), cmp_u64);
  size_t j = 1;
  for (size_t i = 1; i < n; i++)
    if (a[i] != a[j - 1]) a[j++] = a[i];
  return j;
}

/* util-hash.go:125-142 hashValues: a = hi32, b = lo32, h_i = uint64(uint32(a + b*i)) */
void ko_hash_values(uint64_t h, int num_hashes, uint64_t* out) {
  if (num_hashes == 1) { out[0] = h; return; }
  uint32_t a = (uint32_t)(h >> 32), b = (uint32_t)h;
  for (uint32_t i = 0; i < (uint32_t)num_hashes; i++) out[i] = (uint64_t)(uint32_t)(a + b * i);
}

/* util-hash.go:46-50 CalcSignatureSize */
uint64_t ko_calc_signature_size(uint64_t n_elements, int num_hashes, double fpr) {
  double ratio = (double)(-num_hashes) / log(1.0 - ko_go_pow(fpr, 1.0 / (double)num_hashes));
  return (uint64_t)ceil((double)n_elements * ratio);
}

/* ================================================================================================
 * Query FPR — util-fpr.go:32-71 (QueryFPR, BinomialCoeff).
 * The search path calls the CACHED variant (util-fpr.go:140-191, built at util-db-search.go:683 with bufSize 249 / 499):
 * for n <= bufSize it stores FPR(n, k) in slot n*h + min(k, n-k), so (n, k) and (n, n-k) share a slot and the Go binary
 * returns whichever of the two was computed first.  For k > n-k for every passing count - i.e. `-t >= 0.5`, the default 0.55
 * included - the slot is unique per k and cached == uncached.  With `-t < 0.5` and n <= 249 (499 paired) the reference's FPR
 * column depends on arrival order; this restatement (and the product) always return the uncached FPR(n, k).  The cache is
 * restated slot for slot in tests/test_fpr_golden.py (GoFprCache), which shows both outputs for one (n, k).
 * ============================================================================================== */
/* Go math.Pow (src/math/pow.go, pure Go on amd64), restated. */
double ko_go_pow(double x, double y) {
  if (y == 0 || x == 1) return 1;
  if (y == 1) return x;
  if (isnan(x) || isnan(y)) return NAN;
  if (x == 0) {
    if (y < 0) return INFINITY;
    return 0;
  }
  if (isinf(y)) {
    if (x == -1) return 1;
    if ((fabs(x) < 1) == (y > 0)) return 0;
    return INFINITY;
  }
  if (isinf(x)) return y < 0 ? 0 : INFINITY;
  if (y == 0.5) return sqrt(x);
  if (y == -0.5) return 1 / sqrt(x);
  double yi, yf;
  yf = modf(fabs(y), &yi);
  if (yf != 0 && x < 0) return NAN;
  if (yi >= 9.223372036854775808e18) {
    if (x == -1) return 1;
    if ((fabs(x) < 1) == (y > 0)) return 0;
    return INFINITY;
  }
  double a1 = 1.0;
  int ae = 0;
  if (yf != 0) {
    if (yf > 0.5) { yf -= 1; yi += 1; }
    a1 = exp(yf * log(x));
  }
  int xe;
  double x1 = frexp(x, &xe);
  for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
    if (xe < -(1 << 12) || (1 << 12) < xe) {
      ae += xe;
      break;
    }
    if (i & 1) { a1 *= x1; ae += xe; }
    x1 *= x1;
    xe <<= 1;
    if (x1 < .5) { x1 += x1; xe--; }
  }
  if (y < 0) { a1 = 1 / a1; ae = -ae; }
  return ldexp(a1, ae);
}

/* util-fpr.go:54-71: product in big.Float (53-bit mantissa, unbounded exponent).  A normalised
 * (mantissa, exponent) pair of doubles gives the identical roundings. */
double ko_binomial_coeff(int n, int k) {
  if (k > n - k) k = n - k;
  double m = 0.5; /* value = m * 2^e = 1 */
  int e = 1;
  for (int i = 0; i < k; i++) {
    int de;
    m = frexp(m * (double)(n - i), &de); e += de;
    m = frexp(m / (double)(i + 1), &de); e += de;
  }
  return ldexp(m, e); /* +Inf when > MaxFloat64, as big.Float.Float64() */
}

/* util-fpr.go:32-50 QueryFPR (Theorem 2 of doi:10.1038/nbt.3442) */
double ko_query_fpr(int n, int k, double p) {
  double r = 1;
  for (int i = 0; i <= k; i++) {
    double coeff = ko_binomial_coeff(n, i);
    if (coeff > 1.79769313486231570814527423731704356798070e+308) return 0;
    double t = coeff * ko_go_pow(p, (double)i);
    t = t * ko_go_pow(1 - p, (double)(n - i));
    r -= t;
    if (r < 0) return 0;
  }
  return r;
}

/* ================================================================================================
 * .uniki block files — index/serialization.go:159-300 (writer), :383-593 (reader)
 * ============================================================================================== */
typedef struct {
  int k, canonical, compact, num_hashes;
  uint64_t num_sigs;
  uint32_t ncols, row_bytes;
  char** names;       /* first name of each name group (single-set DBs: exactly one, index.go:622-626) */
  uint64_t* gsizes;
  uint32_t* indices;
  uint64_t* sizes;
  uint8_t* file;      /* whole file */
  size_t file_len;
  size_t offset0;     /* byte offset of row 0 (util-db-search.go:1207) */
  uint32_t col_base;  /* global column number of column 0 */
  int borrowed;       /* rows not owned (ko_db_create_mem) */
} ko_block;

struct ko_db {
  ko_sketch_cfg cfg;
  int ks[16];   /* k-mer sizes of the database, descending (util-db-search.go:752-758); cfg.k = ks[0] */
  int nks;
  int num_hashes;
  double fpr;
  int nblocks;
  ko_block* blocks;
  uint64_t ncols_total;
};

static void put_be32(FILE* f, uint32_t v) { uint8_t b[4] = {v >> 24, v >> 16, v >> 8, v}; fwrite(b, 1, 4, f); }
static void put_be64(FILE* f, uint64_t v) { put_be32(f, (uint32_t)(v >> 32)); put_be32(f, (uint32_t)v); }
static uint32_t get_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint64_t get_be64(const uint8_t* p) { return ((uint64_t)get_be32(p) << 32) | get_be32(p + 4); }

static int cmp_col_by_kmers(const void* a, const void* b) {
  const ko_column* x = *(const ko_column* const*)a; const ko_column* y = *(const ko_column* const*)b;
  if (x->n_hashes != y->n_hashes) return x->n_hashes < y->n_hashes ? -1 : 1;
  return x < y ? -1 : x > y; /* stable tie-break (the reference's parallel quicksort is unstable) */
}

/* index.go:1023 (numSigs), :1107-1309 (Bloom columns, bit 7-_k of byte col/8), serialization.go:159-300 */
int ko_write_block(const char* path, int k, int canonical, int num_hashes, double fpr, uint64_t num_sigs,
                   const ko_column* cols, uint32_t ncols) {
  if (ncols == 0) FAIL("no columns");
  uint64_t max_elems = 0;
  for (uint32_t c = 0; c < ncols; c++) if (cols[c].n_hashes > max_elems) max_elems = cols[c].n_hashes;
  if (num_sigs == 0) num_sigs = ko_calc_signature_size(max_elems, num_hashes, fpr);
  uint32_t row_bytes = (ncols + 7) / 8;
  uint8_t* sigs = (uint8_t*)calloc((size_t)num_sigs * row_bytes, 1);
  if (!sigs) FAIL("oom for %llu x %u matrix", (unsigned long long)num_sigs, row_bytes);
  uint64_t hv[8];
  for (uint32_t c = 0; c < ncols; c++) {
    uint8_t bit = (uint8_t)(1u << (7 - (c & 7)));
    for (uint64_t i = 0; i < cols[c].n_hashes; i++) {
      ko_hash_values(cols[c].hashes[i], num_hashes, hv);
      for (int h = 0; h < num_hashes; h++) sigs[(size_t)(hv[h] % num_sigs) * row_bytes + (c >> 3)] |= bit;
    }
  }
  FILE* f = fopen(path, "wb");
  if (!f) { free(sigs); FAIL("cannot write %s: %s", path, strerror(errno)); }
  fwrite(".kmcpidx", 1, 8, f);
  uint8_t meta[4] = {4, (uint8_t)k, (uint8_t)((canonical ? 1 : 0) | 2 /* COMPACT = !faster */), (uint8_t)num_hashes};
  fwrite(meta, 1, 4, f);
  put_be64(f, num_sigs);
  put_be32(f, ncols);
  for (uint32_t c = 0; c < ncols; c++) {
    put_be32(f, (uint32_t)strlen(cols[c].name) + 1);
    fwrite(cols[c].name, 1, strlen(cols[c].name), f);
    fputc('\n', f);
  }
  put_be32(f, ncols);
  for (uint32_t c = 0; c < ncols; c++) { put_be32(f, 1); put_be64(f, cols[c].gsize); }
  put_be32(f, ncols);
  for (uint32_t c = 0; c < ncols; c++) { put_be32(f, 1); put_be32(f, cols[c].chunk_idx + (cols[c].chunks << 16)); }
  for (uint32_t c = 0; c < ncols; c++) put_be64(f, cols[c].n_hashes);
  fwrite(sigs, 1, (size_t)num_sigs * row_bytes, f);
  fclose(f);
  free(sigs);
  return 0;
}

static int mkdir_p(const char* d) {
  char tmp[1024];
  snprintf(tmp, sizeof tmp, "%s", d);
  for (char* p = tmp + 1; *p; p++)
    if (*p == '/') { *p = 0; mkdir(tmp, 0755); *p = '/'; }
  if (mkdir(tmp, 0755) != 0 && errno != EEXIST) return -1;
  return 0;
}

/* index.go:787-894: the loop that cuts the ascending list of columns into blocks.  It runs one step past the end
 * (i == n) to flush; `last` is the column held back when a size threshold is crossed ("leave this file process in
 * the next round"), so that the columns already batched are written as a (possibly short) block first. */
int ko_block_layout(const uint64_t* kmers, uint32_t n, int sblock, const ko_block_rules* rules, int* block_of) {
  uint64_t thr_x = rules && rules->kmers_x ? rules->kmers_x : 10ull << 20;
  uint64_t thr_8 = rules && rules->kmers_8 ? rules->kmers_8 : 20ull << 20;
  uint64_t thr_1 = rules && rules->kmers_1 ? rules->kmers_1 : 200ull << 20;
  int size_x = rules && rules->block_size_x ? rules->block_size_x : 256;
  int skip_x = 0;
  if (size_x >= sblock) { skip_x = 1; size_x = sblock; } /* index.go:684-689 */
  int flag = 0, flag8 = 0, flagx = 0, nb = 0;
  long last = -1;
  int nbatch = 0;               /* columns in the open batch; they are the nbatch most recent non-held-back ones */
  uint32_t* batch = (uint32_t*)malloc(((size_t)n + 1) * sizeof(uint32_t));
  for (uint32_t i = 0; i < n; i++) block_of[i] = 0;
  for (uint32_t i = 0; i <= n; i++) {
    if (i == n) {
      if ((flag || flag8 || flagx) && last >= 0) { batch[nbatch++] = (uint32_t)last; last = -1; }
    } else {
      uint64_t km = kmers[i];
      if (km == 0) continue; /* :799-801 */
      if (flag || flag8 || flagx) {
        if (last >= 0) { batch[nbatch++] = (uint32_t)last; last = -1; }
        if (flag) last = i;                                     /* :810-812 one column per block from now on */
        else if (km > thr_1) { flag = 1; last = i; }            /* :813-817 */
        else if (skip_x) { batch[nbatch++] = i; if (nbatch < sblock) continue; } /* :818-822 */
        else if (km > thr_8) {
          if (flag8) { batch[nbatch++] = i; if (nbatch < sblock) continue; }     /* :825-829 */
          else { sblock = 8; flag8 = 1; last = i; }             /* :830-836 */
        } else { batch[nbatch++] = i; if (nbatch < sblock) continue; }           /* :837-842 */
      } else if (skip_x) {
        if (km > thr_8) {                                       /* :846-856 */
          if (km > thr_1) flag = 1;
          else { sblock = size_x; flagx = 1; }
          last = i;
        } else { batch[nbatch++] = i; if (nbatch < sblock) continue; }
      } else {
        if (km > thr_x) {                                       /* :864-878 */
          if (km > thr_1) flag = 1;
          else if (km > thr_8) { sblock = 8; flag8 = 1; }
          else { sblock = size_x; flagx = 1; }
          last = i;
        } else { batch[nbatch++] = i; if (nbatch < sblock) continue; }
      }
    }
    if (nbatch == 0) {
      if (last < 0) break; /* :887-893 */
      continue;
    }
    nb++;
    for (int j = 0; j < nbatch; j++) block_of[batch[j]] = nb;
    nbatch = 0;
  }
  free(batch);
  return nb;
}

/* index.go:657-682 block sizing; :787-894 batching; :1283-1285 file names; :1352-1373 __db.yml;
 * util-db-info.go:46-79 */
int ko_build_db(const char* out_dir, const ko_sketch_cfg* cfg, int num_hashes, double fpr, int threads,
                int block_size, const ko_column* cols, uint32_t ncols) {
  return ko_build_db2(out_dir, cfg, num_hashes, fpr, threads, block_size, NULL, cols, ncols);
}

int ko_build_db2(const char* out_dir, const ko_sketch_cfg* cfg, int num_hashes, double fpr, int threads,
                 int block_size, const ko_block_rules* rules, const ko_column* cols, uint32_t ncols) {
  if (ncols == 0) FAIL("no columns");
  const ko_column** order = (const ko_column**)malloc(ncols * sizeof(*order));
  uint64_t total = 0;
  for (uint32_t i = 0; i < ncols; i++) {
    order[i] = &cols[i];
    total += cols[i].n_hashes;
  }
  qsort(order, ncols, sizeof(*order), cmp_col_by_kmers);
  int sblock;
  if (block_size <= 0) {
    if (threads < 1) threads = 1;
    sblock = ((int)((double)ncols / (double)threads) + 7) / 8 * 8;
  } else sblock = block_size;
  if (sblock > (int)ncols) sblock = (int)ncols;
  if (sblock < 8) sblock = 8;
  char dir[900], path[1024];
  snprintf(dir, sizeof dir, "%s/R001", out_dir);
  if (mkdir_p(dir) != 0) { free(order); FAIL("cannot create %s", dir); }
  uint64_t* km = (uint64_t*)malloc(ncols * sizeof(uint64_t));
  int* block_of = (int*)malloc(ncols * sizeof(int));
  for (uint32_t i = 0; i < ncols; i++) km[i] = order[i]->n_hashes;
  int nb = ko_block_layout(km, ncols, sblock, rules, block_of);
  free(km);
  ko_column* batch = (ko_column*)malloc((size_t)ncols * sizeof(ko_column));
  for (int b = 1; b <= nb; b++) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < ncols; i++)
      if (block_of[i] == b) batch[n++] = *order[i];
    snprintf(path, sizeof path, "%s/_block%03d.uniki", dir, b);
    if (ko_write_block(path, cfg->k, cfg->canonical, num_hashes, fpr, 0, batch, n) != 0) { free(batch); free(block_of); free(order); return -1; }
  }
  free(batch);
  free(block_of);
  snprintf(path, sizeof path, "%s/__db.yml", dir);
  FILE* f = fopen(path, "w");
  if (!f) { free(order); FAIL("cannot write %s", path); }
  fprintf(f, "version: 4\nunikiVersion: 4\nalias: %s\nk: %d\nks:\n- %d\nhashed: true\ncanonical: %s\n", "oracle-db", cfg->k, cfg->k,
          cfg->canonical ? "true" : "false");
  fprintf(f, "scaled: %s\nscale: %u\nminimizer: %s\nminimizer-w: %u\nsyncmer: %s\nsyncmer-s: %u\n", cfg->scaled ? "true" : "false",
          cfg->scale, cfg->minimizer ? "true" : "false", cfg->minimizer_w, cfg->syncmer ? "true" : "false", cfg->syncmer_s);
  fprintf(f, "split-seq: false\nsplit-size: 0\nsplit-num: 1\nsplit-overlap: 0\ncompact-size: true\n");
  fprintf(f, "hashes: %d\nfpr: %.17g\nnumNameGroups: %u\nblocksize: %d\ntotalKmers: %llu\nfiles:\n", num_hashes, fpr, ncols, sblock,
          (unsigned long long)total);
  for (int b = 1; b <= nb; b++) fprintf(f, "- _block%03d.uniki\n", b);
  fclose(f);
  snprintf(path, sizeof path, "%s/__name_mapping.tsv", dir);
  f = fopen(path, "w");
  if (f) {
    for (uint32_t i = 0; i < ncols; i++) fprintf(f, "%s\t%s\n", cols[i].name, cols[i].name);
    fclose(f);
  }
  free(order);
  return 0;
}

static char* trim(char* s) {
  while (*s == ' ' || *s == '\t') s++;
  size_t n = strlen(s);
  while (n && (s[n - 1] == '\n' || s[n - 1] == '\r' || s[n - 1] == ' ' || s[n - 1] == '\t')) s[--n] = 0;
  if (n >= 2 && ((s[0] == '"' && s[n - 1] == '"') || (s[0] == '\'' && s[n - 1] == '\''))) { s[n - 1] = 0; s++; }
  return s;
}
static int yaml_bool(const char* v) { return !strcmp(v, "true") || !strcmp(v, "True") || !strcmp(v, "yes"); }

/* serialization.go:383-593 readHeader */
static int read_block(const char* path, ko_block* b) {
  FILE* f = fopen(path, "rb");
  if (!f) FAIL("cannot open %s: %s", path, strerror(errno));
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* d = (uint8_t*)malloc((size_t)sz);
  if (!d || fread(d, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(d); FAIL("cannot read %s", path); }
  fclose(f);
  if (sz < 24 || memcmp(d, ".kmcpidx", 8) != 0) { free(d); FAIL("kmcp: invalid index format: %s", path); }
  if (d[8] != 4) { free(d); FAIL("kmcp: version mismatch: %s", path); }
  memset(b, 0, sizeof *b);
  b->k = d[9];
  b->canonical = (d[10] & 1) != 0;
  b->compact = (d[10] & 2) != 0;
  b->num_hashes = d[11];
  b->num_sigs = get_be64(d + 12);
  size_t p = 20;
  uint32_t n = get_be32(d + p); p += 4;
  b->ncols = n;
  b->names = (char**)calloc(n, sizeof(char*));
  b->gsizes = (uint64_t*)calloc(n, 8);
  b->indices = (uint32_t*)calloc(n, 4);
  b->sizes = (uint64_t*)calloc(n, 8);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t l = get_be32(d + p); p += 4;
    const char* s = (const char*)d + p;
    size_t first = 0;
    while (first < l && s[first] != '\n') first++;
    b->names[i] = strndup(s, first); /* Target[0] */
    p += l;
  }
  uint32_t ng = get_be32(d + p); p += 4;
  for (uint32_t i = 0; i < ng; i++) {
    uint32_t m = get_be32(d + p); p += 4;
    if (m && i < n) b->gsizes[i] = get_be64(d + p);
    p += (size_t)m * 8;
  }
  uint32_t ni = get_be32(d + p); p += 4;
  for (uint32_t i = 0; i < ni; i++) {
    uint32_t m = get_be32(d + p); p += 4;
    if (m && i < n) b->indices[i] = get_be32(d + p);
    p += (size_t)m * 4;
  }
  for (uint32_t i = 0; i < n; i++) { b->sizes[i] = get_be64(d + p); p += 8; }
  b->row_bytes = (n + 7) / 8;
  b->offset0 = p;
  b->file = d;
  b->file_len = (size_t)sz;
  if (p + (size_t)b->num_sigs * b->row_bytes > (size_t)sz) FAIL("kmcp: truncated index file: %s", path);
  return 0;
}

/* util-db-info.go:98-129 UnikIndexDBInfoFromFile + NewUnikIndexDB (util-db-search.go:648-743) */
ko_db* ko_db_open(const char* db_dir) {
  if (!g_seed_ready) seed_init();
  char path[1024];
  snprintf(path, sizeof path, "%s/__db.yml", db_dir);
  FILE* f = fopen(path, "r");
  if (!f) { snprintf(g_err, sizeof g_err, "fail to open kmcp database info file: %s", path); return NULL; }
  ko_db* db = (ko_db*)calloc(1, sizeof *db);
  db->cfg.canonical = 1;
  db->cfg.scale = 1;
  char line[4096];
  char** files = NULL;
  int nfiles = 0, in_files = 0, in_ks = 0, version = -1;
  int ks_max = -1;
  while (fgets(line, sizeof line, f)) {
    char* s = trim(line);
    if (!*s || *s == '#') continue;
    if (s[0] == '-') {
      char* v = trim(s + 1);
      if (in_files) { files = (char**)realloc(files, (size_t)(nfiles + 1) * sizeof(char*)); files[nfiles++] = strdup(v); }
      else if (in_ks) { int kk = atoi(v); if (kk > ks_max) ks_max = kk; if (db->nks < 16) db->ks[db->nks++] = kk; }
      continue;
    }
    char* colon = strchr(s, ':');
    if (!colon) continue;
    *colon = 0;
    char* key = trim(s);
    char* val = trim(colon + 1);
    in_files = !strcmp(key, "files");
    in_ks = !strcmp(key, "ks");
    if (in_ks && *val == '[') { /* flow style: ks: [21] */
      for (char* q = val + 1; *q; q++) if (*q >= '0' && *q <= '9') { int kk = atoi(q); if (kk > ks_max) ks_max = kk; if (db->nks < 16) db->ks[db->nks++] = kk; while (*q >= '0' && *q <= '9') q++; if (!*q) break; }
    }
    if (!strcmp(key, "version")) version = atoi(val);
    else if (!strcmp(key, "k")) db->cfg.k = atoi(val);
    else if (!strcmp(key, "canonical")) db->cfg.canonical = yaml_bool(val);
    else if (!strcmp(key, "scaled")) db->cfg.scaled = yaml_bool(val);
    else if (!strcmp(key, "scale")) db->cfg.scale = (uint32_t)strtoul(val, NULL, 10);
    else if (!strcmp(key, "minimizer")) db->cfg.minimizer = yaml_bool(val);
    else if (!strcmp(key, "minimizer-w")) db->cfg.minimizer_w = (uint32_t)strtoul(val, NULL, 10);
    else if (!strcmp(key, "syncmer")) db->cfg.syncmer = yaml_bool(val);
    else if (!strcmp(key, "syncmer-s")) db->cfg.syncmer_s = (uint32_t)strtoul(val, NULL, 10);
    else if (!strcmp(key, "hashes")) db->num_hashes = atoi(val);
    else if (!strcmp(key, "fpr")) db->fpr = strtod(val, NULL);
  }
  fclose(f);
  if (ks_max > 0) db->cfg.k = ks_max; /* handleQuery walks ks descending (:752-758, :764) */
  if (db->nks == 0) db->ks[db->nks++] = db->cfg.k; /* util-db-info.go:124-126 */
  for (int i = 0; i < db->nks; i++) for (int j = i + 1; j < db->nks; j++) if (db->ks[j] > db->ks[i]) { int t = db->ks[i]; db->ks[i] = db->ks[j]; db->ks[j] = t; }
  if (version != 4) { snprintf(g_err, sizeof g_err, "kmcp/index: version mismatch"); free(db); return NULL; }
  if (nfiles == 0) { snprintf(g_err, sizeof g_err, "no index files"); free(db); return NULL; }
  db->nblocks = nfiles;
  db->blocks = (ko_block*)calloc((size_t)nfiles, sizeof(ko_block));
  uint32_t base = 0;
  for (int i = 0; i < nfiles; i++) {
    snprintf(path, sizeof path, "%s/%s", db_dir, files[i]);
    if (read_block(path, &db->blocks[i]) != 0) { ko_db_close(db); return NULL; }
    db->blocks[i].col_base = base;
    base += db->blocks[i].ncols;
    if (db->blocks[i].k != db->cfg.k || db->blocks[i].num_hashes != db->num_hashes || db->blocks[i].canonical != db->cfg.canonical) {
      snprintf(g_err, sizeof g_err, "index files not compatible"); ko_db_close(db); return NULL;
    }
    free(files[i]);
  }
  free(files);
  db->ncols_total = base;
  return db;
}

ko_db* ko_db_create_mem(const ko_sketch_cfg* cfg, int num_hashes, double fpr, int nblocks, const uint64_t* num_sigs,
                        const uint32_t* ncols, const uint32_t* col_base, const uint8_t* const* rows, uint64_t size_all) {
  if (!g_seed_ready) seed_init();
  ko_db* db = (ko_db*)calloc(1, sizeof *db);
  db->cfg = *cfg;
  db->num_hashes = num_hashes;
  db->fpr = fpr;
  db->nblocks = nblocks;
  db->ks[0] = cfg->k;
  db->nks = 1;
  db->blocks = (ko_block*)calloc((size_t)nblocks, sizeof(ko_block));
  for (int i = 0; i < nblocks; i++) {
    ko_block* b = &db->blocks[i];
    b->k = cfg->k; b->canonical = cfg->canonical; b->compact = 1; b->num_hashes = num_hashes;
    b->num_sigs = num_sigs[i]; b->ncols = ncols[i]; b->row_bytes = (ncols[i] + 7) / 8;
    b->names = (char**)calloc(ncols[i], sizeof(char*));
    b->gsizes = (uint64_t*)calloc(ncols[i], 8);
    b->indices = (uint32_t*)calloc(ncols[i], 4);
    b->sizes = (uint64_t*)calloc(ncols[i], 8);
    for (uint32_t c = 0; c < ncols[i]; c++) {
      char nm[32];
      snprintf(nm, sizeof nm, "syn%u", col_base[i] + c);
      b->names[c] = strdup(nm);
      b->gsizes[c] = 4000000; b->indices[c] = (c % 10) | (10u << 16); b->sizes[c] = size_all;
    }
    b->file = (uint8_t*)rows[i]; b->offset0 = 0; b->borrowed = 1; b->col_base = col_base[i];
    if ((uint64_t)col_base[i] + ncols[i] > db->ncols_total) db->ncols_total = (uint64_t)col_base[i] + ncols[i];
  }
  return db;
}

void ko_db_close(ko_db* db) {
  if (!db) return;
  for (int i = 0; i < db->nblocks; i++) {
    ko_block* b = &db->blocks[i];
    if (b->names) for (uint32_t c = 0; c < b->ncols; c++) free(b->names[c]);
    free(b->names); free(b->gsizes); free(b->indices); free(b->sizes);
    if (!b->borrowed) free(b->file);
  }
  free(db->blocks);
  free(db);
}

int ko_db_info(const ko_db* db, ko_sketch_cfg* cfg, int* num_hashes, double* fpr, int* nblocks, uint64_t* ncols_total) {
  if (cfg) *cfg = db->cfg;
  if (num_hashes) *num_hashes = db->num_hashes;
  if (fpr) *fpr = db->fpr;
  if (nblocks) *nblocks = db->nblocks;
  if (ncols_total) *ncols_total = db->ncols_total;
  return 0;
}
int ko_db_block_info(const ko_db* db, int block, uint64_t* num_sigs, uint32_t* ncols, uint32_t* row_bytes) {
  if (block < 0 || block >= db->nblocks) FAIL("bad block");
  if (num_sigs) *num_sigs = db->blocks[block].num_sigs;
  if (ncols) *ncols = db->blocks[block].ncols;
  if (row_bytes) *row_bytes = db->blocks[block].row_bytes;
  return 0;
}
const uint8_t* ko_db_block_rows(const ko_db* db, int block) { return db->blocks[block].file + db->blocks[block].offset0; }
const char* ko_db_col_name(const ko_db* db, uint32_t cg, uint32_t* target_idx, uint64_t* gsize, uint64_t* size) {
  for (int i = 0; i < db->nblocks; i++) {
    const ko_block* b = &db->blocks[i];
    if (cg >= b->col_base && cg < b->col_base + b->ncols) {
      uint32_t c = cg - b->col_base;
      if (target_idx) *target_idx = b->indices[c];
      if (gsize) *gsize = b->gsizes[c];
      if (size) *size = b->sizes[c];
      return b->names[c];
    }
  }
  return NULL;
}

/* ================================================================================================
 * The COBS query of one block — UnikIndex worker, util-db-search.go:6611-6616 (reset), :6622-6803
 * (multi-hash: AND of h rows, pand), :6804-6974 (single hash), transposed positional popcount
 * (pospop.Count8; bit 7 of byte i = column 8i, :7466-7704).  Written as the plain definition:
 * counts[col] = #k-mers whose h rows all have the column bit set.
 * ============================================================================================== */
int ko_block_counts(const ko_db* db, int block, const uint64_t* kmers, size_t n, uint32_t* counts) {
  const ko_block* b = &db->blocks[block];
  const uint8_t* sigs = b->file + b->offset0;
  uint32_t rb = b->row_bytes;
  memset(counts, 0, (size_t)b->ncols * sizeof(uint32_t));
  uint8_t* acc = (uint8_t*)malloc(rb);
  uint64_t hv[8];
  for (size_t i = 0; i < n; i++) {
    ko_hash_values(kmers[i], b->num_hashes, hv);
    const uint8_t* row = sigs + (size_t)(hv[0] % b->num_sigs) * rb; /* div.Mod(_h): exact modulo */
    memcpy(acc, row, rb);
    for (int h = 1; h < b->num_hashes; h++) {
      row = sigs + (size_t)(hv[h] % b->num_sigs) * rb;
      for (uint32_t j = 0; j < rb; j++) acc[j] &= row[j];
    }
    for (uint32_t j = 0; j < rb; j++) {
      uint8_t v = acc[j];
      while (v) {
        int bit = 31 - __builtin_clz((unsigned)v);
        uint32_t col = 8 * j + (7 - (uint32_t)bit);
        if (col < b->ncols) counts[col]++;
        v &= (uint8_t)~(1u << bit);
      }
    }
  }
  free(acc);
  return 0;
}

/* threshold + Match, util-db-search.go:7415-7733 */
static void block_matches(const ko_db* db, int block, const uint32_t* counts, int n_hashes, const ko_search_params* p,
                          ko_match** out, int* nout, int* cap) {
  const ko_block* b = &db->blocks[block];
  double nh = (double)n_hashes;
  double thr = nh * p->min_qcov;
  for (uint32_t c = 0; c < b->ncols; c++) {
    int count = (int)counts[c];
    if (count < p->min_matched) continue;
    double cf = (double)count;
    if (!(cf > thr)) continue;
    double t = cf / nh;
    double nt = (double)b->sizes[c];
    double T = cf / nt;
    if (!(T >= p->min_tcov)) continue;
    double fpr = ko_query_fpr(n_hashes, count, db->fpr);
    if (!(fpr <= p->max_fpr)) continue;
    if (*nout == *cap) { *cap = *cap ? *cap * 2 : 16; *out = (ko_match*)realloc(*out, (size_t)*cap * sizeof(ko_match)); }
    ko_match* m = &(*out)[(*nout)++];
    m->block = (uint32_t)block; m->col = c; m->col_global = b->col_base + c;
    m->target = b->names[c]; m->target_idx = b->indices[c]; m->gsize = b->gsizes[c]; m->size = b->sizes[c];
    m->mkmers = count; m->fpr = fpr; m->qcov = t; m->tcov = T; m->jacc = cf / (nh + nt - cf);
  }
}

/* Matches.Less / SortByTCov.Less / SortByJacc.Less, util-db-search.go:105-145.  The reference sorts with an
 * unstable parallel quicksort over an arrival-order list; the oracle uses (block, col) order + stable sort. */
static int g_sort_by;
static int cmp_match(const void* a, const void* b) {
  const ko_match* x = (const ko_match*)a; const ko_match* y = (const ko_match*)b;
  double s1, s2, t1, t2;
  switch (g_sort_by) {
    case 1: s1 = x->tcov; s2 = y->tcov; t1 = x->mkmers; t2 = y->mkmers; break;
    case 2: s1 = x->jacc; s2 = y->jacc; t1 = x->mkmers; t2 = y->mkmers; break;
    default: s1 = x->qcov; s2 = y->qcov; t1 = x->tcov; t2 = y->tcov; break;
  }
  if (s1 > s2) return -1;
  if (s1 < s2) return 1;
  if (t1 > t2) return -1;
  if (t1 < t2) return 1;
  if (x->col_global != y->col_global) return x->col_global < y->col_global ? -1 : 1;
  return 0;
}

static int search_kmers(ko_db* db, uint64_t* kmers, size_t nk, const ko_search_params* p, ko_result* out) {
  /* :854-869 */
  if ((int)nk < p->min_matched) return 0;
  size_t n = nk;
  out->qkmers = (int32_t)n;
  /* :874-908 */
  if ((int)n > p->dedup_threshold) n = ko_sort_unique(kmers, n);
  out->qkmers = (int32_t)n; /* :910 */
  ko_match* ms = NULL;
  int nm = 0, cap = 0;
  uint32_t maxc = 0;
  for (int b = 0; b < db->nblocks; b++) if (db->blocks[b].ncols > maxc) maxc = db->blocks[b].ncols;
  uint32_t* counts = (uint32_t*)malloc((size_t)maxc * sizeof(uint32_t));
  for (int b = 0; b < db->nblocks; b++) {
    ko_block_counts(db, b, kmers, n, counts);
    block_matches(db, b, counts, (int)n, p, &ms, &nm, &cap);
  }
  free(counts);
  if (nm > 0) { out->matches = ms; out->nmatches = nm; return 1; }
  free(ms);
  return 0;
}

/* one pass of the `for _ik, k := range ks` body (:764-1023).  Returns 1 when the result is final (matched, or one of the
 * early returns), 0 when nothing matched and the caller may try the next smaller k. */
static int search_one_k(ko_db* db, const ko_sketch_cfg* cfg, const uint8_t* seq1, size_t len1, const uint8_t* seq2, size_t len2,
                        const ko_search_params* p, ko_result* out, int* found_out) {
  *found_out = 0;
  memset(out, 0, sizeof *out); /* a fresh QueryResult per k (:765); NumKmers of the pooled object is stale: the oracle says 0 */
  out->k = cfg->k;
  out->nmatches = -1;
  out->qlen = (int32_t)len1 + (seq2 ? (int32_t)len2 : 0);
  int try_se = p->try_se && seq2 != NULL;
  if ((int)len1 < p->min_qlen) { /* :778-786 */
    if (!(seq2 && (int)len2 >= p->min_qlen)) { out->qkmers = 0; return 1; }
  }
  uint64_t* kmers = (uint64_t*)malloc((len1 + (seq2 ? len2 : 0) + 1) * sizeof(uint64_t));
  size_t n1 = ko_generate_kmers(seq1, len1, cfg, kmers);
  size_t n = n1;
  if (seq2) n += ko_generate_kmers(seq2, len2, cfg, kmers + n1);
  if ((int)n < p->min_matched) { free(kmers); return 1; } /* :854-869: final, also for the smaller k */
  uint64_t* copy = NULL;
  if (try_se) { copy = (uint64_t*)malloc((n + 1) * sizeof(uint64_t)); memcpy(copy, kmers, n * sizeof(uint64_t)); }
  int found = search_kmers(db, kmers, n, p, out);
  if (!found && try_se) {
    out->qlen = (int32_t)len1; /* tries == 1: read1 */
    memcpy(kmers, copy, n1 * sizeof(uint64_t));
    if ((int)n1 < p->min_matched) { free(kmers); free(copy); return 1; }
    found = search_kmers(db, kmers, n1, p, out);
    if (!found) {
      out->qlen = (int32_t)len2; /* tries == 2: read2 */
      memcpy(kmers, copy + n1, (n - n1) * sizeof(uint64_t));
      if ((int)(n - n1) < p->min_matched) { free(kmers); free(copy); return 1; }
      found = search_kmers(db, kmers, n - n1, p, out);
    }
  }
  free(kmers);
  free(copy);
  *found_out = found;
  return found;
}

/* handleQuery, util-db-search.go:763-1025 (+ --try-se retry :831-850,:1001-1014; smaller k of a multi-k database :764,
 * :1016-1022) and the post-processing of handleQuerySingleDB :260-345 (sort, --keep-top-scores). */
int ko_search(ko_db* db, const uint8_t* seq1, size_t len1, const uint8_t* seq2, size_t len2, const ko_search_params* p,
              ko_result* out) {
  int found = 0;
  for (int ik = 0; ik < db->nks; ik++) {
    ko_sketch_cfg cfg = db->cfg;
    cfg.k = db->ks[ik];
    if (search_one_k(db, &cfg, seq1, len1, seq2, len2, p, out, &found)) break;
  }
  if (found) {
    if (out->nmatches > 1 && !p->do_not_sort) { g_sort_by = p->sort_by; qsort(out->matches, (size_t)out->nmatches, sizeof(ko_match), cmp_match); }
    if (p->top_n_scores > 0 && !p->do_not_sort) { /* :285-311 */
      int nn = 0, i;
      double pscore = 1024;
      for (i = 0; i < out->nmatches; i++) {
        const ko_match* m = &out->matches[i];
        double score = p->sort_by == 1 ? m->tcov : p->sort_by == 2 ? m->jacc : m->qcov;
        if (score < pscore) {
          nn++;
          if (nn > p->top_n_scores) break;
          pscore = score;
        }
      }
      /* the reference keeps [:i+1] where i is the last range index visited */
      if (i >= out->nmatches) i = out->nmatches - 1;
      out->nmatches = i + 1;
    }
  }
  return 0;
}

void ko_result_free(ko_result* r) { free(r->matches); r->matches = NULL; }

/* search.go:517-575 */
int ko_format_match(char* buf, size_t cap, const char* qid, const ko_result* r, const ko_match* m, uint64_t qidx) {
  return snprintf(buf, cap, "%s\t%d\t%d\t%.4e\t%d\t%s\t%d\t%d\t%llu\t%d\t%d\t%.4f\t%.4f\t%.4f\t%llu\n", qid, r->qlen, r->qkmers, m->fpr,
                  r->nmatches, m->target, (int)(uint16_t)m->target_idx, (int)(m->target_idx >> 16), (unsigned long long)m->gsize,
                  r->k, m->mkmers, m->qcov, m->tcov, m->jacc, (unsigned long long)qidx);
}

/* ================================================================================================
 * Batch search for the bench's cpu_baseline leg ("port"): same results as ko_search for single-end
 * plain/scaled k-mer DBs, but threaded over reads and counting with word-wise vertical counters
 * instead of the 64-row byte transposition of :6821-6972 (a faster CPU formulation of the same
 * positional popcount, so the reported baseline errs on the fast side).
 * ============================================================================================== */
typedef uint16_t v8u16 __attribute__((vector_size(16)));
static v8u16 g_lut[256];
static int g_lut_ready = 0;
static void lut_init(void) {
  for (int v = 0; v < 256; v++)
    for (int j = 0; j < 8; j++) g_lut[v][j] = (uint16_t)((v >> (7 - j)) & 1); /* bit 7 = first column of the byte */
  g_lut_ready = 1;
}
static void block_counts_fast(const ko_block* b, const uint64_t* kmers, size_t n, uint16_t* cnt /* row_bytes*8 */, uint8_t* acc) {
  const uint8_t* sigs = b->file + b->offset0;
  uint32_t rb = b->row_bytes;
  memset(cnt, 0, (size_t)rb * 8 * sizeof(uint16_t));
  uint64_t hv[8];
  for (size_t i = 0; i < n; i++) {
    const uint8_t* row;
    if (b->num_hashes == 1) {
      row = sigs + (size_t)(kmers[i] % b->num_sigs) * rb;
    } else {
      ko_hash_values(kmers[i], b->num_hashes, hv);
      memcpy(acc, sigs + (size_t)(hv[0] % b->num_sigs) * rb, rb);
      for (int h = 1; h < b->num_hashes; h++) {
        const uint8_t* r2 = sigs + (size_t)(hv[h] % b->num_sigs) * rb;
        for (uint32_t j = 0; j < rb; j++) acc[j] &= r2[j];
      }
      row = acc;
    }
    v8u16* cv = (v8u16*)cnt;
    for (uint32_t j = 0; j < rb; j++) cv[j] += g_lut[row[j]];
  }
}

int64_t ko_search_batch(ko_db* db, const uint8_t* seqs, const uint64_t* offs, uint32_t n_reads, const ko_search_params* p,
                        int threads, int32_t* qkmers, uint32_t* hits_out, int64_t hits_cap) {
  if (!g_seed_ready) seed_init();
  if (!g_lut_ready) lut_init();
  uint32_t max_rb = 0;
  for (int b = 0; b < db->nblocks; b++) if (db->blocks[b].row_bytes > max_rb) max_rb = db->blocks[b].row_bytes;
  int64_t nh = 0;
  int overflow = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
  {
    uint16_t* cnt = (uint16_t*)aligned_alloc(64, ((size_t)max_rb * 8 * sizeof(uint16_t) + 63) / 64 * 64);
    uint8_t* acc = (uint8_t*)malloc(max_rb ? max_rb : 1);
    uint64_t* kmers = NULL;
    size_t kcap = 0;
#pragma omp for schedule(dynamic, 16)
    for (uint32_t r = 0; r < n_reads; r++) {
      size_t len = (size_t)(offs[r + 1] - offs[r]);
      qkmers[r] = 0;
      if ((int)len < p->min_qlen) continue;
      if (len + 1 > kcap) { kcap = len + 1; kmers = (uint64_t*)realloc(kmers, kcap * sizeof(uint64_t)); }
      size_t n = ko_generate_kmers(seqs + offs[r], len, &db->cfg, kmers);
      if ((int)n < p->min_matched) continue; /* qKmers left at 0 (stale pooled value in the reference) */
      if ((int)n > p->dedup_threshold) n = ko_sort_unique(kmers, n);
      qkmers[r] = (int32_t)n;
      if (n > 65535) continue; /* uint16 counters: the bench workload is 150-bp reads */
      double thr = (double)n * p->min_qcov;
      for (int b = 0; b < db->nblocks; b++) {
        const ko_block* blk = &db->blocks[b];
        block_counts_fast(blk, kmers, n, cnt, acc);
        for (uint32_t c = 0; c < blk->ncols; c++) {
          int count = cnt[c];
          if (count < p->min_matched) continue;
          double cf = (double)count;
          if (!(cf > thr)) continue;
          if (!(cf / (double)blk->sizes[c] >= p->min_tcov)) continue;
          if (!(ko_query_fpr((int)n, count, db->fpr) <= p->max_fpr)) continue;
          int64_t slot;
#pragma omp atomic capture
          slot = nh++;
          if (slot < hits_cap) {
            hits_out[3 * slot] = r; hits_out[3 * slot + 1] = blk->col_base + c; hits_out[3 * slot + 2] = (uint32_t)count;
          } else overflow = 1;
        }
      }
    }
    free(cnt); free(acc); free(kmers);
  }
  (void)overflow;
  return nh;
}


/* ================================================================================================
 * The reference's own loop shape for the bench's second cpu_baseline leg ("reference-shaped port"):
 *  - one worker per index file, each handling every query against its block (NewUnikIndex's `fn`, util-db-search.go:1296-1320,
 *    :7735-7756; with threads <= #blocks extraWorkers (:213-219) is 0), the query goroutines only generate k-mers (:793-919);
 *  - rows are not copied: 64 row pointers are buffered (PosPopCountBufSize, :1164, :6811-6819), then for every byte i of the row
 *    the 64 bytes buffs[j][i] are gathered into buf (the transposition the author calls the bottleneck, :1158-1163, :6823-6966)
 *    and pospop.Count8(&counts[i], buf) adds the per-bit-position popcounts (:6968); the tail of < 64 rows goes through the same
 *    code with a shorter buf (countKmerss[bufIdx], :1448-6539, :7408);
 *  - counts are [NumRowBytes][8]int, reset by copy() per query (:6616); the threshold scan reads _counts[7]..[0] (:7466-7704).
 * pospop v1.2.3 is AVX2 assembly; Count8 here is the same movemask formulation with intrinsics (scalar fallback).
 * Single-hash, single-end databases only (the bench workload); results equal ko_search_batch's.
 * ============================================================================================== */
#if defined(__AVX2__)
#include <immintrin.h>
#endif
static inline void count8(int64_t* counts /*[8]*/, const uint8_t* buf, int n) {
  /* counts[b] += number of bytes of buf with bit b set (pospop.Count8: counts[0] is the LSB position) */
  int i = 0;
#if defined(__AVX2__)
  for (; i + 32 <= n; i += 32) {
    __m256i v = _mm256_loadu_si256((const __m256i*)(buf + i));
    for (int b = 7; b >= 0; b--) {
      counts[b] += __builtin_popcount((unsigned)_mm256_movemask_epi8(v));
      v = _mm256_add_epi8(v, v);
    }
  }
#endif
  for (; i < n; i++) {
    uint8_t v = buf[i];
    for (int b = 0; b < 8; b++) counts[b] += (v >> b) & 1;
  }
}

int64_t ko_search_batch_refshape(ko_db* db, const uint8_t* seqs, const uint64_t* offs, uint32_t n_reads, const ko_search_params* p,
                                 int threads, int32_t* qkmers, uint32_t* hits_out, int64_t hits_cap) {
  if (!g_seed_ready) seed_init();
  if (db->num_hashes != 1) { snprintf(g_err, sizeof g_err, "ko_search_batch_refshape: single-hash databases only"); return -1; }
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  /* query goroutines: k-mers of every read (:793-919) */
  uint64_t* kmers = (uint64_t*)malloc((size_t)(offs[n_reads] + 1) * sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t r = 0; r < n_reads; r++) {
    size_t len = (size_t)(offs[r + 1] - offs[r]);
    qkmers[r] = 0;
    if ((int)len < p->min_qlen) continue;
    size_t n = ko_generate_kmers(seqs + offs[r], len, &db->cfg, kmers + offs[r]);
    if ((int)n < p->min_matched) continue;
    if ((int)n > p->dedup_threshold) n = ko_sort_unique(kmers + offs[r], n);
    qkmers[r] = (int32_t)n;
  }
  int64_t nh = 0;
  /* index workers: one per block */
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < db->nblocks; bi++) {
    const ko_block* b = &db->blocks[bi];
    const uint8_t* sigs = b->file + b->offset0;
    const uint32_t rb = b->row_bytes;
    int64_t (*counts)[8] = (int64_t(*)[8])malloc((size_t)rb * sizeof(int64_t[8]));
    const uint8_t* buffs[64];
    uint8_t buf[64];
    for (uint32_t r = 0; r < n_reads; r++) {
      const int n = qkmers[r];
      if (n <= 0) continue;
      const uint64_t* km = kmers + offs[r];
      memset(counts, 0, (size_t)rb * sizeof(int64_t[8])); /* copy(counts, counts0), :6616 */
      int bufidx = 0;
      for (int i = 0; i < n; i++) {
        buffs[bufidx++] = sigs + (size_t)(km[i] % b->num_sigs) * rb; /* :6811-6816 */
        if (bufidx == 64 || i == n - 1) {
          for (uint32_t c = 0; c < rb; c++) { /* every column byte of the matrix, :6823 */
            for (int j = 0; j < bufidx; j++) buf[j] = buffs[j][c];
            count8(counts[c], buf, bufidx);
          }
          bufidx = 0;
        }
      }
      const double nhf = (double)n, thr = nhf * p->min_qcov;
      for (uint32_t c8 = 0; c8 < rb; c8++) {
        for (int j = 0; j < 8; j++) { /* column 8i+j reads _counts[7-j], :7466-7704 */
          const uint32_t c = c8 * 8 + (uint32_t)j;
          const int count = (int)counts[c8][7 - j];
          if (c >= b->ncols || count < p->min_matched) continue;
          const double cf = (double)count;
          if (!(cf > thr)) continue;
          if (!(cf / (double)b->sizes[c] >= p->min_tcov)) continue;
          if (!(ko_query_fpr(n, count, db->fpr) <= p->max_fpr)) continue;
          int64_t slot;
#pragma omp atomic capture
          slot = nh++;
          if (slot < hits_cap) { hits_out[3 * slot] = r; hits_out[3 * slot + 1] = b->col_base + c; hits_out[3 * slot + 2] = (uint32_t)count; }
        }
      }
    }
    free(counts);
  }
  free(kmers);
  return nh;
}
