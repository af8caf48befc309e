/* shim_replay.c — the C call sequence of shim/kmcp_gpu.go, replayed without Go (there is no Go toolchain in this image).
 *
 * What the cgo shim does per batch, from several OS threads at once:
 *   Submit: malloc two CSR buffers, fill them, kmcpg_submit, free them at once (the library has copied them);
 *   Wait (on ANOTHER thread than the submitter's; round 6: the compact form): kmcpg_wait_pairs, read qlen/qkmers/ksize/match_offs,
 *         per matched query kmcpg_expand_pairs(db, qkmers[i], pairs + offs[i], m, scratch) into a malloc'd scratch array that grows to
 *         the largest query of the batch, read the records, kmcpg_result_pairs_free — exactly (*GPUDB).Wait of shim/kmcp_gpu.go;
 *   open: kmcpg_db_info + one kmcpg_col_info per column (names cached on the caller's side);
 *   errors: the failing call and kmcpg_last_error() on the same thread.
 * One submitter thread and two waiter threads share a handle, with at most IN_FLIGHT tickets between them — RunGPUEngine's
 * structure.  Output: one line per batch "batch <i> reads <n> matches <m> sum <checksum>" that the test compares with the
 * Python binding's results, plus "error-path ok".
 *
 * usage: shim_replay <db_dir> <batches.bin>     batches.bin: u32 n_batches, then per batch u32 n, u64 offs[n+1], bytes
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmcp_gpu.h"

#define IN_FLIGHT 3

typedef struct {
  uint32_t n;
  uint64_t* offs;
  uint8_t* seqs;
} batch_t;

typedef struct {
  int idx;
  uint32_t n;
  kmcpg_ticket* ticket;
} job_t;

static kmcpg_db* g_db;
static char** g_names;
static uint32_t g_ncols;
static batch_t* g_batches;
static uint32_t g_nbatches;
static char (*g_lines)[160];

/* bounded queue of submitted batches + the token count of RunGPUEngine */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static job_t g_q[IN_FLIGHT];
static int g_qn = 0, g_tokens = 0, g_closed = 0;

static void die(const char* what) {
  fprintf(stderr, "%s: %s\n", what, kmcpg_last_error());
  exit(1);
}

static void* submitter(void* arg) {
  (void)arg;
  kmcpg_params p;
  memset(&p, 0, sizeof p);
  p.min_qlen = 30; p.min_matched = 10; p.min_qcov = 0.55; p.max_fpr = 0.01; p.dedup_threshold = 256; p.fpr_buf_size = 249;
  for (uint32_t b = 0; b < g_nbatches; b++) {
    pthread_mutex_lock(&g_mu);
    while (g_tokens == IN_FLIGHT) pthread_cond_wait(&g_cv, &g_mu);
    g_tokens++;
    pthread_mutex_unlock(&g_mu);
    const batch_t* B = &g_batches[b];
    /* the shim's pack(): C.malloc'd copies that live only for the call */
    uint8_t* seqs = (uint8_t*)malloc((size_t)B->offs[B->n] + 1);
    uint64_t* offs = (uint64_t*)malloc(8 * ((size_t)B->n + 1));
    memcpy(seqs, B->seqs, (size_t)B->offs[B->n]);
    memcpy(offs, B->offs, 8 * ((size_t)B->n + 1));
    job_t j;
    j.idx = (int)b;
    j.n = B->n;
    if (kmcpg_submit(g_db, seqs, offs, NULL, NULL, B->n, &p, &j.ticket) != 0) die("kmcpg_submit");
    memset(seqs, 'N', (size_t)B->offs[B->n]);  /* the buffers are the caller's again */
    free(seqs);
    free(offs);
    pthread_mutex_lock(&g_mu);
    g_q[g_qn++] = j;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
  }
  pthread_mutex_lock(&g_mu);
  g_closed = 1;
  pthread_cond_broadcast(&g_cv);
  pthread_mutex_unlock(&g_mu);
  return NULL;
}

static void* waiter(void* arg) {
  (void)arg;
  for (;;) {
    pthread_mutex_lock(&g_mu);
    while (g_qn == 0 && !g_closed) pthread_cond_wait(&g_cv, &g_mu);
    if (g_qn == 0) {
      pthread_mutex_unlock(&g_mu);
      return NULL;
    }
    job_t j = g_q[--g_qn]; /* LIFO on purpose: tickets are waited for out of order */
    pthread_mutex_unlock(&g_mu);
    kmcpg_result_pairs r;
    if (kmcpg_wait_pairs(j.ticket, &r) != 0) die("kmcpg_wait_pairs");
    uint64_t sum = 0, nm = r.match_offs[j.n];
    size_t scratch_cap = 4; /* (the shim starts at 256; small here so that the growth path runs) */
    kmcpg_match* scratch = (kmcpg_match*)malloc(scratch_cap * sizeof(kmcpg_match));
    for (uint32_t i = 0; i < j.n; i++) {
      sum = sum * 1099511628211ULL + (uint64_t)(uint32_t)r.qlen[i] * 31 + (uint64_t)(uint32_t)r.qkmers[i] * 7 + (uint64_t)(uint32_t)r.ksize[i];
      const uint64_t cnt = r.match_offs[i + 1] - r.match_offs[i];
      if (cnt == 0) continue;
      if (cnt > scratch_cap) {
        free(scratch);
        scratch_cap = (size_t)(cnt + cnt / 2);
        scratch = (kmcpg_match*)malloc(scratch_cap * sizeof(kmcpg_match));
      }
      if (kmcpg_expand_pairs(g_db, r.qkmers[i], r.pairs + r.match_offs[i], cnt, scratch) != 0) die("kmcpg_expand_pairs");
      for (uint64_t m = 0; m < cnt; m++) {
        const kmcpg_match* M = &scratch[m];
        if (M->col >= g_ncols) die("column out of range");
        uint64_t qc;
        memcpy(&qc, &M->qcov, 8);
        sum = sum * 1099511628211ULL + ((uint64_t)M->col << 20) + (uint64_t)(uint32_t)M->mkmers + qc + (uint64_t)strlen(g_names[M->col]);
      }
    }
    free(scratch);
    kmcpg_result_pairs_free(&r);
    snprintf(g_lines[j.idx], sizeof g_lines[0], "batch %d reads %u matches %llu sum %016llx", j.idx, j.n, (unsigned long long)nm, (unsigned long long)sum);
    pthread_mutex_lock(&g_mu);
    g_tokens--;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
  }
}

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: shim_replay <db_dir> <batches.bin>\n");
    return 2;
  }
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(&g_nbatches, 4, 1, f) != 1) return 2;
  g_batches = (batch_t*)calloc(g_nbatches, sizeof(batch_t));
  g_lines = calloc(g_nbatches, sizeof g_lines[0]);
  for (uint32_t b = 0; b < g_nbatches; b++) {
    batch_t* B = &g_batches[b];
    if (fread(&B->n, 4, 1, f) != 1) return 2;
    B->offs = (uint64_t*)malloc(8 * ((size_t)B->n + 1));
    if (fread(B->offs, 8, (size_t)B->n + 1, f) != (size_t)B->n + 1) return 2;
    B->seqs = (uint8_t*)malloc((size_t)B->offs[B->n] + 1);
    if (B->offs[B->n] && fread(B->seqs, 1, (size_t)B->offs[B->n], f) != (size_t)B->offs[B->n]) return 2;
  }
  fclose(f);

  kmcpg_opts o = {0, 0, 1, 0};
  if (kmcpg_open(argv[1], &o, &g_db) != 0) die("kmcpg_open");
  kmcpg_info info;
  if (kmcpg_db_info(g_db, &info) != 0) die("kmcpg_db_info");
  g_ncols = (uint32_t)info.n_cols;
  g_names = (char**)calloc(g_ncols, sizeof(char*));
  for (uint32_t c = 0; c < g_ncols; c++) { /* OpenGPUDB caches the names: no call per match later */
    const char* nm = NULL;
    if (kmcpg_col_info(g_db, c, &nm, NULL, NULL, NULL) != 0) die("kmcpg_col_info");
    g_names[c] = strdup(nm);
  }
  /* error path: the failing call and the message on one thread; the handle stays usable */
  {
    kmcpg_params bad;
    memset(&bad, 0, sizeof bad);
    bad.min_matched = 10; bad.min_qcov = 0.55; bad.max_fpr = 0.01; bad.dedup_threshold = 256;
    bad.k = 7; /* not a k-mer size of the database */
    uint64_t offs1[2] = {0, 40};
    uint8_t seq1[40];
    memset(seq1, 'A', sizeof seq1);
    kmcpg_result r;
    int rc = kmcpg_search_batch(g_db, seq1, offs1, NULL, NULL, 1, &bad, &r);
    if (rc == 0 || strlen(kmcpg_last_error()) == 0) {
      fprintf(stderr, "expected an error for k=7\n");
      return 1;
    }
    kmcpg_ticket* t = NULL;
    if (kmcpg_submit(g_db, NULL, NULL, NULL, NULL, 5, NULL, &t) == 0 || t != NULL) return 1; /* null buffers */
    printf("error-path ok\n");
  }
  pthread_t ts, tw[2];
  pthread_create(&ts, NULL, submitter, NULL);
  for (int i = 0; i < 2; i++) pthread_create(&tw[i], NULL, waiter, NULL);
  pthread_join(ts, NULL);
  for (int i = 0; i < 2; i++) pthread_join(tw[i], NULL);
  for (uint32_t b = 0; b < g_nbatches; b++) puts(g_lines[b]);
  if (kmcpg_close(g_db) != 0) die("kmcpg_close");
  for (uint32_t c = 0; c < g_ncols; c++) free(g_names[c]);
  free(g_names);
  for (uint32_t b = 0; b < g_nbatches; b++) { free(g_batches[b].offs); free(g_batches[b].seqs); }
  free(g_batches);
  free(g_lines);
  return 0;
}
