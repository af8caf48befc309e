// query.cpp — the GPU half behind the C ABI: k-mer generation (K1, K1d) and the COBS query (K2) on device pointers.
// Reference counterparts: generateKmers (util-db-search.go:1037-1107), dedup (:874-908), the UnikIndex workers (:6611-7742).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "dbformat.hpp"
#include "engine.hpp"
#include "fpr.hpp"
#include "kernels.hpp"

using namespace kmcpg;

namespace kmcpg {
void release_fpr_bounds(kmcpg_db* db) {
  for (auto& t : db->fpr_bounds) {
    if (t.d) (void)hipFree(t.d);
    if (t.h) (void)hipHostFree(t.h);
  }
  db->fpr_bounds.clear();
}
}  // namespace kmcpg

// ------------------------------------------------------------------------------------------------
// GPU half
// ------------------------------------------------------------------------------------------------
thread_local kmcpg::PackedSrc kmcpg::tl_packed_src;

// a batch the k-mer stage hashes segment by segment (run_kmers: plain or FracMinHash k-mers, single-end, some query above one segment)
bool kmcpg::whole_genome_batch(const kmcpg_db* db, uint32_t max_read_len, bool paired) {
  return !paired && !db->info.syncmer && !db->info.minimizer && max_read_len > (uint32_t)k1_segment_len();
}

namespace {

uint64_t max_hash_for(uint32_t scale) {
  // uint64(float64(^uint64(0)) / float64(scale))  (util-db-search.go:1040-1043)
  const double d = 18446744073709551616.0 / (double)scale;
  if (d >= 18446744073709551616.0) return ~0ULL;
  return (uint64_t)d;
}

// K1 (+K1d): hashes of read i end up at d_hashes[offs[i] + offs2[i] ...], NumKmers in d_nk_search
int run_kmers(kmcpg_db* db, kmcpg_db::Workspace& W, const uint8_t* d_seqs, const uint64_t* d_offs, const uint8_t* d_seqs2, const uint64_t* d_offs2, uint32_t n_reads,
              uint32_t max_read_len, const kmcpg_params& p, uint64_t* d_hashes, uint64_t* d_scratch, uint64_t scratch_half, int32_t* d_nk_raw, int32_t* d_nk1,
              int32_t* d_nk_search, int32_t* d_qlen, hipStream_t st, uint64_t* max_n_out) {
  const kmcpg_info& I = db->info;
  if (!I.canonical) return kmcpg_fail(KMCPG_EUNSUPPORTED, "non-canonical index");
  K1Args a{};
  a.seqs = d_seqs;
  a.offs = d_offs;
  a.seqs2 = d_seqs2;
  a.offs2 = d_offs2;
  a.n_reads = n_reads;
  a.k = p.k > 0 ? p.k : I.k;
  a.min_qlen = p.min_qlen;
  a.scaled = I.scaled;
  a.max_hash = I.scaled ? max_hash_for(I.scale) : ~0ULL;
  a.mode = I.syncmer ? 2 : (I.minimizer ? 1 : 0);  // syncmer > minimizer > plain (:1052-1058)
  a.w_or_s = I.syncmer ? I.syncmer_s : I.minimizer_w;
  a.hashes = d_hashes;
  a.scratch = d_scratch;
  a.scratch2 = d_scratch ? d_scratch + scratch_half : nullptr;
  a.nk_raw = d_nk_raw;
  a.nk1 = d_nk1;
  a.qlen = d_qlen;
  a.flags = getenv("KMCPG_K1_FLAGS") ? atoi(getenv("KMCPG_K1_FLAGS")) : 3;
  // whole genomes (single-end, plain or FracMinHash k-mers): segments of a read on their own workgroups
  const uint32_t segs = (max_read_len + (uint32_t)k1_segment_len() - 1) / (uint32_t)k1_segment_len();
  if (a.mode == 0 && !d_seqs2 && segs > 1 && d_scratch && (uint64_t)n_reads * segs <= (1ull << 21)) {  // one workgroup of 1024 threads per segment, < 2^32 threads per launch
    if (W.w_seg_cnt.ensure(3 * (size_t)n_reads * segs + 2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");  // counts + launch_k1's fallback list + its marks
    a.seg_cnt = W.w_seg_cnt.p;
    a.segs_max = segs;
  }
  // A batch that came as 2-bit codes (host.cpp: kmcpg_submit_packed, or text stage() packed): the whole-genome kernel reads the codes as
  // they are, and text exists only for the segments a foreign byte reaches (launch_k1); every other k-mer kernel reads text, expanded
  // here.  (A batch with a run per 4 kb or more is not what the direct form is for: expanded whole.)  KMCPG_K1_CODES=0: always expand.
  const PackedSrc src = tl_packed_src;
  tl_packed_src = PackedSrc{};
  if (src.codes) {
    static const int codes_mode = getenv("KMCPG_K1_CODES") ? atoi(getenv("KMCPG_K1_CODES")) : 1;  // 2 (tests): however many runs there are
    const bool direct = codes_mode != 0 && a.seg_cnt && a.segs_max > 1 && a.k <= 128 && !(a.flags & 24) && src.text == d_seqs &&
                        (codes_mode == 2 || (uint64_t)src.n_exc <= src.n_bases / 4096 + 64);
    if (direct) {
      a.codes = src.codes;
      a.exc = src.exc;
      a.n_exc = src.n_exc;
      a.seqs_w = src.text;
    } else {
      launch_unpack2(src.codes, src.text, src.n_bases, src.n_exc ? src.exc : nullptr, src.n_exc, st);  // codes -> the text the kernels read
    }
    (direct ? db->k1_codes_direct : db->k1_codes_expanded)++;
  }
  if (a.mode != 0 && !d_seqs2 && d_scratch) {  // the list the rolling window-sketch kernel leaves to k1_windows_wave (launch_k1): count + read indices
    if (W.w_seg_cnt.ensure((size_t)n_reads + 2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    a.seg_nflag = (uint32_t*)W.w_seg_cnt.p;
    a.seg_list = a.seg_nflag + 1;
  }
  a.nk_adj = d_nk_search;
  a.dedup_threshold = p.dedup_threshold;
  const bool adj_done = launch_k1(a, max_read_len, st);
  uint64_t ub = max_read_len >= (uint32_t)a.k ? (uint64_t)(max_read_len - a.k + 1) : 0;
  if (d_seqs2) ub *= 2;
  *max_n_out = ub;
  if (ub > (uint64_t)p.dedup_threshold) {
    DedupArgs d{};
    d.offs = d_offs;
    d.offs2 = d_offs2;
    d.n_reads = n_reads;
    d.dedup_threshold = p.dedup_threshold;
    d.min_matched = p.min_matched;
    d.hashes = d_hashes;
    d.scratch = d_scratch;
    d.nk_raw = d_nk_raw;
    d.nk_search = d_nk_search;
    d.pre = a.mode != 0;
    d.pre_done = adj_done;
    d.key_shift = (I.scaled && a.max_hash) ? __builtin_clzll(a.max_hash) : 0;
    launch_dedup(d, ub, st);
    if (ub > HUGE_MIN) {
      // whole-genome queries: which ones they are is only known on the device -> one small read-back, then a device-wide
      // sort + unique per such query
      const int32_t thr = std::max<int32_t>((int32_t)HUGE_MIN, p.dedup_threshold);
      uint32_t meta[2] = {0, 0};
      if (W.w_long_list.ensure(n_reads + 1) || W.w_long_meta.ensure(2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
      HIPCHK(hipMemsetAsync(W.w_long_meta.p, 0, 2 * sizeof(uint32_t), st));
      launch_list_long(d_nk_raw, n_reads, thr, W.w_long_list.p, W.w_long_meta.p, st);
      HIPCHK(hipMemcpyAsync(meta, W.w_long_meta.p, sizeof meta, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (meta[0]) {
        const size_t tb = huge_dedup_temp_bytes(meta[1]);
        if (W.w_huge_info.ensure(3 * (size_t)meta[0] + 1) || W.w_huge_temp.ensure(tb + 64)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
        launch_gather_huge(W.w_long_list.p, meta[0], d_nk_raw, d_offs, d_offs2, W.w_huge_info.p, st);
        std::vector<uint64_t> hinfo(3 * (size_t)meta[0]);
        HIPCHK(hipMemcpyAsync(hinfo.data(), W.w_huge_info.p, hinfo.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int* d_num = (int*)W.w_huge_temp.p;  // first 64 bytes: the unique count
        for (uint32_t i = 0; i < meta[0]; i++) {
          const uint32_t r = (uint32_t)hinfo[3 * i], n = (uint32_t)hinfo[3 * i + 1];
          const uint64_t koff = hinfo[3 * i + 2];
          if (huge_dedup(d_hashes + koff, d_scratch + koff, n, d_num, W.w_huge_temp.p + 64, tb, d_nk_search, r, p.min_matched, st) != 0)
            return kmcpg_fail(KMCPG_EDEVICE, "device-wide sort of a %u-k-mer query failed", n);
        }
      }
    }
  } else {
    launch_nk_simple(d_nk_raw, d_nk_search, n_reads, p.min_matched, st);
  }
  return 0;
}

// The k-mer workspace exists twice (engine.hpp Workspace): calls are enqueued under db->mu and take the slots in turn; a call waits
// for the last kernel of the previous user of ITS slot, so the k-mer kernels of a batch may run beside the COBS kernels of the
// batch before it when the two calls are on different streams.  The second slot costs a second workspace (24 bytes per base of a
// batch): it is used only while that fits into a quarter of the free HBM (or is there already).
// Which batches: by default those of WHOLE GENOMES only (kmcpg::whole_genome_batch) — their k-mer stage is one fat VALU-bound kernel that fits
// beside the memory-bound COBS kernel of the batch before (genome search 47.2 -> 51.5 k genomes/s, profiles/r06_cobs_overlap.txt); on every other
// shape the second stream costs more in cross-stream waits than it hides (-3 ... -11 %, profiles/r05_k1_beside_k2.txt and the same file).
// KMCPG_WS_SLOTS=1: never, 2: every batch.
int pick_slot(kmcpg_db* db, uint64_t total_bases, bool whole_genomes) {
  static const int env = getenv("KMCPG_WS_SLOTS") ? atoi(getenv("KMCPG_WS_SLOTS")) : -1;
  if (env >= 0 ? env < 2 : !whole_genomes) return 0;
  // (by default not before the handle's fifth batch: the second workspace of a 256-Mbase batch is 6 GB to allocate and map, more than a run of
  // `kmcp-search -g` over a few hundred assemblies — four batches — can win back: 0.38 -> 0.49 s exec to exit when it was taken at once)
  // (... and then at once, with the fifth: a caller that warms up with five batches has the allocation behind it)
  if (env < 0 && db->ws_calls < 4) return 0;
  const int slot = (int)((db->ws_calls + (env < 0 ? 1 : 0)) & 1);
  if (slot == 0) return 0;
  kmcpg_db::Workspace& W = db->ws[1];
  if (W.w_hashes.cap >= total_bases + 1 && W.w_scratch.cap >= 2 * total_bases + 2) return 1;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  uint64_t limit = free_b;
  if (const char* e = getenv("KMCPG_HBM_LIMIT_MB")) limit = std::min<uint64_t>(limit, (uint64_t)std::max(0ll, atoll(e)) << 20);
  return 24 * (total_bases + 2) <= limit / 4 ? 1 : 0;
}

int ws_begin(kmcpg_db::Workspace& W, hipStream_t st) {
  if (!W.ev) HIPCHK(hipEventCreateWithFlags(&W.ev, hipEventDisableTiming));
  if (W.ev_valid) HIPCHK(hipStreamWaitEvent(st, W.ev, 0));
  return 0;
}

int ws_end(kmcpg_db::Workspace& W, hipStream_t st) {
  HIPCHK(hipEventRecord(W.ev, st));
  W.ev_valid = true;
  return 0;
}

// Every way out of a GPU-half call that has passed ws_begin leaves the event behind, error paths included: kernels that use
// the workspace may already be on the stream when a later step fails (an allocation, the FPR table, a launch), and the next
// user of the slot — possibly on another stream — must still be ordered behind them.
struct WsGuard {
  kmcpg_db::Workspace& W;
  hipStream_t st;
  hipStream_t kst = nullptr;  // the k-mer kernels' own stream when they have one (KMCPG_K1_STREAM=1), else == st
  bool armed = true;
  ~WsGuard() {
    if (!armed || !W.ev) return;
    // work queued on kst must be covered too: st waits for it first, then the slot's event on st stands for both streams
    if (kst != st && W.k1_ev && hipEventRecord(W.k1_ev, kst) == hipSuccess) (void)hipStreamWaitEvent(st, W.k1_ev, 0);
    if (hipEventRecord(W.ev, st) == hipSuccess) W.ev_valid = true;
  }
  int finish() {  // the success path: errors of the record are reported
    armed = false;
    return ws_end(W, st);
  }
};

// a per-handle event of its own kind (COBS kernels one batch at a time; K3's scratch one user at a time)
int chain_begin(hipEvent_t* ev, bool valid, hipStream_t st) {
  if (!*ev) HIPCHK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  if (valid) HIPCHK(hipStreamWaitEvent(st, *ev, 0));
  return 0;
}

// The reference drops a column whose FPR(n, count) exceeds -f right where it counts it (util-db-search.go:7474-7478); here
// that test runs on the host in float64, but the GPU can already leave out every count that cannot pass it: for each
// NumKmers n <= kFprBoundMaxN the smallest count c with FPR(n, c) <= max_fpr (the very values kmcpg_finalize compares, so a
// count below it fails there by definition; nothing is assumed about monotonicity).  With the defaults the query-coverage
// threshold is the stricter one; with -t just above the database's FPR it is this bound that keeps the hit list — and, through
// the pruning test, the row traffic — from exploding (n = 130, p = 0.3, -t 0.31: 45 % of all columns would be "hits").
// The table covers n <= 512 (every short read, single or paired up to 2 x 250; 18 ms of FPR rows on first use) and grows to
// 1024 when a batch may hold longer queries (+60 ms once).  Beyond ~1 100 k-mers the reference's FPR is numerically dead
// anyway: BinomialCoeff leaves the float64 range, the running value drops below zero and is clamped (util-fpr.go:32-50), so
// FPR(n, c) = 0 long before the crossing and the bound could never be the stricter threshold.
constexpr int kFprBoundMinN = kFprBoundAlways, kFprBoundMaxN = 1024;

int fpr_bound(kmcpg_db* db, double max_fpr, uint64_t max_kmers, hipStream_t st, const uint16_t** out, int32_t* out_n) {
  *out = nullptr;
  *out_n = 0;
  if (!fpr_bound_enabled()) return 0;
  int want_n = kFprBoundMinN;
  while (want_n < kFprBoundMaxN && (uint64_t)want_n < max_kmers) want_n *= 2;
  uint64_t key;
  memcpy(&key, &max_fpr, sizeof key);
  // One immutable table per (-f value, size): kernels of earlier calls may still read theirs, so a table is never rewritten —
  // another -f value, or a batch with longer queries, gets a table of its own (uploaded from its pinned copy on the caller's
  // stream: nothing here waits for the GPU, kmcpg_submit stays non-blocking).  Tables live until kmcpg_close.
  for (const auto& t : db->fpr_bounds)
    if (t.key == key && t.n >= want_n) {
      *out = t.d;
      *out_n = t.n;
      return 0;
    }
  if (db->fpr_bounds.size() >= 64) {  // a host that sweeps -f: start over once nothing can be reading the old tables
    HIPCHK(hipDeviceSynchronize());
    release_fpr_bounds(db);
  }
  FprBoundTable t{};
  t.key = key;
  t.n = want_n;
  const size_t bytes = ((size_t)want_n + 1) * sizeof(uint16_t);
  HIPCHK(hipHostMalloc((void**)&t.h, bytes, hipHostMallocDefault));
  if (hipMalloc((void**)&t.d, bytes) != hipSuccess) {
    (void)hipHostFree(t.h);
    return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  }
  QueryFpr* F = db->fpr.get();
  t.h[0] = 0;
  for (int n = 1; n <= want_n; n++) {
    const FprRow held = F->ensure_row(n);
    const std::vector<double>& row = *held;
    int c = 0;
    while (c <= n && !(QueryFpr::value(row, n, c) <= max_fpr)) c++;
    t.h[n] = (uint16_t)c;  // n + 1: no count passes
  }
  db->fpr_bounds.push_back(t);
  HIPCHK(hipMemcpyAsync(t.d, t.h, bytes, hipMemcpyHostToDevice, st));
  *out = t.d;
  *out_n = t.n;
  return 0;
}

}  // namespace

extern "C" int kmcpg_kmers_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, uint32_t n_reads, uint64_t total_bases,
                                  uint32_t max_read_len, const kmcpg_params* params, uint64_t* d_hashes, uint64_t hashes_cap, uint64_t* d_koff,
                                  int32_t* d_nk, void* stream) {
  if (!db || !d_seqs || !d_offs || !d_hashes || !d_nk) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (hashes_cap < total_bases) return kmcpg_fail(KMCPG_EINVAL, "hashes_cap must be >= total_bases");
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  const kmcpg_params p = params ? *params : default_params();
  hipStream_t st = (hipStream_t)stream;
  kmcpg_db::Workspace& W = db->ws[0];
  if (int rc0 = ws_begin(W, st)) return rc0;
  WsGuard wsg{W, st, st};
  if (W.w_scratch.ensure(2 * total_bases + 2) || W.w_nk_raw.ensure(n_reads + 1) || W.w_nk1.ensure(n_reads + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  DevBuf<int32_t> ql;
  if (ql.ensure(n_reads + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  uint64_t maxn = 0;
  int rc = run_kmers(db, W, d_seqs, d_offs, nullptr, nullptr, n_reads, max_read_len, p, d_hashes, W.w_scratch.p, total_bases + 1, W.w_nk_raw.p, W.w_nk1.p,
                     d_nk, ql.p, st, &maxn);
  if (rc == 0 && d_koff) HIPCHK(hipMemcpyAsync(d_koff, d_offs, (size_t)n_reads * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
  hipError_t e = hipStreamSynchronize(st);
  ql.release();
  if (rc) return rc;
  if (e != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "k-mer kernel failed: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int kmcpg_kmers_device_packed(kmcpg_db* db, const uint8_t* d_codes, const kmcpg_exc_run* d_exc, uint32_t n_exc, uint8_t* d_text,
                                         const uint64_t* d_offs, uint32_t n_reads, uint64_t total_bases, uint32_t max_read_len,
                                         const kmcpg_params* params, uint64_t* d_hashes, uint64_t hashes_cap, uint64_t* d_koff, int32_t* d_nk,
                                         void* stream) {
  static_assert(sizeof(kmcpg_exc_run) == sizeof(ExcRun), "one layout");
  if (!d_codes || !d_text || (n_exc && !d_exc)) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (((uintptr_t)d_codes & 3) || ((uintptr_t)d_text & 3)) return kmcpg_fail(KMCPG_EINVAL, "d_codes and d_text must be 4-byte aligned");
  PackedSrc src;
  src.codes = d_codes;
  src.exc = reinterpret_cast<const ExcRun*>(d_exc);
  src.n_exc = n_exc;
  src.text = d_text;
  src.n_bases = total_bases;
  tl_packed_src = src;
  const int rc = kmcpg_kmers_device(db, d_text, d_offs, n_reads, total_bases, max_read_len, params, d_hashes, hashes_cap, d_koff, d_nk, stream);
  tl_packed_src = PackedSrc{};  // (a call that failed before the k-mer kernels has not taken it)
  return rc;
}

extern "C" int kmcpg_k1_codes_batches(kmcpg_db* db, uint64_t* direct, uint64_t* expanded) {
  if (!db) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  if (direct) *direct = db->k1_codes_direct;
  if (expanded) *expanded = db->k1_codes_expanded;
  return 0;
}

extern "C" int kmcpg_query_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, const uint8_t* d_seqs2, const uint64_t* d_offs2,
                                  uint32_t n_reads, uint64_t total_bases, uint32_t max_read_len, const kmcpg_params* params, kmcpg_hit* d_hits,
                                  uint64_t hit_cap, uint64_t* d_counters, int32_t* d_qkmers, int32_t* d_qlen, void* stream) {
  return kmcpg::query_device_after(db, d_seqs, d_offs, d_seqs2, d_offs2, n_reads, total_bases, max_read_len, params, d_hits, hit_cap, d_counters, d_qkmers, d_qlen,
                                   stream, nullptr);
}

// kmcpg_query_device with a prologue that runs once the handle's enqueue lock is held, i.e. right in front of this batch's first kernel.
// host.cpp puts "wait for this batch's upload" (+ the expansion of packed input) there: enqueued BEFORE taking the lock, that wait could
// land in the kernel stream between the k-mer kernels and the COBS kernel of the batch before — a call that reads a word back in the middle
// (whole-genome queries) releases nothing until it returns — and the earlier batch's COBS kernel then sat behind the later batch's 4.5-ms
// upload (profiles/r06_h2h.txt: 2.4 ms of idle GPU per batch of 256 assemblies).
int kmcpg::query_device_after(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, const uint8_t* d_seqs2, const uint64_t* d_offs2, uint32_t n_reads,
                              uint64_t total_bases, uint32_t max_read_len, const kmcpg_params* params, kmcpg_hit* d_hits, uint64_t hit_cap, uint64_t* d_counters,
                              int32_t* d_qkmers, int32_t* d_qlen, void* stream, const std::function<int()>* prologue) {
  if (!db || !d_seqs || !d_offs || !d_counters || !d_qkmers || !d_qlen || (!d_hits && hit_cap)) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if ((d_seqs2 == nullptr) != (d_offs2 == nullptr)) return kmcpg_fail(KMCPG_EINVAL, "seqs2 and offs2 must be given together");
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  if (prologue)
    if (int rcp = (*prologue)()) return rcp;
  const kmcpg_params p = params ? *params : default_params();
  if (p.min_matched < 1) return kmcpg_fail(KMCPG_EINVAL, "min_matched must be >= 1");  // getFlagPositiveInt (search.go:165)
  if (p.k > 0 && std::find(db->ks_desc.begin(), db->ks_desc.end(), p.k) == db->ks_desc.end())
    return kmcpg_fail(KMCPG_EINVAL, "k=%d is not a k-mer size of this database", p.k);
  const int k_used = p.k > 0 ? p.k : db->info.k;
  hipStream_t st = (hipStream_t)stream;
  // test hook: behave as if the k-mer workspace of a batch above this many bases could not be allocated (the batch-halving path
  // of kmcpg_search_batch, tests/test_gpu_paged.py)
  // (honoured only together with KMCPG_TEST_HOOKS=1: a stray variable in a production environment must not fake an ENOMEM)
  static const bool test_hooks = getenv("KMCPG_TEST_HOOKS") && atoi(getenv("KMCPG_TEST_HOOKS")) == 1;
  if (test_hooks)
    if (const char* e = getenv("KMCPG_TEST_MAX_BASES"))
      if (total_bases > (uint64_t)atoll(e)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed (KMCPG_TEST_MAX_BASES)");
  const int slot = pick_slot(db, total_bases, whole_genome_batch(db, max_read_len, d_seqs2 != nullptr));
  db->ws_calls++;
  db->ws_last = slot;
  kmcpg_db::Workspace& W = db->ws[slot];
  // experiment (KMCPG_K1_STREAM=1): the k-mer kernels on a high-priority stream of the handle's own, so that their workgroups are
  // dispatched AHEAD of the previous batch's COBS workgroups whenever a slot frees up (on equal terms the dispatcher keeps feeding the
  // kernel that came first and the k-mer kernels only get the COBS kernel's tail: tools/ab/r05_call3.sh)
  static const bool k1_own = getenv("KMCPG_K1_STREAM") && atoi(getenv("KMCPG_K1_STREAM")) == 1;
  hipStream_t kst = st;
  if (k1_own) {
    if (!db->k1_stream) {
      int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
      HIPCHK(hipStreamCreateWithPriority(&db->k1_stream, hipStreamNonBlocking, hi));
    }
    if (!W.in_ev) HIPCHK(hipEventCreateWithFlags(&W.in_ev, hipEventDisableTiming));
    if (!W.k1_ev) HIPCHK(hipEventCreateWithFlags(&W.k1_ev, hipEventDisableTiming));
    kst = db->k1_stream;
    HIPCHK(hipEventRecord(W.in_ev, st));  // the batch's inputs are ordered on the caller's stream
    HIPCHK(hipStreamWaitEvent(kst, W.in_ev, 0));
  }
  if (int rc0 = ws_begin(W, kst)) return rc0;
  WsGuard wsg{W, st, kst};
  if (W.w_hashes.ensure(total_bases + 1) || W.w_nk_raw.ensure(n_reads + 1) || W.w_nk1.ensure(n_reads + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  uint64_t ub = max_read_len >= (uint32_t)k_used ? (uint64_t)(max_read_len - k_used + 1) : 0;
  if (d_seqs2) ub *= 2;
  const bool window_sketch = db->info.syncmer || db->info.minimizer;
  if ((ub > (uint64_t)p.dedup_threshold || window_sketch) && W.w_scratch.ensure(2 * total_bases + 2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  uint64_t maxn = 0;
  hipEvent_t* pev = db->ev + 4 * (db->ev_calls % 4);
  if (db->profiling) {
    for (auto& ev : db->ev)
      if (!ev) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipEventRecord(pev[0], kst));
  }
  int rc = run_kmers(db, W, d_seqs, d_offs, d_seqs2, d_offs2, n_reads, max_read_len, p, W.w_hashes.p, W.w_scratch.p, total_bases + 1, W.w_nk_raw.p,
                     W.w_nk1.p, d_qkmers, d_qlen, kst, &maxn);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(d_counters, 0, 2 * sizeof(uint64_t), kst));
  launch_max_nk(d_qkmers, n_reads, (unsigned long long*)d_counters + 1, kst);
  if (db->profiling) HIPCHK(hipEventRecord(pev[3], kst));  // k-mers done
  if (k1_own) {
    HIPCHK(hipEventRecord(W.k1_ev, kst));
    HIPCHK(hipStreamWaitEvent(st, W.k1_ev, 0));
  }
  static const int debug_rowsort = getenv("KMCPG_DEBUG_ROWSORT") ? atoi(getenv("KMCPG_DEBUG_ROWSORT")) : 0;
  if (debug_rowsort && !d_offs2 && !db->h_groupdev.empty())  // experiment only: profiles/r05_rowsort_gate.txt
    launch_debug_rowsort(W.w_hashes.p, d_offs, d_qkmers, n_reads, db->h_groupdev[0].num_sigs, db->h_groupdev[0].magic_hi, debug_rowsort, st);
  // long queries (whole genomes, -g) are split into chunks of k-mers so that they spread over the chip; short ones keep
  // the one-wave-per-(query, slot) kernel.  Which queries are long is only known on the device: one small D2H read.
  const char* sm_env = getenv("KMCPG_SPLIT_MIN");
  const int32_t split_min = sm_env ? atoi(sm_env) : 2048;
  uint32_t long_meta[2] = {0, 0};
  size_t total_slots = 0;
  for (const auto& c : db->classes) total_slots += c.slots.size();
  // The read-back below costs a host round trip in the middle of the batch (~2.5 ms: more than the kernels of a batch of HiFi
  // reads take).  It is only worth it when splitting could pay: a batch that fills the chip with its (query, slot) pairs anyway
  // and whose queries are bounded by 32 768 k-mers (HiFi reads, contigs) runs the plain kernel on 16 planes without asking.
  const bool ask = split_min > 0 && maxn > (uint64_t)split_min && (sm_env || maxn > 32768 || (uint64_t)n_reads * total_slots <= 16384);
  if (ask) {
    if (W.w_long_list.ensure(n_reads + 1) || W.w_long_meta.ensure(2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    HIPCHK(hipMemsetAsync(W.w_long_meta.p, 0, 2 * sizeof(uint32_t), st));
    launch_list_long(d_qkmers, n_reads, split_min, W.w_long_list.p, W.w_long_meta.p, st);
    HIPCHK(hipMemcpyAsync(long_meta, W.w_long_meta.p, sizeof long_meta, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  uint32_t n_long = long_meta[0];
  // splitting pays when the long queries alone would leave the chip idle (few (query, slot) pairs) or need more than 16
  // counter planes; a batch of thousands of 10-kb reads already fills it and keeps the plain kernel (unless forced by env)
  // (round 4, genome search with 3 hash functions, same-box A/B over batch sizes: from ~1 500 (query, slot) units on — 1.5 waves per SIMD — the
  // plain kernel wins, because it prunes (a third of the row bytes are never fetched for 8 000-k-mer sketches at -t 0.4) and needs no atomics:
  // 192 queries x 8 slots 5.0 vs 6.0 ms, 512 x 8 11.1 vs 16.3 ms; at 128 x 8 the chunked form still leads, 4.1 vs 4.5 ms.)
  if (n_long && !sm_env && (uint64_t)n_long * total_slots >= 1536 && long_meta[1] <= 65534) n_long = 0;
  // largest NumKmers the plain kernel will meet: bounded by the read length, and exactly known once the long ones were listed
  uint64_t max_short = maxn;
  if (ask)
    max_short = n_long ? (uint64_t)split_min : std::max<uint64_t>(long_meta[1], (uint64_t)split_min);
  // counter planes: 8 for single short reads, 10 for pairs (2 x 150 bp = 260 k-mers, up to 2 x 500 bp), 16 for long reads
  // (one below the planes' range: the largest threshold a query of n k-mers can get is n + 1 — `-t 1`, or an FPR bound no count passes —
  // and k2_cobs compares counts with it on NPL bits: n + 1 <= 2^NPL - 1 keeps that compare exact.  A threshold past the planes' range
  // made the kernel's epilogue emit every column with a count above its low bits: filtered again by the host half, so no wrong
  // match, but a hit list of the whole row for reads of exactly 255 / 1 023 k-mers at -t 1.)
  const int npl = max_short <= 254 ? 8 : (max_short <= 1022 ? 10 : (max_short <= 65534 ? 16 : (max_short <= 16777214 ? 24 : 0)));
  if (!npl) return kmcpg_fail(KMCPG_EUNSUPPORTED, "queries with more than 16777214 k-mers need KMCPG_SPLIT_MIN > 0");
  K2Args a{};
  a.blocks = db->d_groupdev;
  a.segs = db->d_segs;
  a.n_reads = n_reads;
  a.hashes = W.w_hashes.p;
  a.offs = d_offs;
  a.offs2 = d_offs2;
  a.nk = d_qkmers;
  a.min_qcov = p.min_qcov;
  a.min_matched = p.min_matched;
  a.num_hashes = db->info.num_hashes;
  a.nt_loads = getenv("KMCPG_NT_LOADS") ? atoi(getenv("KMCPG_NT_LOADS")) : 1;
  a.prune = getenv("KMCPG_PRUNE") ? atoi(getenv("KMCPG_PRUNE")) : 1;
  // Rows gathered between two pruning tests.  4 instead of 8 saves 2.3 % of the row traffic (sectors are dropped ~2 rows sooner)
  // for ~15 % more VALU work: a gain where the kernel waits for HBM (GTDB scale, 8 planes: 511 -> 488 ms per 524 k reads), a loss
  // where it runs near its issue limits (16-plane kernels at 3 waves per SIMD: 248 -> 361 ms; indexes that half live in the
  // Infinity Cache: 17.8 -> 19.6 ms) — profiles/r02_group_rows.txt.
  a.group_rows = (a.prune && npl <= 10 && db->info.matrix_bytes_local >= (4ull << 30)) ? 4 : 8;
  if (const char* e = getenv("KMCPG_GROUP_ROWS")) a.group_rows = atoi(e) == 4 ? 4 : 8;
  // How often the test runs in the 8/10-plane kernels: after every group (they wait for HBM; KMCPG_PRUNE_EVERY = 2/4/8 for experiments).
  // The 16/24-plane kernels resolve their carries every 32 rows and test there (k2_cobs.hip): the test was a quarter of their VALU
  // work at one test per group, and they run near their issue limits — same-box A/B tools/ab/r04_call13.sh: equal-width HiFi index
  // 3.88 -> 3.55 ms per 16 384 reads, genome search 5.73 -> 5.48 ms per 256 genomes with a test every 4th group alone.
  a.prune_every = 1;
  if (const char* e = getenv("KMCPG_PRUNE_EVERY")) {
    const int v = atoi(e);
    a.prune_every = (v == 2 || v == 4 || v == 8) ? v : 1;
  }
  a.split_min = n_long ? split_min : 0;
  // slot-major unit order: the waves in flight share one (block, tile) slice of the index, so the address range they gather
  // from is ~1/64 of the index (GTDB scale: 575 -> 510 ms per 524 k reads; profiles/r02_order_exp.txt)
  a.slot_major = getenv("KMCPG_SLOT_MAJOR") ? atoi(getenv("KMCPG_SLOT_MAJOR")) : 1;
  // tail mode of the 16/24-plane kernels on 1-KiB tiles (k2_cobs.hip): KMCPG_TAIL_SECTORS=0 switches it off
  // (2, not 4: a wave that enters with 3-4 live sectors carries near misses that would have died a little later through the rest of the
  // query, unpruned — same-box A/B on the genome search: 5.05 ms without, 4.74 with 2, 4.93 with 4; profiles/r06_tail_mode.txt)
  a.tail_sectors = getenv("KMCPG_TAIL_SECTORS") ? std::max(0, std::min(atoi(getenv("KMCPG_TAIL_SECTORS")), 4)) : 2;
  a.tail_min = getenv("KMCPG_TAIL_MIN") ? std::max(1, atoi(getenv("KMCPG_TAIL_MIN"))) : 64;
  if (db->profiling >= 2) {
    if (W.w_gathered.ensure((size_t)K2_GATHER_SLOTS * 16)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    HIPCHK(hipMemsetAsync(W.w_gathered.p, 0, (size_t)K2_GATHER_SLOTS * 16 * sizeof(uint64_t), st));
    a.gathered = (unsigned long long*)W.w_gathered.p;
  }
  tl_query_bound_n = 0;
  if (int rcb = fpr_bound(db, p.max_fpr, max_short, st, &a.cmin_fpr, &a.cmin_fpr_n)) return rcb;
  tl_query_bound_n = a.cmin_fpr ? a.cmin_fpr_n : 0;  // both kernel forms apply the table to every query of up to this many k-mers
  a.hits = d_hits;
  a.hit_cap = hit_cap;
  a.counter = (unsigned long long*)d_counters;
  // COBS kernels one batch at a time (the k-mer kernels above may have run beside the previous batch's)
  // (KMCPG_COBS_CHAIN=0, experiment with two kernel streams + two workspace slots: the next batch's COBS kernel may start in the previous
  // one's ragged end — nothing is shared between them but the read-only index; their HIP-event durations then overlap)
  static const bool cobs_chain = !(getenv("KMCPG_COBS_CHAIN") && atoi(getenv("KMCPG_COBS_CHAIN")) == 0);
  if (int rcc = chain_begin(&db->cobs_ev, cobs_chain && db->cobs_ev_valid, st)) return rcc;
  if (db->profiling) HIPCHK(hipEventRecord(pev[1], st));
  // Long queries on rows cut into a 64-lane tile form + one narrower form: both in one grid (k2_cobs_pair: the second form's workgroups
  // take the slots the first one's last waves free; KMCPG_PAIR=0: two launches, as before round 6)
  bool paired = false;
  if (npl >= 16 && db->classes.size() == 2 && db->classes[0].lpr == 64 && db->classes[1].lpr < 64 && !(getenv("KMCPG_PAIR") && atoi(getenv("KMCPG_PAIR")) == 0)) {
    K2Args a1 = a, b1 = a;
    a1.slots = db->classes[0].d_slots;
    a1.nslots = (uint32_t)db->classes[0].slots.size();
    b1.slots = db->classes[1].d_slots;
    b1.nslots = (uint32_t)db->classes[1].slots.size();
    paired = launch_k2_pair(a1, b1, db->classes[1].lpr, npl, st) == 0;
  }
  for (const auto& c : db->classes) {
    if (paired) break;
    a.slots = c.d_slots;
    a.nslots = (uint32_t)c.slots.size();
    if (launch_k2(a, c.lpr, npl, st) != 0) return kmcpg_fail(KMCPG_EINVAL, "batch too large for one launch: split it");
  }
  if (n_long) {
    a.ncols_total = (uint32_t)db->info.n_cols;
    // ~64 chunks for the largest query, 1024..8192 k-mers each (at most 8192: the chunk's counts fit 16 planes)
    uint32_t chk = 1024;
    while (chk < 8192 && (uint64_t)chk * 64 < long_meta[1]) chk <<= 1;
    if (const char* e = getenv("KMCPG_SPLIT_CHUNK")) chk = (uint32_t)std::max(64, std::min(atoi(e), 8192));
    a.split_chk = chk;
    a.split_chunks = (long_meta[1] + chk - 1) / chk;
    // count arrays of at most ~2 GB at a time
    const uint32_t group = (uint32_t)std::max<uint64_t>(1, (2ull << 30) / ((uint64_t)a.ncols_total * 4));
    if (W.w_long_counts.ensure((size_t)std::min<uint32_t>(group, n_long) * a.ncols_total)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    a.long_counts = W.w_long_counts.p;
    for (uint32_t g0 = 0; g0 < n_long; g0 += group) {
      a.long_list = W.w_long_list.p + g0;
      a.n_long = std::min<uint32_t>(group, n_long - g0);
      HIPCHK(hipMemsetAsync(a.long_counts, 0, (size_t)a.n_long * a.ncols_total * sizeof(uint32_t), st));
      for (const auto& c : db->classes) {
        a.slots = c.d_slots;
        a.nslots = (uint32_t)c.slots.size();
        if (launch_k2_split(a, c.lpr, st) != 0) return kmcpg_fail(KMCPG_EINVAL, "batch too large for one launch: split it");
      }
      launch_threshold_long(a, st);
    }
  }
  HIPCHK(hipEventRecord(db->cobs_ev, st));
  db->cobs_ev_valid = true;
  if (db->profiling) {
    HIPCHK(hipEventRecord(pev[2], st));
    db->ev_calls++;
  }
  if (int rc1 = wsg.finish()) return rc1;
  HIPCHK(hipGetLastError());
  return 0;
}

// K3: the device half of finalize (k3_finalize.hip) — group by read, -T, per-query order.  Enqueue only.
extern "C" int kmcpg_group_device(kmcpg_db* db, const kmcpg_hit* d_hits, const uint64_t* d_n_hits, uint64_t hit_cap, const int32_t* d_qkmers, uint32_t n_reads,
                                  const kmcpg_params* params, kmcpg_pair* d_pairs, uint64_t* d_read_offs, void* stream) {
  if (!db || !d_n_hits || !d_read_offs || (n_reads && !d_qkmers) || (hit_cap && (!d_hits || !d_pairs))) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (db->opts.device < 0 || !db->d_col_size) return kmcpg_fail(KMCPG_EDEVICE, "metadata-only handle (device -1): no GPU work possible");
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  const kmcpg_params p = params ? *params : default_params();
  hipStream_t st = (hipStream_t)stream;
  if (int rc0 = chain_begin(&db->fin_ev, db->fin_ev_valid, st)) return rc0;
  struct FinGuard {  // as WsGuard: every way out leaves the event behind
    kmcpg_db* db;
    hipStream_t st;
    ~FinGuard() {
      if (db->fin_ev && hipEventRecord(db->fin_ev, st) == hipSuccess) db->fin_ev_valid = true;
    }
  } fing{db, st};
  if (db->w_fin_cnt.ensure((size_t)n_reads + 2) || db->w_fin_sums.ensure((size_t)k3_scan_tiles_for(n_reads + 1) + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  // counters of the reads + the word that counts hits naming a read / column that does not exist
  HIPCHK(hipMemsetAsync(db->w_fin_cnt.p, 0, ((size_t)n_reads + 2) * sizeof(uint32_t), st));
  K3Args a{};
  a.hits = d_hits;
  a.n_hits = (const unsigned long long*)d_n_hits;
  a.hit_cap = hit_cap;
  a.nk = d_qkmers;
  a.n_reads = n_reads;
  a.n_cols = (uint32_t)db->col_meta.size();
  a.col_size = db->d_col_size;
  a.min_tcov = p.min_tcov;
  a.sort_mode = p.do_not_sort ? 3 : (p.sort_by == 1 ? 1 : (p.sort_by == 2 ? 2 : 0));
  a.cnt = db->w_fin_cnt.p;
  a.bad = db->w_fin_cnt.p + n_reads + 1;
  a.offs = d_read_offs;
  a.sums = db->w_fin_sums.p;
  a.pairs = d_pairs;
  if (n_reads == 0) HIPCHK(hipMemsetAsync(d_read_offs, 0, 2 * sizeof(uint64_t), st));
  else {
    launch_k3(a, 0, st);
    // d_read_offs[n_reads + 1] = the bad-hit count (a 32-bit word widened on the device side of the copy: two words cleared first)
    HIPCHK(hipMemsetAsync(d_read_offs + n_reads + 1, 0, sizeof(uint64_t), st));
    HIPCHK(hipMemcpyAsync(d_read_offs + n_reads + 1, a.bad, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int kmcpg_set_profiling(kmcpg_db* db, int enable) {
  if (!db) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  db->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  db->ev_calls = 0;
  return 0;
}

// word 0 of every counter slot: 16-byte row loads; word 1: 8-byte hash loads; word 2: waves that finished in tail mode
static int read_gather_slots(kmcpg_db* db, int word, uint64_t unit, uint64_t* bytes) {
  if (!db || !bytes) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  kmcpg_db::Workspace& W = db->ws[db->ws_last];
  if (db->profiling < 2 || !W.w_gathered.p || db->ev_calls == 0) return kmcpg_fail(KMCPG_EINVAL, "no kmcpg_query_device call at profiling level 2 yet");
  KMCPG_USE_DEVICE(db);
  hipEvent_t* pev = db->ev + 4 * ((db->ev_calls - 1) % 4);
  HIPCHK(hipEventSynchronize(pev[2]));
  std::vector<uint64_t> slots((size_t)K2_GATHER_SLOTS * 16);
  HIPCHK(hipMemcpy(slots.data(), W.w_gathered.p, slots.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  uint64_t n = 0;
  for (int i = 0; i < K2_GATHER_SLOTS; i++) n += slots[(size_t)i * 16 + (size_t)word];
  *bytes = n * unit;
  return 0;
}

extern "C" int kmcpg_last_gathered_bytes(kmcpg_db* db, uint64_t* bytes) { return read_gather_slots(db, 0, 16, bytes); }
extern "C" int kmcpg_last_hash_bytes(kmcpg_db* db, uint64_t* bytes) { return read_gather_slots(db, 1, 8, bytes); }
extern "C" int kmcpg_last_tail_waves(kmcpg_db* db, uint64_t* waves) { return read_gather_slots(db, 2, 1, waves); }

extern "C" int kmcpg_timing_at(kmcpg_db* db, uint32_t age, float* kmers_ms, float* cobs_ms) {
  if (!db) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  if (!db->profiling || age >= 4 || db->ev_calls <= age) return kmcpg_fail(KMCPG_EINVAL, "no profiled kmcpg_query_device call of that age (the last 4 are kept)");
  KMCPG_USE_DEVICE(db);
  hipEvent_t* pev = db->ev + 4 * ((db->ev_calls - 1 - age) % 4);
  HIPCHK(hipEventSynchronize(pev[2]));
  float a = 0, b = 0;
  HIPCHK(hipEventElapsedTime(&a, pev[0], pev[3]));
  HIPCHK(hipEventElapsedTime(&b, pev[1], pev[2]));
  if (kmers_ms) *kmers_ms = a;
  if (cobs_ms) *cobs_ms = b;
  return 0;
}

extern "C" int kmcpg_last_timing(kmcpg_db* db, float* kmers_ms, float* cobs_ms) { return kmcpg_timing_at(db, 0, kmers_ms, cobs_ms); }

extern "C" int kmcpg_plant_reads_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, uint32_t n_reads, uint64_t total_bases,
                                        uint32_t max_read_len, const uint32_t* d_cols, void* stream) {
  if (!db || !d_seqs || !d_offs || !d_cols) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  hipStream_t st = (hipStream_t)stream;
  kmcpg_db::Workspace& W = db->ws[0];
  if (int rc0 = ws_begin(W, st)) return rc0;
  WsGuard wsg{W, st, st};
  if (W.w_hashes.ensure(total_bases + 1) || W.w_nk_raw.ensure(n_reads + 1) || W.w_nk1.ensure(n_reads + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  DevBuf<int32_t> tmp;
  if (tmp.ensure(2 * (size_t)n_reads + 2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  kmcpg_params p = default_params();
  p.min_qlen = 0;
  p.min_matched = 1;
  p.dedup_threshold = 0x7fffffff;  // plant every k-mer occurrence (idempotent)
  uint64_t maxn = 0;
  if ((db->info.syncmer || db->info.minimizer) && W.w_scratch.ensure(2 * total_bases + 2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  int rc = run_kmers(db, W, d_seqs, d_offs, nullptr, nullptr, n_reads, max_read_len, p, W.w_hashes.p, W.w_scratch.p, total_bases + 1, W.w_nk_raw.p,
                     W.w_nk1.p, tmp.p, tmp.p + n_reads + 1, st, &maxn);
  if (rc == 0)
    launch_plant_reads(db->d_blockdev, (uint32_t)db->h_blockdev.size(), db->info.num_hashes, W.w_hashes.p, d_offs, W.w_nk_raw.p, d_cols, n_reads, st);
  hipError_t e = hipStreamSynchronize(st);
  tmp.release();
  if (rc) return rc;
  if (e != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "plant kernel failed: %s", hipGetErrorString(e));
  return 0;
}

