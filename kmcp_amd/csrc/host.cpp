// host.cpp — the whole per-batch pipeline on host buffers (kmcpg_search_batch, --try-se), the in-process multi-GPU handle,
// and the bench/parity helpers.  Reference counterparts: handleQuery (util-db-search.go:763-1025), :831-850, :1001-1014.
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
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "dbformat.hpp"
#include "engine.hpp"
#include "exchange.hpp"
#include "fpr.hpp"
#include "kernels.hpp"
#include "pack2.hpp"

using namespace kmcpg;

// ------------------------------------------------------------------------------------------------
// whole pipeline on host buffers: kmcpg_submit / kmcpg_wait, kmcpg_search_batch = submit + wait
//
// A handle owns a private non-blocking stream and a few LANES; a lane is the staging of one batch in flight (pinned host
// input and output buffers, device input and output buffers, a completion event).  kmcpg_submit copies the caller's batch
// into a free lane, enqueues H2D + K1 + K2 + D2H on the stream and returns; kmcpg_wait waits for the lane's event and runs
// the host half (float64 thresholds, FPR, sort) on the calling thread, so the GPU works on the next batches meanwhile.
// The number of lanes bounds the batches in flight the way the reference's token channel bounds its queries
// (util-db-search.go:243, :347-351).  Retries (--try-se, smaller k of a multi-k database) are rare, small and synchronous;
// they run on a lane of their own so that they never wait for a lane another waiter holds.
// ------------------------------------------------------------------------------------------------
namespace kmcpg {

template <typename T>
struct PinBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 8 + 64;
    if (hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocPortable) != hipSuccess) return -1;
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Lane {
  PinBuf<uint8_t> h_seqs, h_seqs2;
  PinBuf<uint64_t> h_offs, h_offs2, h_cnt;
  PinBuf<int32_t> h_qk, h_ql;
  PinBuf<kmcpg_hit> h_hits;
  DevBuf<uint8_t> d_seqs, d_seqs2;
  DevBuf<uint64_t> d_offs, d_offs2, d_cnt;
  DevBuf<int32_t> d_qk, d_ql;
  DevBuf<kmcpg_hit> d_hits;
  // K3 (device half of finalize): the batch's matches grouped by read and in final order, 8 bytes each + CSR offsets of the reads
  DevBuf<kmcpg_pair> d_pairs;
  DevBuf<uint64_t> d_roffs;
  PinBuf<kmcpg_pair> h_pairs;
  PinBuf<uint64_t> h_roffs;
  // 2-bit packed upload (stage): the batch's bases as codes + the runs of foreign bytes instead of h_seqs; unpacked on the device
  PinBuf<uint8_t> h_pack;
  PinBuf<ExcRun> h_exc;
  DevBuf<uint8_t> d_pack;
  DevBuf<ExcRun> d_exc;
  bool packed = false;
  uint32_t n_exc = 0;
  const uint8_t* ext_pack = nullptr;  // codes that live in the CALLER's pinned memory (kmcpg_host_alloc): uploaded from there, no staging copy
  int32_t bound_n = 0;   // the -f bound of the LAST kmcpg_query_device call for this batch covered queries of up to this many k-mers
  bool grouped = false;  // this batch went through K3: h_pairs / h_roffs hold its result, h_hits is not filled
  hipEvent_t k3_ev = nullptr, eager_ev = nullptr;  // K3 done on the kernel stream -> the eager copy of the pairs on the copy stream
  bool eager_aside = false;                        // ... when it was put there (pieces of a large kmcpg_search_batch call)
  hipEvent_t done = nullptr, uploaded = nullptr;
  hipStream_t st = nullptr;  // the kernel stream this batch was enqueued on (AsyncState::stream or stream2, taken in turn)
  uint32_t n = 0;
  bool paired = false;
  uint64_t tb1 = 0, tb2 = 0;
  uint32_t maxlen = 0;
  uint64_t copied = 0;  // hits already on their way to h_hits when `done` fires
  bool busy = false;
  void release() {
    h_seqs.release(); h_seqs2.release(); h_offs.release(); h_offs2.release(); h_cnt.release(); h_qk.release(); h_ql.release(); h_hits.release();
    d_seqs.release(); d_seqs2.release(); d_offs.release(); d_offs2.release(); d_cnt.release(); d_qk.release(); d_ql.release(); d_hits.release();
    d_pairs.release(); d_roffs.release(); h_pairs.release(); h_roffs.release();
    h_pack.release(); h_exc.release(); d_pack.release(); d_exc.release();
    if (done) (void)hipEventDestroy(done);
    if (uploaded) (void)hipEventDestroy(uploaded);
    if (k3_ev) (void)hipEventDestroy(k3_ev);
    if (eager_ev) (void)hipEventDestroy(eager_ev);
    done = uploaded = k3_ev = eager_ev = nullptr;
  }
};

struct AsyncState {
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // late D2H of a finished batch's hits: must not queue behind the kernels of later batches
  // (Exactly three streams per handle beside the null stream: the runtime multiplexes HIP streams onto 4 hardware queues, and a
  // fifth stream — tried for the pieces' eager copies — made uploads and kernels share one: -8 % on the pipelined path, same-box A/B.)
  hipStream_t up_stream = nullptr;    // H2D of a batch's reads, beside the kernels of the batches before it
  // a second kernel stream: consecutive batches alternate between the two, so that the k-mer kernels of one run beside the COBS
  // kernels of the one before it (with KMCPG_WS_SLOTS=2 the handle keeps two k-mer workspaces, engine.hpp).  Off unless KMCPG_KSTREAMS=2: measured a loss on
  // four of five workloads (profiles/r05_k1_beside_k2.txt)
  hipStream_t stream2 = nullptr;
  int kstreams = 0;  // KMCPG_KSTREAMS: 2 = consecutive batches alternate between the two kernel streams, 1 = never, 0 (unset) = batches of whole genomes only
  uint64_t enqueued = 0;
  std::atomic<uint64_t> hits_hint{0};  // hits per 1024 reads seen lately: sizes the hit buffers and the eager D2H of the next batches
  uint64_t lane_hit_budget = 0;        // entries a lane's device hit buffer may grow to beyond the plain size (from free HBM at first use)
  bool hits_stay_on_device = false;    // shard of a handle that gathers the hit lists over RCCL: no per-shard D2H of hits
  bool device_finalize = true;         // K3 after K2: grouping, -T and the per-query order on the GPU (KMCPG_DEVICE_FINALIZE=0: host)
  std::vector<std::unique_ptr<Lane>> lanes;
  size_t max_lanes = 4;
  std::mutex mu;
  std::condition_variable cv;
  Lane retry;
  std::mutex retry_mu;
};

// batches between kmcpg_submit and the end of kmcpg_wait on this handle (or its shards)
int async_in_flight(kmcpg_db* db) {
  int n = 0;
  for (kmcpg_db* sh : db->shards) n += async_in_flight(sh);
  if (!db->async) return n;
  std::lock_guard<std::mutex> g(db->async->mu);
  for (auto& l : db->async->lanes) n += l->busy ? 1 : 0;
  return n;
}

void async_release(kmcpg_db* db) {
  if (!db->async) return;
  if (db->opts.device >= 0) (void)hipSetDevice(db->opts.device);
  if (db->async->up_stream) (void)hipStreamSynchronize(db->async->up_stream);
  if (db->async->stream) (void)hipStreamSynchronize(db->async->stream);
  if (db->async->stream2) (void)hipStreamSynchronize(db->async->stream2);
  for (auto& l : db->async->lanes) l->release();
  db->async->retry.release();
  if (db->async->stream) (void)hipStreamDestroy(db->async->stream);
  if (db->async->stream2) (void)hipStreamDestroy(db->async->stream2);
  if (db->async->copy_stream) (void)hipStreamDestroy(db->async->copy_stream);
  if (db->async->up_stream) (void)hipStreamDestroy(db->async->up_stream);
  delete db->async;
  db->async = nullptr;
}

}  // namespace kmcpg

// one batch in flight; single-device handles have one part, kmcpg_open_devices handles one part per GPU
struct kmcpg_ticket {
  kmcpg_db* db = nullptr;
  struct Part {
    kmcpg_db* shard;
    Lane* lane;
    bool retry;
  };
  std::vector<Part> parts;
  uint32_t n = 0;
  bool paired = false;
  kmcpg_params p{};
  // the batch as staged (retries re-read it): the first lane's pinned copy, or — paged handles — the ticket's own
  const uint8_t* S[2] = {nullptr, nullptr};
  const uint64_t* O[2] = {nullptr, nullptr};
  // paged handles (kmcpg_open_paged): the search ran inside kmcpg_submit, one resident shard after the other; results wait here
  bool paged = false;
  std::vector<uint8_t> seqs, seqs2;
  std::vector<uint64_t> offs, offs2;
  std::vector<kmcpg_hit, NoInitAlloc<kmcpg_hit>> hits;
  std::vector<int32_t> qk, ql;
};

namespace {

int async_state(kmcpg_db* db, AsyncState** out) {
  static std::mutex init_mu;
  std::lock_guard<std::mutex> g(init_mu);
  if (!db->async) {
    std::unique_ptr<AsyncState> a(new AsyncState());
    HIPCHK(hipSetDevice(db->opts.device));
    HIPCHK(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    // the copies back run at the highest stream priority: where a copy is done by a blit kernel it must not queue behind the
    // chip-filling kernels of the next batch (same-box A/B: the eager copy at default priority cost the pipelined path 10 %)
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    HIPCHK(hipStreamCreateWithPriority(&a->copy_stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(hipStreamCreateWithFlags(&a->up_stream, hipStreamNonBlocking));
    if (getenv("KMCPG_KSTREAMS")) a->kstreams = atoi(getenv("KMCPG_KSTREAMS")) >= 2 ? 2 : 1;  // 2: every batch, 1: never; unset: batches of whole genomes
    if (a->kstreams == 2) HIPCHK(hipStreamCreateWithFlags(&a->stream2, hipStreamNonBlocking));
    if (const char* e = getenv("KMCPG_INFLIGHT")) a->max_lanes = (size_t)std::max(1, std::min(atoi(e), 16));
    if (const char* e = getenv("KMCPG_DEVICE_FINALIZE")) a->device_finalize = atoi(e) != 0;
    // Hit buffers follow the data (a database full of close relatives returns hundreds of hits per read) but must never crowd
    // out the k-mer workspace next to a large index: all lanes together may take a quarter of what is free now (the index is
    // resident already), at most 16 GB.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    uint64_t budget = std::min<uint64_t>(free_b / 4, 16ull << 30);
    if (const char* e = getenv("KMCPG_HIT_BUDGET_MB")) budget = (uint64_t)std::max(0ll, atoll(e)) << 20;
    a->lane_hit_budget = budget / (sizeof(kmcpg_hit) + sizeof(kmcpg_pair)) / (a->max_lanes + 1);  // a hit entry has its K3 pair beside it (d_pairs)
    db->async = a.release();
  }
  *out = db->async;
  return 0;
}

// block = false: nullptr when every lane is in flight (kmcpg_submit never blocks: a caller that submits from one thread
// would otherwise wait for itself); block = true: wait for a lane (kmcpg_search_batch: whoever holds a lane through it gives
// it back without needing another one)
Lane* acquire_lane(AsyncState* A, bool retry, bool block) {
  if (retry) {
    A->retry_mu.lock();
    return &A->retry;
  }
  std::unique_lock<std::mutex> g(A->mu);
  for (;;) {
    for (auto& l : A->lanes)
      if (!l->busy) {
        l->busy = true;
        return l.get();
      }
    if (A->lanes.size() < A->max_lanes) {
      A->lanes.emplace_back(new Lane());
      A->lanes.back()->busy = true;
      return A->lanes.back().get();
    }
    if (!block) return nullptr;
    A->cv.wait(g);
  }
}

void release_lane(AsyncState* A, Lane* l, bool retry) {
  if (retry) {
    A->retry_mu.unlock();
    return;
  }
  {
    std::lock_guard<std::mutex> g(A->mu);
    l->busy = false;
  }
  A->cv.notify_one();
}

void par_memcpy(void* dst, const void* src, size_t n) {
  const size_t kMin = 8u << 20;
  if (n < 2 * kMin) {
    memcpy(dst, src, n);
    return;
  }
  const size_t T = std::min<size_t>(4, n / kMin);
  std::vector<std::thread> th;
  for (size_t t = 1; t < T; t++) th.emplace_back([=] { memcpy((char*)dst + n * t / T, (const char*)src + n * t / T, n * (t + 1) / T - n * t / T); });
  memcpy(dst, src, n / T);
  for (auto& t : th) t.join();
}

// caller's batch -> the lane's pinned buffers (the caller may reuse its buffers as soon as kmcpg_submit returns)
// Long-query batches go up as 2-bit codes (pack2.hpp): a batch of 256 genomes is 1 GB of ASCII, and at ~35 k genomes/s of kernel
// rate the 40-50 GB/s of PCIe were what bounded the host-to-host rate (9.9 k genomes/s, round 4).  The pack IS the staging copy —
// it reads the caller's buffer once and writes a quarter of it to pinned memory — and the device expands it again in ~0.3 ms per
// GB (k_unpack2); the k-mer kernels read ASCII as before.  Only where nothing re-reads the staged ASCII (retries of --try-se and
// of multi-k databases do: `allow_pack` is false there) and only for batches whose bases dominate the upload (mean query >= 1 kb:
// a batch of 150-bp reads is 150 MB and its staging copy is not what limits it).  KMCPG_PACK=0 switches it off.
bool pack_wanted(uint64_t total_bases, uint32_t n) {
  static const int env = getenv("KMCPG_PACK") ? atoi(getenv("KMCPG_PACK")) : -1;  // 0 never, 1 always (tests), default by shape
  if (env == 0) return false;
  if (env == 1) return total_bases >= 64;
  return total_bases >= (8ull << 20) && total_bases / std::max<uint32_t>(1, n) >= 1000;
}

int stage(Lane* L, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n, bool allow_pack = false) {
  L->n = n;
  L->packed = false;
  L->n_exc = 0;
  L->paired = seqs2 != nullptr;
  L->tb1 = L->tb2 = 0;
  L->maxlen = 0;
  L->copied = 0;
  if (n == 0) return 0;
  if (offs[0] != 0 || (seqs2 && offs2[0] != 0)) return kmcpg_fail(KMCPG_EINVAL, "offs[0] must be 0");
  uint64_t maxlen = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (offs[i + 1] < offs[i] || (seqs2 && offs2[i + 1] < offs2[i])) return kmcpg_fail(KMCPG_EINVAL, "offsets must not decrease (read %u)", i);
    maxlen = std::max(maxlen, offs[i + 1] - offs[i]);
    if (seqs2) maxlen = std::max(maxlen, offs2[i + 1] - offs2[i]);
  }
  if (maxlen > 0x7fffffffULL) return kmcpg_fail(KMCPG_EUNSUPPORTED, "query longer than 2^31-1 bases");
  L->maxlen = (uint32_t)maxlen;
  L->tb1 = offs[n];
  L->tb2 = seqs2 ? offs2[n] : 0;
  if (L->h_offs.ensure((size_t)n + 1) || (seqs2 && (L->h_seqs2.ensure(L->tb2 + 16) || L->h_offs2.ensure((size_t)n + 1))))
    return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
  if (allow_pack && !seqs2 && pack_wanted(L->tb1, n)) {
    static thread_local std::vector<PackRun> exc;
    static_assert(sizeof(PackRun) == sizeof(ExcRun), "one layout");
    if (L->h_pack.ensure(L->tb1 / 4 + 16)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
    static const unsigned pack_threads = getenv("KMCPG_PACK_THREADS") ? (unsigned)std::max(1, std::min(atoi(getenv("KMCPG_PACK_THREADS")), 64))
                                                                      : std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    // more than one foreign run per 256 bases: not nucleotide text, sent as it is
    if (pack2_parallel(seqs, L->tb1, L->h_pack.p, exc, std::max<size_t>(1024, L->tb1 / 256), pack_threads)) {
      if (L->h_exc.ensure(exc.size() + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
      if (!exc.empty()) memcpy(L->h_exc.p, exc.data(), exc.size() * sizeof(ExcRun));
      L->n_exc = (uint32_t)exc.size();
      L->packed = true;
    }
  }
  if (!L->packed) {
    if (L->h_seqs.ensure(L->tb1 + 16)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
    par_memcpy(L->h_seqs.p, seqs, L->tb1);
  }
  par_memcpy(L->h_offs.p, offs, ((size_t)n + 1) * sizeof(uint64_t));
  if (seqs2) {
    par_memcpy(L->h_seqs2.p, seqs2, L->tb2);
    par_memcpy(L->h_offs2.p, offs2, ((size_t)n + 1) * sizeof(uint64_t));
  }
  return 0;
}

// A batch that arrives as 2-bit codes + exception runs (kmcpg_submit_packed): the staging copy is a quarter of the text's; from the
// upload on the lane looks exactly like one whose text stage() packed itself.
struct PackedIn {
  const uint8_t* codes;
  const kmcpg_exc_run* exc;
  uint64_t n_exc;
};

int stage_packed(Lane* L, const PackedIn& in, const uint64_t* offs, uint32_t n) {
  static_assert(sizeof(kmcpg_exc_run) == sizeof(ExcRun), "one layout");
  L->n = n;
  L->packed = false;
  L->n_exc = 0;
  L->ext_pack = nullptr;
  L->paired = false;
  L->tb1 = L->tb2 = 0;
  L->maxlen = 0;
  L->copied = 0;
  if (n == 0) return 0;
  if (offs[0] != 0) return kmcpg_fail(KMCPG_EINVAL, "offs[0] must be 0");
  uint64_t maxlen = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (offs[i + 1] < offs[i]) return kmcpg_fail(KMCPG_EINVAL, "offsets must not decrease (read %u)", i);
    maxlen = std::max(maxlen, offs[i + 1] - offs[i]);
  }
  if (maxlen > 0x7fffffffULL) return kmcpg_fail(KMCPG_EUNSUPPORTED, "query longer than 2^31-1 bases");
  const uint64_t tb = offs[n];
  if (in.n_exc > 0xffffffffULL) return kmcpg_fail(KMCPG_EINVAL, "too many exception runs");
  for (uint64_t i = 0; i < in.n_exc; i++) {  // the device writes these runs into the batch's text: they must lie inside it
    const kmcpg_exc_run& e = in.exc[i];
    if (e.pos > tb || e.len > tb - e.pos || e.byte > 255) return kmcpg_fail(KMCPG_EINVAL, "exception run %llu lies outside the batch", (unsigned long long)i);
  }
  L->maxlen = (uint32_t)maxlen;
  L->tb1 = tb;
  // Codes in memory from kmcpg_host_alloc are page-locked already: the DMA engine reads them where they are (the caller keeps them untouched
  // until kmcpg_wait has returned).  Staging 256 MB — a batch of 256 assemblies — took the submitting thread 15-25 ms, longer than the GPU
  // needs for the batch (profiles/r06_h2h.txt); anything else is copied to the lane's pinned buffer as before.
  bool pinned = false;
  {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, in.codes) == hipSuccess) pinned = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();  // ordinary memory: not an error
  }
  if (L->h_offs.ensure((size_t)n + 1) || (!pinned && L->h_pack.ensure(tb / 4 + 16)) || L->h_exc.ensure(in.n_exc + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
  if (pinned) L->ext_pack = in.codes;
  else par_memcpy(L->h_pack.p, in.codes, (tb + 3) / 4);
  if (in.n_exc) memcpy(L->h_exc.p, in.exc, in.n_exc * sizeof(ExcRun));
  par_memcpy(L->h_offs.p, offs, ((size_t)n + 1) * sizeof(uint64_t));
  L->n_exc = (uint32_t)in.n_exc;
  L->packed = true;
  return 0;
}

int enqueue_query(kmcpg_db* db, AsyncState* A, Lane* L, const kmcpg_params& p, const std::function<int()>* prologue = nullptr) {
  hipStream_t st = L->st;
  // packed input: the k-mer stage of this call (query.cpp run_kmers, on this thread) reads the codes where it can — whole genomes — and
  // expands them to the text in d_seqs where it cannot; a rerun of the batch (ENOMEM, hit-buffer overflow) does the same again
  struct PackedScope {
    explicit PackedScope(Lane* L) {
      if (!L->packed) return;
      PackedSrc s;
      s.codes = L->d_pack.p;
      s.exc = L->n_exc ? L->d_exc.p : nullptr;
      s.n_exc = L->n_exc;
      s.text = L->d_seqs.p;
      s.n_bases = L->tb1;
      tl_packed_src = s;
    }
    ~PackedScope() { tl_packed_src = PackedSrc{}; }
  } packed_scope(L);
  int rc = query_device_after(db, L->d_seqs.p, L->d_offs.p, L->paired ? L->d_seqs2.p : nullptr, L->paired ? L->d_offs2.p : nullptr, L->n, L->tb1 + L->tb2,
                              L->maxlen, &p, L->d_hits.p, L->d_hits.cap, L->d_cnt.p, L->d_qk.p, L->d_ql.p, st, prologue);
  if (rc) return rc;
  L->bound_n = tl_query_bound_n;
  HIPCHK(hipMemcpyAsync(L->h_cnt.p, L->d_cnt.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  if (L->grouped) {  // K3 right behind K2 (an overflowing hit buffer makes both run again, collect())
    if (L->d_pairs.ensure(L->d_hits.cap) || L->d_roffs.ensure((size_t)L->n + 2) || L->h_roffs.ensure((size_t)L->n + 2)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    rc = kmcpg_group_device(db, L->d_hits.p, L->d_cnt.p, L->d_hits.cap, L->d_qk.p, L->n, &p, L->d_pairs.p, L->d_roffs.p, st);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(L->h_roffs.p, L->d_roffs.p, ((size_t)L->n + 2) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  }
  return 0;
}

// H2D + K1 + K2 (+ K3) + D2H of one staged lane on the handle's streams; returns without waiting.  generous_eager: a piece of a
// large kmcpg_search_batch call — its neighbours of the same batch have just shown how many matches to expect, so the eager copy
// may take all of them (it runs on the copy stream, beside the next piece's kernels)
int enqueue(kmcpg_db* db, AsyncState* A, Lane* L, const kmcpg_params& p, bool generous_eager = false) {
  KMCPG_USE_DEVICE(db);
  if (!L->done) HIPCHK(hipEventCreateWithFlags(&L->done, hipEventDisableTiming));
  if (!L->uploaded) HIPCHK(hipEventCreateWithFlags(&L->uploaded, hipEventDisableTiming));
  const uint32_t n = L->n;
  if (n == 0) return 0;
  {
    // two kernel streams taken in turn: for every batch (KMCPG_KSTREAMS=2) or, by default, for batches of whole genomes — their k-mer kernel runs
    // beside the COBS kernel of the batch before (query.cpp pick_slot gives such a batch the second workspace); the stream is made when first needed
    std::lock_guard<std::mutex> g(A->mu);
    const bool two = A->kstreams == 2 || (A->kstreams == 0 && whole_genome_batch(db, L->maxlen, L->paired));
    if (two && !A->stream2) HIPCHK(hipStreamCreateWithFlags(&A->stream2, hipStreamNonBlocking));
    L->st = (two && (A->enqueued++ & 1)) ? A->stream2 : A->stream;
  }
  hipStream_t st = L->st;
  // ~1 hit per read is typical for distinct references, dozens to hundreds for a database full of close relatives: the
  // buffers follow what the last large batches produced (hits_hint = hits per 1024 reads).  Too small a device buffer costs
  // a rerun of the batch (collect), too short an eager copy a late one on the copy stream, so the device buffer is generous
  // (up to 512 hits = 6 KB per read, and never smaller than it already is) while the eager copy stays within 32 hits per
  // read: a burst of hit-heavy queries must not make every later batch drag gigabytes over PCIe.
  const uint64_t expect = std::min<uint64_t>(A->hits_hint.load() * (uint64_t)n / 1024, (uint64_t)n * 512);
  const uint64_t base_cap = (uint64_t)n * 8 + 1024;
  // what this batch should have: the recent hit rate + 50 %, within the lane's share of the handle's hit-buffer budget
  const uint64_t want = std::max<uint64_t>(base_cap, std::min<uint64_t>(expect + expect / 2, std::max<uint64_t>(base_cap, A->lane_hit_budget)));
  // a buffer left over from a burst of hit-heavy batches goes back once the data has calmed down (hits_hint decays by 1/8 per
  // batch): HBM pinned for the life of the handle is HBM the workspace of a later, larger batch may need
  // (16x, not 4x: a lane that serves whole batches and quarter-batch pieces in turn must not free and reallocate its buffer every time —
  // hipFree waits for the device)
  // (d_pairs is sized by d_hits.cap and re-ensured by enqueue_query: it goes wherever d_hits goes)
  if (L->d_hits.cap > 16 * want && L->d_hits.cap > (1u << 20)) {
    L->d_hits.release();
    L->d_pairs.release();
  }
  uint64_t cap = std::max<uint64_t>(L->d_hits.cap, want);
  if (L->d_seqs.ensure(L->tb1 + 16) || L->d_offs.ensure((size_t)n + 1) || L->d_cnt.ensure(2) || L->d_qk.ensure(n) || L->d_ql.ensure(n) ||
      (L->paired && (L->d_seqs2.ensure(L->tb2 + 16) || L->d_offs2.ensure((size_t)n + 1))))
    return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  if (L->d_hits.ensure(cap)) {  // no room for the generous size next to the index: the plain one, and a rerun if it overflows
    (void)hipGetLastError();
    cap = base_cap;
    if (L->d_hits.ensure(cap)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  }
  const uint64_t first = std::min<uint64_t>(cap, std::max<uint64_t>((uint64_t)n * 2 + 1024, std::min<uint64_t>(expect + expect / 4, (uint64_t)n * (generous_eager ? 1024 : 32))));
  // K3 only where this handle holds the whole database: the lists of shards (several GPUs, passes of a paged handle) are merged first
  L->grouped = A->device_finalize && !A->hits_stay_on_device && db->opts.shard_count == 1;
  if (L->h_cnt.ensure(2) || L->h_qk.ensure(n) || L->h_ql.ensure(n) || (L->grouped ? L->h_pairs.ensure(first) : L->h_hits.ensure(first)))
    return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
  // the reads go up on a stream of their own (150 MB per million reads: 4 ms of PCIe that would otherwise sit between the
  // kernels of consecutive batches); the lane's device buffers are idle, its previous batch was waited for
  hipStream_t up = A->up_stream;
  if (L->packed) {
    if (L->d_pack.ensure(L->tb1 / 4 + 16) || (L->n_exc && L->d_exc.ensure(L->n_exc))) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    HIPCHK(hipMemcpyAsync(L->d_pack.p, L->ext_pack ? L->ext_pack : L->h_pack.p, (L->tb1 + 3) / 4, hipMemcpyHostToDevice, up));
    if (L->n_exc) HIPCHK(hipMemcpyAsync(L->d_exc.p, L->h_exc.p, (size_t)L->n_exc * sizeof(ExcRun), hipMemcpyHostToDevice, up));
  } else if (L->tb1) {
    HIPCHK(hipMemcpyAsync(L->d_seqs.p, L->h_seqs.p, L->tb1, hipMemcpyHostToDevice, up));
  }
  HIPCHK(hipMemcpyAsync(L->d_offs.p, L->h_offs.p, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
  if (L->paired) {
    if (L->tb2) HIPCHK(hipMemcpyAsync(L->d_seqs2.p, L->h_seqs2.p, L->tb2, hipMemcpyHostToDevice, up));
    HIPCHK(hipMemcpyAsync(L->d_offs2.p, L->h_offs2.p, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
  }
  HIPCHK(hipEventRecord(L->uploaded, up));
  // the kernel stream waits for the upload (and expands packed input) in front of THIS batch's first kernel, under the handle's enqueue
  // lock: see query.cpp query_device_after
  const std::function<int()> after_upload = [&]() -> int {
    HIPCHK(hipStreamWaitEvent(st, L->uploaded, 0));
    return 0;
  };
  int rc = enqueue_query(db, A, L, p, &after_upload);
  if (rc == KMCPG_ENOMEM) {
    // the shared workspace (hashes, dedup scratch, long-query counters) did not fit: hit buffers above the plain size are the
    // one thing on this handle that can give memory back — those of the idle lanes and this lane's own — then once more
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> g(A->mu);
      for (auto& l : A->lanes)
        if (!l->busy && l.get() != L && l->d_hits.cap > 0) {
          l->d_hits.release();
          l->d_pairs.release();
        }
    }
    if (L->d_hits.cap > base_cap + base_cap / 8 + 64) {
      HIPCHK(hipStreamSynchronize(st));  // kernels of the failed attempt may have been enqueued with the old buffer
      L->d_hits.release();
      L->d_pairs.release();
      if (L->d_hits.ensure(base_cap)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    }
    rc = enqueue_query(db, A, L, p);
  }
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(L->h_qk.p, L->d_qk.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(L->h_ql.p, L->d_ql.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  L->copied = A->hits_stay_on_device ? 0 : std::min<uint64_t>(first, L->d_hits.cap);
  L->eager_aside = false;
  if (L->copied && L->grouped && !generous_eager) {
    // a small copy (<= 32 pairs per read), in stream order before the lane's completion event (same-box A/B: on a stream of its own it
    // cost the pipelined submit / wait path 8 %)
    HIPCHK(hipMemcpyAsync(L->h_pairs.p, L->d_pairs.p, L->copied * sizeof(kmcpg_pair), hipMemcpyDeviceToHost, st));
  } else if (L->copied && L->grouped) {
    // a piece of a large kmcpg_search_batch call: all of its ordered pairs leave on the copy stream, behind K3 only, while the
    // next piece's kernels run
    L->eager_aside = true;
    if (!L->k3_ev) HIPCHK(hipEventCreateWithFlags(&L->k3_ev, hipEventDisableTiming));
    if (!L->eager_ev) HIPCHK(hipEventCreateWithFlags(&L->eager_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(L->k3_ev, st));
    HIPCHK(hipStreamWaitEvent(A->copy_stream, L->k3_ev, 0));
    HIPCHK(hipMemcpyAsync(L->h_pairs.p, L->d_pairs.p, L->copied * sizeof(kmcpg_pair), hipMemcpyDeviceToHost, A->copy_stream));
    HIPCHK(hipEventRecord(L->eager_ev, A->copy_stream));
  } else if (L->copied) HIPCHK(hipMemcpyAsync(L->h_hits.p, L->d_hits.p, L->copied * sizeof(kmcpg_hit), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(L->done, st));
  return 0;
}

// waits for the lane; afterwards h_hits[0..*n_hits), h_qk, h_ql are complete (fetch = false: the hits stay in d_hits[0..*n_hits),
// the caller gathers them on the device)
int collect(kmcpg_db* db, AsyncState* A, Lane* L, const kmcpg_params& p, uint64_t* n_hits, bool fetch = true) {
  *n_hits = 0;
  if (L->n == 0) return 0;
  KMCPG_USE_DEVICE(db);
  HIPCHK(hipEventSynchronize(L->done));
  uint64_t cnt = L->h_cnt.p[0];
  for (int attempt = 0; cnt > L->d_hits.cap; attempt++) {  // the hit buffer was too small: rerun with room for every hit
    if (attempt == 2) return kmcpg_fail(KMCPG_ENOMEM, "hit buffer overflow");
    HIPCHK(hipStreamSynchronize(L->st));
    HIPCHK(hipStreamSynchronize(A->copy_stream));  // (the eager copy of the attempt that overflowed)
    if (L->d_hits.ensure(cnt + cnt / 4)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    int rc = enqueue_query(db, A, L, p);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(L->st));
    cnt = L->h_cnt.p[0];
    L->copied = 0;
  }
  // the k-mer count the counter planes were sized for (the longest read) bounds every query's NumKmers
  const uint64_t bound = (uint64_t)L->maxlen * (L->paired ? 2 : 1);
  if (L->h_cnt.p[1] > bound) return kmcpg_fail(KMCPG_EDEVICE, "internal: a query reported %llu k-mers, more than its length allows", (unsigned long long)L->h_cnt.p[1]);
  if (fetch && L->grouped) {
    // K3 ran behind K2: what comes to the host is its output — offsets (already here) and the pairs that passed -T, in final order
    if (L->copied && L->eager_aside) HIPCHK(hipEventSynchronize(L->eager_ev));
    const uint64_t kept = L->h_roffs.p[L->n];
    if (kept > cnt) return kmcpg_fail(KMCPG_EDEVICE, "internal: K3 kept %llu of %llu hits", (unsigned long long)kept, (unsigned long long)cnt);
    if (kept > L->copied) {
      if (kept > L->h_pairs.cap) {
        if (L->h_pairs.ensure(kept)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
        L->copied = 0;
      }
      HIPCHK(hipMemcpyAsync(L->h_pairs.p + L->copied, L->d_pairs.p + L->copied, (kept - L->copied) * sizeof(kmcpg_pair), hipMemcpyDeviceToHost, A->copy_stream));
      HIPCHK(hipStreamSynchronize(A->copy_stream));
      L->copied = kept;
    }
  } else if (fetch && cnt > L->copied) {
    if (cnt > L->h_hits.cap) {
      if (L->h_hits.ensure(cnt)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
      L->copied = 0;
    }
    // the batch's kernels are done (event above): the rest of its hits comes over the copy stream, beside later batches' kernels
    HIPCHK(hipMemcpyAsync(L->h_hits.p + L->copied, L->d_hits.p + L->copied, (cnt - L->copied) * sizeof(kmcpg_hit), hipMemcpyDeviceToHost, A->copy_stream));
    HIPCHK(hipStreamSynchronize(A->copy_stream));
    L->copied = cnt;
  }
  if (L->n >= 1024) {  // a batch large enough to say something about the data (one genome with 5 000 matches does not)
    const uint64_t per_k = cnt * 1024 / L->n + 1, old = A->hits_hint.load();
    A->hits_hint.store(per_k > old ? per_k : old - old / 8 + per_k / 8);  // rises at once, decays slowly
  }
  *n_hits = cnt;
  return 0;
}

void drop_ticket(kmcpg_ticket* t, bool failed = false) {
  for (auto& pt : t->parts) {
    if (pt.shard->opts.device >= 0) (void)hipSetDevice(pt.shard->opts.device);
    // the lane must be idle before someone else stages into it
    if (failed && pt.shard->async && pt.shard->async->stream) {
      if (pt.shard->async->up_stream) (void)hipStreamSynchronize(pt.shard->async->up_stream);
      (void)hipStreamSynchronize(pt.shard->async->stream);
      if (pt.shard->async->stream2) (void)hipStreamSynchronize(pt.shard->async->stream2);
      if (pt.shard->async->copy_stream) (void)hipStreamSynchronize(pt.shard->async->copy_stream);
    }
    else if (pt.lane->done && pt.lane->n) {
      (void)hipEventSynchronize(pt.lane->done);
      if (pt.lane->grouped && pt.lane->copied && pt.lane->eager_aside && pt.lane->eager_ev) (void)hipEventSynchronize(pt.lane->eager_ev);
    }
    release_lane(pt.shard->async, pt.lane, pt.retry);
  }
  delete t;
}

// A database larger than the HBM at hand (the reference reads any size through mmap / --low-mem, util-db-search.go:1238-1280,
// :6975-7335): the batch is searched against one shard of the index after the other on the same GPU — make shard r resident
// (kmcpg_open with shard r of S), K1 + K2, keep the hit tuples, drop it, next shard — and the concatenated hit lists are
// finalized once, exactly as the lists of S GPUs would be.  The shard searched last stays resident and is the first one of
// the next batch, so a batch costs S - 1 uploads: the larger the batch, the smaller their share.
int search_paged(kmcpg_db* front, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n, const kmcpg_params& p,
                 kmcpg_ticket* t) {
  std::lock_guard<std::mutex> g(front->paged_mu);
  t->paged = true;
  t->hits.clear();
  t->qk.assign(n, 0);
  t->ql.assign(n, 0);
  if (n == 0) return 0;
  if (offs[0] != 0 || (seqs2 && offs2[0] != 0)) return kmcpg_fail(KMCPG_EINVAL, "offs[0] must be 0");
  for (uint32_t i = 0; i < n; i++)
    if (offs[i + 1] < offs[i] || (seqs2 && offs2[i + 1] < offs2[i])) return kmcpg_fail(KMCPG_EINVAL, "offsets must not decrease (read %u)", i);
  // retries (--try-se, smaller k) read the batch again after kmcpg_submit has returned the caller's buffers
  t->seqs.assign(seqs, seqs + offs[n]);
  t->offs.assign(offs, offs + n + 1);
  if (seqs2) {
    t->seqs2.assign(seqs2, seqs2 + offs2[n]);
    t->offs2.assign(offs2, offs2 + n + 1);
  }
  t->S[0] = t->seqs.data();
  t->O[0] = t->offs.data();
  t->S[1] = seqs2 ? t->seqs2.data() : nullptr;
  t->O[1] = seqs2 ? t->offs2.data() : nullptr;
  const int S = front->paged_passes;
  const int first = front->paged_rank >= 0 ? front->paged_rank : 0;
  for (int i = 0; i < S; i++) {
    const int r = (first + i) % S;
    if (!front->paged_resident || front->paged_rank != r) {
      if (front->paged_resident) {
        int rc = kmcpg_close(front->paged_resident);
        front->paged_resident = nullptr;
        front->paged_rank = -1;
        if (rc) return rc;
      }
      kmcpg_opts so{front->paged_device, r, S, 0};
      kmcpg_db* sh = nullptr;
      int rc = open_like(front, &so, &sh);  // (the front has parsed every header already)
      if (rc) return kmcpg_fail(rc, "pass %d of %d: %s", r + 1, S, std::string(kmcpg_err_ref()).c_str());
      front->paged_resident = sh;
      front->paged_rank = r;
      front->paged_uploads++;
    }
    kmcpg_db* sh = front->paged_resident;
    if (sh->info.n_blocks_local == 0) continue;
    AsyncState* A = nullptr;
    int rc = async_state(sh, &A);
    if (rc) return rc;
    Lane* L = acquire_lane(A, false, true);
    uint64_t cnt = 0;
    rc = stage(L, seqs, offs, seqs2, offs2, n);
    if (rc == 0) rc = enqueue(sh, A, L, p);
    if (rc == 0) rc = collect(sh, A, L, p, &cnt);
    if (rc == 0) {
      t->hits.insert(t->hits.end(), L->h_hits.p, L->h_hits.p + cnt);
      if (i == 0) {  // every shard generates the same k-mers
        memcpy(t->qk.data(), L->h_qk.p, (size_t)n * sizeof(int32_t));
        memcpy(t->ql.data(), L->h_ql.p, (size_t)n * sizeof(int32_t));
      }
    } else {
      (void)hipStreamSynchronize(A->stream);
      if (A->stream2) (void)hipStreamSynchronize(A->stream2);
    }
    release_lane(A, L, false);
    if (rc) return kmcpg_fail(rc, "pass %d of %d: %s", r + 1, S, std::string(kmcpg_err_ref()).c_str());
  }
  return 0;
}

int submit_impl(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n, const kmcpg_params& p, bool retry,
                bool block, kmcpg_ticket** out, const PackedIn* packed = nullptr) {
  std::unique_ptr<kmcpg_ticket> t(new kmcpg_ticket());
  t->db = db;
  t->n = n;
  t->paired = seqs2 != nullptr;
  t->p = p;
  if (db->paged_passes > 0) {
    int rc = search_paged(db, seqs, offs, seqs2, offs2, n, p, t.get());
    if (rc) return rc;
    *out = t.release();
    return 0;
  }
  std::vector<kmcpg_db*> targets = db->shards.empty() ? std::vector<kmcpg_db*>{db} : db->shards;
  for (kmcpg_db* sh : targets) {
    AsyncState* A = nullptr;
    int rc = async_state(sh, &A);
    if (rc) {
      drop_ticket(t.release(), true);
      return rc;
    }
    Lane* lane = acquire_lane(A, retry, block);
    if (!lane) {
      drop_ticket(t.release());
      return kmcpg_fail(KMCPG_EBUSY, "all %zu lanes of this handle are in flight: kmcpg_wait for a ticket first (KMCPG_INFLIGHT)", A->max_lanes);
    }
    t->parts.push_back({sh, lane, retry});
  }
  // every GPU gets the whole batch (150 B per read; cheaper than moving k-mer hashes between GPUs)
  std::vector<int> rcs(t->parts.size(), 0);
  std::vector<std::string> errs(t->parts.size());
  // the staged ASCII is read again only by the retries of retry_unmatched (--try-se on pairs, the smaller k of a multi-k database)
  const bool can_pack = !seqs2 && !retry && (p.k > 0 || db->ks_desc.size() < 2);
  auto one = [&](size_t i) {
    auto& pt = t->parts[i];
    rcs[i] = packed ? stage_packed(pt.lane, *packed, offs, n) : stage(pt.lane, seqs, offs, seqs2, offs2, n, can_pack);
    if (rcs[i] == 0) rcs[i] = enqueue(pt.shard, pt.shard->async, pt.lane, p);
    if (rcs[i]) errs[i] = kmcpg_err_ref();
  };
  if (t->parts.size() == 1) one(0);
  else {
    std::vector<std::thread> th;
    for (size_t i = 0; i < t->parts.size(); i++) th.emplace_back(one, i);
    for (auto& x : th) x.join();
  }
  for (size_t i = 0; i < rcs.size(); i++)
    if (rcs[i]) {
      const int rc = rcs[i];
      const std::string msg = t->parts.size() > 1 ? "device " + std::to_string(t->parts[i].shard->opts.device) + ": " + errs[i] : errs[i];
      drop_ticket(t.release(), true);
      return kmcpg_fail(rc, "%s", msg.c_str());
    }
  Lane* L0 = t->parts[0].lane;
  t->S[0] = L0->h_seqs.p;
  t->O[0] = L0->h_offs.p;
  t->S[1] = t->paired ? L0->h_seqs2.p : nullptr;
  t->O[1] = t->paired ? L0->h_offs2.p : nullptr;
  *out = t.release();
  return 0;
}

// raw results of a ticket -> finalized matches (host half).  The hit lists of the parts are concatenated exactly as the
// reference concatenates the replies of its per-block workers (:946-964).
int finish_raw(kmcpg_ticket* t, kmcpg_result* out) {
  const kmcpg_hit* hits = nullptr;
  uint64_t n_hits = 0;
  static thread_local std::vector<kmcpg_hit, NoInitAlloc<kmcpg_hit>> merged;
  if (t->paged)
    return kmcpg_finalize(t->db, t->hits.data(), t->hits.size(), t->n ? t->qk.data() : nullptr, t->n ? t->ql.data() : nullptr, t->n, &t->p, out);
  if (t->parts.size() == 1 && !t->db->exchange) {
    auto& pt = t->parts[0];
    int rc = collect(pt.shard, pt.shard->async, pt.lane, t->p, &n_hits);
    if (rc) return rc;
    if (pt.lane->grouped && t->n)
      return finalize_grouped_trusted(t->db, pt.lane->h_pairs.p, pt.lane->h_roffs.p, pt.lane->h_qk.p, pt.lane->h_ql.p, t->n, t->p, out, pt.lane->bound_n);
    hits = pt.lane->h_hits.p;
  } else if (Exchange* x = t->db->exchange) {
    // the shards' lists meet on the first GPU (RCCL send/recv over xGMI, exactly the bytes each shard produced) and come to
    // the host in one copy
    std::vector<const void*> src;
    std::vector<uint64_t> bytes;
    for (auto& pt : t->parts) {
      uint64_t c = 0;
      int rc = collect(pt.shard, pt.shard->async, pt.lane, t->p, &c, false);
      if (rc) return kmcpg_fail(rc, "device %d: %s", pt.shard->opts.device, std::string(kmcpg_err_ref()).c_str());
      src.push_back(pt.lane->d_hits.p);
      bytes.push_back(c * sizeof(kmcpg_hit));
      n_hits += c;
    }
    Lane* L0 = t->parts[0].lane;
    kmcpg_db* sh0 = t->parts[0].shard;
    if (hipSetDevice(sh0->opts.device) != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "hipSetDevice failed");
    if (sh0->async->device_finalize && t->n && n_hits) {
      // K3 on the gathering GPU, over the concatenation of all shards' lists (every shard knows every column's k-mer count):
      // grouped by read, filtered by -T, in final order; 8 bytes per match and the reads' offsets come to the host
      if (L0->d_pairs.ensure(n_hits) || L0->d_roffs.ensure((size_t)t->n + 2) || L0->h_roffs.ensure((size_t)t->n + 2))
        return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
      uint64_t* h_word = L0->h_cnt.p;  // pinned; the lane's counters have been read by collect()
      h_word[0] = n_hits;
      const kmcpg_params p = t->p;
      const uint32_t n = t->n;
      int rc = exchange_gather(x, src, bytes, nullptr, [&](uint8_t* d_cat, uint64_t, void* st) -> int {
        uint64_t* d_word = (uint64_t*)(d_cat - 16);
        HIPCHK(hipMemcpyAsync(d_word, h_word, sizeof(uint64_t), hipMemcpyHostToDevice, (hipStream_t)st));
        int rc2 = kmcpg_group_device(sh0, (const kmcpg_hit*)d_cat, d_word, n_hits, L0->d_qk.p, n, &p, L0->d_pairs.p, L0->d_roffs.p, st);
        if (rc2) return rc2;
        HIPCHK(hipMemcpyAsync(L0->h_roffs.p, L0->d_roffs.p, ((size_t)n + 2) * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)st));
        // only what survived -T comes down: the offsets first (nobody else's gather is held up by this wait: the exchange mutex is
        // not held here), then h_roffs[n] pairs, as collect() does on the single-GPU path
        HIPCHK(hipStreamSynchronize((hipStream_t)st));
        const uint64_t kept = std::min<uint64_t>(L0->h_roffs.p[n], n_hits);
        if (L0->h_pairs.ensure(kept + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
        if (kept) HIPCHK(hipMemcpyAsync(L0->h_pairs.p, L0->d_pairs.p, kept * sizeof(kmcpg_pair), hipMemcpyDeviceToHost, (hipStream_t)st));
        return 0;
      });
      if (rc) return rc;
      int32_t bound_n = INT32_MAX;  // every shard ran the same params on the same batch; the smallest cover is what holds for the merged list
      for (auto& pt : t->parts) bound_n = std::min(bound_n, pt.lane->bound_n);
      return finalize_grouped_trusted(t->db, L0->h_pairs.p, L0->h_roffs.p, L0->h_qk.p, L0->h_ql.p, t->n, t->p, out, bound_n);
    }
    if (L0->h_hits.ensure(n_hits + 1)) return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc failed");
    int rc = exchange_gather(x, src, bytes, (uint8_t*)L0->h_hits.p);
    if (rc) return rc;
    hits = L0->h_hits.p;
  } else {
    merged.clear();
    for (auto& pt : t->parts) {
      uint64_t c = 0;
      int rc = collect(pt.shard, pt.shard->async, pt.lane, t->p, &c);
      if (rc) return kmcpg_fail(rc, "device %d: %s", pt.shard->opts.device, std::string(kmcpg_err_ref()).c_str());
      merged.insert(merged.end(), pt.lane->h_hits.p, pt.lane->h_hits.p + c);
    }
    hits = merged.data();
    n_hits = merged.size();
  }
  Lane* L0 = t->parts[0].lane;  // every shard generates the same k-mers
  return kmcpg_finalize(t->db, hits, n_hits, t->n ? L0->h_qk.p : nullptr, t->n ? L0->h_ql.p : nullptr, t->n, &t->p, out);
}

// a small synchronous search on the retry lane(s): the sub-batches of --try-se and of the smaller k of multi-k databases
int search_sync(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n, const kmcpg_params& p,
                kmcpg_result* out) {
  kmcpg_ticket* t = nullptr;
  int rc = submit_impl(db, seqs, offs, seqs2, offs2, n, p, true, true, &t);
  if (rc) return rc;
  rc = finish_raw(t, out);
  const std::string keep = rc ? kmcpg_err_ref() : std::string();
  drop_ticket(t, rc != 0);
  if (rc) kmcpg_err_ref() = keep;
  return rc;
}

// What the reference does with a query that was searched but matched nothing (handleQuery, util-db-search.go:763-1025):
// with --try-se the k-mers of read 1, then of read 2 are searched on their own (:831-850, :1001-1014; the length gate is not
// applied again); if the database holds several k-mer sizes the whole procedure is repeated with the next smaller k
// (:764, :1016-1022).  A query with fewer than MinMatched k-mers at any step is final (:854-869).
int retry_unmatched(kmcpg_ticket* t, kmcpg_result* out) {
  kmcpg_db* db = t->db;
  const kmcpg_params& p = t->p;
  const uint32_t n = t->n;
  std::vector<int> ks = db->ks_desc;  // descending
  if (p.k > 0) ks.assign(1, p.k);     // an explicit k: no walk over the database's sizes
  const bool try_se = p.try_se && t->paired;
  if (n == 0 || (!try_se && ks.size() < 2)) return 0;
  ResultOwner* o = (ResultOwner*)out->owner;
  const uint8_t* S[2] = {t->S[0], t->S[1]};  // the batch as staged
  const uint64_t* O[2] = {t->O[0], t->O[1]};
  std::vector<char> final_(n, 0);
  std::vector<uint32_t> todo;
  std::vector<uint8_t> sub[2];
  std::vector<uint64_t> so[2];
  // splices the result of a sub-batch (queries `todo`) into the batch result
  auto splice = [&](const kmcpg_result& r2, bool whole_query, int k) {
    std::vector<uint64_t> noffs((size_t)n + 1, 0);
    MatchVec nm;
    size_t ti = 0;
    for (uint32_t r = 0; r < n; r++) {
      if (ti < todo.size() && todo[ti] == r) {
        o->qlen[r] = r2.qlen[ti];
        o->ksize[r] = k;
        if (r2.qkmers[ti] > 0) o->qkmers[r] = r2.qkmers[ti];
        else {
          final_[r] = 1;  // fewer than MinMatched k-mers: the reference returns here (:854-869)
          if (whole_query) o->qkmers[r] = 0;  // a fresh QueryResult: NumKmers never set (this build reports 0)
        }
        if (r2.match_offs[ti + 1] > r2.match_offs[ti]) final_[r] = 1;
        nm.insert(nm.end(), r2.matches + r2.match_offs[ti], r2.matches + r2.match_offs[ti + 1]);
        ti++;
      } else {
        nm.insert(nm.end(), o->matches.begin() + (ptrdiff_t)o->offs[r], o->matches.begin() + (ptrdiff_t)o->offs[r + 1]);
      }
      noffs[r + 1] = nm.size();
    }
    o->matches.swap(nm);
    o->offs.swap(noffs);
  };
  // after the first pass: matched, or never searched (too short / fewer than MinMatched k-mers) => final
  for (uint32_t r = 0; r < n; r++) final_[r] = (o->offs[r + 1] > o->offs[r]) || o->qkmers[r] <= 0;
  int rc = 0;
  for (size_t ik = 0; ik < ks.size() && rc == 0; ik++) {
    if (ik > 0) {  // the whole query again with the next smaller k
      todo.clear();
      for (uint32_t r = 0; r < n; r++)
        if (!final_[r]) todo.push_back(r);
      if (todo.empty()) break;
      for (int m = 0; m < 2; m++) {
        sub[m].clear();
        so[m].assign(1, 0);
        if (!S[m]) continue;
        for (uint32_t r : todo) {
          sub[m].insert(sub[m].end(), S[m] + O[m][r], S[m] + O[m][r + 1]);
          so[m].push_back(sub[m].size());
        }
      }
      kmcpg_params q = p;
      q.k = ks[ik];
      q.try_se = 0;
      kmcpg_result r2;
      rc = search_sync(db, sub[0].data(), so[0].data(), S[1] ? sub[1].data() : nullptr, S[1] ? so[1].data() : nullptr, (uint32_t)todo.size(), q, &r2);
      if (rc) break;
      splice(r2, true, ks[ik]);
      kmcpg_result_free(&r2);
    }
    if (!try_se) continue;
    for (int mate = 0; mate < 2 && rc == 0; mate++) {
      todo.clear();
      for (uint32_t r = 0; r < n; r++)
        if (!final_[r]) todo.push_back(r);
      if (todo.empty()) break;
      sub[0].clear();
      so[0].assign(1, 0);
      for (uint32_t r : todo) {
        sub[0].insert(sub[0].end(), S[mate] + O[mate][r], S[mate] + O[mate][r + 1]);
        so[0].push_back(sub[0].size());
      }
      kmcpg_params q = p;
      q.k = ks[ik];
      q.min_qlen = 0;  // the length gate was applied once, before k-mer generation
      q.try_se = 0;
      kmcpg_result r2;
      rc = search_sync(db, sub[0].data(), so[0].data(), nullptr, nullptr, (uint32_t)todo.size(), q, &r2);
      if (rc) break;
      splice(r2, false, ks[ik]);
      kmcpg_result_free(&r2);
    }
  }
  out->qlen = o->qlen.data();
  out->qkmers = o->qkmers.data();
  out->ksize = o->ksize.data();
  out->matches = o->matches.data();
  out->match_offs = o->offs.data();
  return rc;
}

}  // namespace

extern "C" int kmcpg_open_devices(const char* db_dir, const int32_t* devices, int32_t n_devices, kmcpg_db** out) {
  if (!db_dir || !devices || !out || n_devices < 1) return kmcpg_fail(KMCPG_EINVAL, "bad argument");
  *out = nullptr;
  kmcpg_opts mo{-1, 0, 1, 0};
  kmcpg_db* front = nullptr;
  int rc = kmcpg_open(db_dir, &mo, &front);  // metadata of every block: names, sizes, FPR table
  if (rc) return rc;
  front->info.n_blocks_local = 0;
  front->info.matrix_bytes_local = 0;
  front->info.row_bytes_sum_local = 0;
  for (int32_t i = 0; i < n_devices; i++) {
    kmcpg_opts so{devices[i], i, n_devices, 0};
    kmcpg_db* sh = nullptr;
    rc = open_like(front, &so, &sh);  // (the headers were parsed once, by the front)
    if (rc) {
      std::string keep = kmcpg_err_ref();
      kmcpg_close(front);
      kmcpg_err_ref() = keep;
      return rc;
    }
    front->shards.push_back(sh);
    front->info.n_blocks_local += sh->info.n_blocks_local;
    front->info.matrix_bytes_local += sh->info.matrix_bytes_local;
    front->info.row_bytes_sum_local += sh->info.row_bytes_sum_local;
  }
  // the exchange step: RCCL gather of the hit lists onto the first GPU when the devices allow it, host merge otherwise
  front->exchange = exchange_create(std::vector<int>(devices, devices + n_devices), &front->exchange_why);
  if (front->exchange)
    for (kmcpg_db* sh : front->shards) {
      AsyncState* A = nullptr;
      rc = async_state(sh, &A);
      if (rc) {
        std::string keep = kmcpg_err_ref();
        kmcpg_close(front);
        kmcpg_err_ref() = keep;
        return rc;
      }
      A->hits_stay_on_device = true;
    }
  *out = front;
  return 0;
}

extern "C" const char* kmcpg_exchange_info(const kmcpg_db* db) {
  if (!db) return "";
  if (db->exchange) return exchange_note(db->exchange);
  static thread_local std::string s;
  s = db->shards.empty() ? "single device: no exchange step" : "host merge of the shards' hit lists (" + db->exchange_why + ")";
  return s.c_str();
}

extern "C" int kmcpg_open_paged(const char* db_dir, int32_t device, int32_t passes, kmcpg_db** out) {
  if (!db_dir || !out || passes < 0) return kmcpg_fail(KMCPG_EINVAL, "bad argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return kmcpg_fail(KMCPG_EDEVICE, "no HIP device available: libkmcpgpu has no CPU fallback");
  if (device < 0 || device >= ndev) return kmcpg_fail(KMCPG_EINVAL, "device %d out of range (%d devices)", device, ndev);
  kmcpg_opts mo{-1, 0, 1, 0};
  kmcpg_db* front = nullptr;
  int rc = kmcpg_open(db_dir, &mo, &front);  // metadata of every block: names, sizes, FPR table
  if (rc) return rc;
  uint64_t largest = 0, free_b = 0, reserve = 0;
  const int fit = plan_passes(front, device, &largest, &free_b, &reserve);
  if (passes == 0) passes = fit;
  if (fit == 0 || passes < fit) {
    kmcpg_close(front);
    if (fit == 0)
      return kmcpg_fail(KMCPG_ENOMEM, "index does not fit in HBM even one block at a time: the largest block needs %.2f GB, %.2f GB free on device %d", largest / 1e9,
                        free_b / 1e9, device);
    return kmcpg_fail(KMCPG_ENOMEM, "%d pass(es) do not fit in HBM (%.2f GB free on device %d): at least %d are needed", passes, free_b / 1e9, device, fit);
  }
  if (passes > (int)std::max<size_t>(1, front->blocks.size())) passes = (int)std::max<size_t>(1, front->blocks.size());
  if (passes == 1) {  // it fits after all: an ordinary resident handle
    kmcpg_close(front);
    kmcpg_opts so{device, 0, 1, 0};
    return kmcpg_open(db_dir, &so, out);
  }
  front->paged_passes = passes;
  front->paged_device = device;
  // what stays free beside the largest shard of the tightest partition that fits (>= the reserve plan_passes asked for; more
  // passes than needed leave more than this: the hint stays on the safe side)
  front->paged_reserve = std::max<uint64_t>(reserve, free_b > largest ? free_b - largest : 0);
  front->info.n_blocks_local = front->info.n_blocks;
  front->info.matrix_bytes_local = front->info.matrix_bytes;
  *out = front;
  return 0;
}

extern "C" int kmcpg_paged_info(const kmcpg_db* db, int32_t* passes, uint64_t* uploads) {
  if (!db) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (passes) *passes = db->paged_passes;
  if (uploads) *uploads = db->paged_uploads;
  return 0;
}

static int submit_checked(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n_reads,
                          const kmcpg_params* params, bool block, kmcpg_ticket** out) {
  if (!db || !out || (n_reads && (!seqs || !offs))) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  *out = nullptr;
  if ((seqs2 == nullptr) != (offs2 == nullptr)) return kmcpg_fail(KMCPG_EINVAL, "seqs2 and offs2 must be given together");
  if (db->opts.shard_count != 1)
    return kmcpg_fail(KMCPG_EINVAL, "kmcpg_submit/kmcpg_search_batch need the whole database: open it on one GPU or with kmcpg_open_devices; use kmcpg_query_device + kmcpg_finalize per shard");
  if (db->shards.empty() && db->opts.device < 0 && db->paged_passes == 0) return kmcpg_fail(KMCPG_EDEVICE, "metadata-only handle (device -1): no GPU work possible");
  const kmcpg_params p = params ? *params : default_params();
  if (p.min_matched < 1) return kmcpg_fail(KMCPG_EINVAL, "min_matched must be >= 1");
  return submit_impl(db, seqs, offs, seqs2, offs2, n_reads, p, false, block, out);
}

extern "C" int kmcpg_submit(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n_reads,
                            const kmcpg_params* params, kmcpg_ticket** out) {
  return submit_checked(db, seqs, offs, seqs2, offs2, n_reads, params, false, out);
}

// ---- packed queries (kmcp_gpu.h): the packer and its inverse for hosts, and the entry that takes codes
extern "C" int kmcpg_pack2(const uint8_t* seq, uint64_t n, uint64_t pos, uint8_t* codes, kmcpg_exc_run* exc, uint64_t exc_cap, uint64_t* n_exc) {
  if ((n && (!seq || !codes)) || !n_exc || (exc_cap && !exc)) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  static_assert(sizeof(kmcpg_exc_run) == sizeof(PackRun), "one layout");
  static thread_local std::vector<PackRun> runs;
  runs.clear();
  pack2_append(seq, (size_t)n, codes, pos, runs);
  const uint64_t have = *n_exc, total = have + runs.size();
  // a run that continues the caller's last one (a gap of N's across two calls) is joined with it
  size_t first = 0;
  bool join = false;
  if (have && have <= exc_cap && !runs.empty()) {
    const kmcpg_exc_run& l = exc[have - 1];
    join = l.byte == runs[0].byte && l.pos + l.len == runs[0].pos && (uint64_t)l.len + runs[0].len <= 0xffffffffu;
  }
  if (join) first = 1;
  const uint64_t need = have + (runs.size() - first);
  if (need > exc_cap) {  // nothing of the caller's list has been touched: the call can be repeated with more room
    *n_exc = total;
    return kmcpg_fail(KMCPG_ENOMEM, "kmcpg_pack2: room for %llu exception runs, %llu needed", (unsigned long long)exc_cap, (unsigned long long)total);
  }
  if (join) exc[have - 1].len += runs[0].len;
  for (size_t i = first; i < runs.size(); i++) exc[have + (i - first)] = kmcpg_exc_run{runs[i].pos, runs[i].len, runs[i].byte};
  *n_exc = need;
  return 0;
}

extern "C" int kmcpg_host_alloc(uint64_t bytes, void** out) {
  if (!out) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return kmcpg_fail(KMCPG_EDEVICE, "no HIP device available: libkmcpgpu has no CPU fallback");
  if (hipHostMalloc(out, (size_t)std::max<uint64_t>(bytes, 64), hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return kmcpg_fail(KMCPG_ENOMEM, "hipHostMalloc of %llu bytes failed", (unsigned long long)bytes);
  }
  return 0;
}

extern "C" int kmcpg_host_free(void* p) {
  if (p && hipHostFree(p) != hipSuccess) {
    (void)hipGetLastError();
    return kmcpg_fail(KMCPG_EINVAL, "not a pointer from kmcpg_host_alloc");
  }
  return 0;
}

extern "C" int kmcpg_unpack2(const uint8_t* codes, uint64_t n_bases, const kmcpg_exc_run* exc, uint64_t n_exc, uint8_t* out) {
  if ((n_bases && (!codes || !out)) || (n_exc && !exc)) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  static const char lut[4] = {'A', 'C', 'T', 'G'};
  for (uint64_t j = 0; j < n_bases; j++) out[j] = (uint8_t)lut[(codes[j >> 2] >> (2 * (j & 3))) & 3];
  for (uint64_t i = 0; i < n_exc; i++) {
    const kmcpg_exc_run& e = exc[i];
    if (e.pos > n_bases || e.len > n_bases - e.pos || e.byte > 255) return kmcpg_fail(KMCPG_EINVAL, "exception run %llu lies outside the batch", (unsigned long long)i);
    memset(out + e.pos, (int)e.byte, e.len);
  }
  return 0;
}

extern "C" int kmcpg_submit_packed(kmcpg_db* db, const uint8_t* codes, const uint64_t* offs, const kmcpg_exc_run* exc, uint64_t n_exc, uint32_t n_reads,
                                   const kmcpg_params* params, kmcpg_ticket** out) {
  if (!db || !out || (n_reads && (!codes || !offs)) || (n_exc && !exc)) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  *out = nullptr;
  if (db->opts.shard_count != 1)
    return kmcpg_fail(KMCPG_EINVAL, "kmcpg_submit/kmcpg_search_batch need the whole database: open it on one GPU or with kmcpg_open_devices; use kmcpg_query_device + kmcpg_finalize per shard");
  if (db->shards.empty() && db->opts.device < 0 && db->paged_passes == 0) return kmcpg_fail(KMCPG_EDEVICE, "metadata-only handle (device -1): no GPU work possible");
  const kmcpg_params p = params ? *params : default_params();
  if (p.min_matched < 1) return kmcpg_fail(KMCPG_EINVAL, "min_matched must be >= 1");
  // handles that read the batch's text again — the smaller k of a multi-k database (retry_unmatched), the passes of a paged index — get text
  if (db->paged_passes > 0 || (p.k <= 0 && db->ks_desc.size() > 1)) {
    const uint64_t tb = n_reads ? offs[n_reads] : 0;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> text((size_t)tb + 16);
    if (int rc = kmcpg_unpack2(codes, tb, exc, n_exc, text.data())) return rc;
    return submit_impl(db, text.data(), offs, nullptr, nullptr, n_reads, p, false, false, out);
  }
  const PackedIn in{codes, exc, n_exc};
  return submit_impl(db, nullptr, offs, nullptr, nullptr, n_reads, p, false, false, out, &in);
}

extern "C" int kmcpg_wait(kmcpg_ticket* t, kmcpg_result* out) {
  if (!t || !out) {
    if (t) drop_ticket(t);
    return kmcpg_fail(KMCPG_EINVAL, "null argument");
  }
  memset(out, 0, sizeof *out);
  int rc = finish_raw(t, out);
  if (rc == 0) rc = retry_unmatched(t, out);
  const std::string keep = rc ? kmcpg_err_ref() : std::string();
  if (rc) kmcpg_result_free(out);
  drop_ticket(t, rc != 0);
  if (rc) kmcpg_err_ref() = keep;
  return rc;
}

// One large batch, several pieces in flight.  kmcpg_search_batch is a synchronous call: upload, K1 + K2 + K3, the copy of the ordered
// pairs and their expansion to Match records would follow each other with nothing overlapped (on a database of close relatives —
// 200 matches per read — the copy and the expansion take as long as the kernels).  Here the batch is cut into up to 4 pieces of at
// least 16 384 queries that go through the handle's lanes one behind the other: while the GPU works on piece j + 1, piece j comes
// down and is written straight into its place in the ONE result of the call (finalize_grouped_into), so the result is exactly that of
// the unsplit batch and nothing is copied twice.  Taken when the handle holds the whole database on one GPU with K3 on and no query can
// need a second search (--try-se on pairs, several k-mer sizes: those splice sub-results and keep the plain path).
// A thread only ever BLOCKS for a lane while it holds none (two threads in here at once share the lanes instead of waiting for each
// other's).  *took = false: not applicable, nothing done.
static int search_batch_pieces(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n,
                               const kmcpg_params& p, kmcpg_result* out, bool* took) {
  *took = false;
  uint32_t kMinPiece = 16384;
  if (const char* e = getenv("KMCPG_PIECE_MIN")) kMinPiece = (uint32_t)std::max(1, atoi(e));  // tests and tools/stress_async.py: pieces of small batches
  int want = 4;
  if (const char* e = getenv("KMCPG_PIECES")) want = atoi(e);
  if (want < 2 || (uint64_t)n < 2 * (uint64_t)kMinPiece) return 0;
  if (!db->shards.empty() || db->paged_passes > 0 || db->opts.device < 0 || db->opts.shard_count != 1) return 0;
  if ((p.try_se && seqs2) || (p.k <= 0 && db->ks_desc.size() > 1)) return 0;
  AsyncState* A = nullptr;
  if (int rc = async_state(db, &A)) return rc;
  if (!A->device_finalize || A->hits_stay_on_device) return 0;
  // The first large batch of a handle goes through whole: it teaches the handle how many hits to expect per read (hits_hint), so that
  // the pieces of later batches get hit buffers and eager copies of the right size at once.  (Cold, every piece would overflow its
  // buffer and run twice, and four lanes would allocate their pinned buffers inside the call: kmcp-search on a 100 000-read file
  // spent 0.34 s instead of 0.14 s in here.)
  if (A->hits_hint.load() == 0) return 0;
  const uint32_t S = (uint32_t)std::min<uint64_t>((uint64_t)want, std::min<uint64_t>(A->max_lanes, n / kMinPiece));
  if (S < 2) return 0;
  *took = true;
  struct Piece {
    Lane* lane = nullptr;
    uint32_t lo = 0, cnt = 0;
  };
  std::vector<Piece> pc(S);
  for (uint32_t j = 0; j < S; j++) {
    pc[j].lo = (uint32_t)((uint64_t)n * j / S);
    pc[j].cnt = (uint32_t)((uint64_t)n * (j + 1) / S) - pc[j].lo;
  }
  ResultOwner* o = result_owner_take();
  result_owner_shape(o, n);
  uint64_t match_base = 0;
  uint32_t next_finish = 0, next_submit = 0;
  int rc = 0;
  std::vector<uint64_t> o1, o2;
  const bool timing = getenv("KMCPG_FIN_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  auto finish_one = [&]() -> int {  // the oldest piece in flight: wait, bring its pairs down, expand them into their place
    Piece& q = pc[next_finish];
    uint64_t n_hits = 0;
    const double ta = now();
    int r = collect(db, A, q.lane, p, &n_hits);
    const double tb = now();
    uint64_t kept = 0;
    if (r == 0)
      r = finalize_grouped_into(db, q.lane->h_pairs.p, q.lane->h_roffs.p, q.lane->h_qk.p, q.lane->h_ql.p, q.cnt, p, o, q.lo, match_base, &kept, true, q.lane->bound_n);
    if (timing)
      fprintf(stderr, "piece %u: collect from %.2f to %.2f ms, expanded by %.2f ms (%llu hits, %llu kept)\n", next_finish, ta - t_begin, tb - t_begin, now() - t_begin,
              (unsigned long long)n_hits, (unsigned long long)kept);
    if (r) {
      (void)hipStreamSynchronize(A->up_stream);
      (void)hipStreamSynchronize(A->stream);
      if (A->stream2) (void)hipStreamSynchronize(A->stream2);
      (void)hipStreamSynchronize(A->copy_stream);
    }
    release_lane(A, q.lane, false);
    q.lane = nullptr;
    next_finish++;
    match_base += kept;
    return r;
  };
  while (rc == 0 && next_submit < S) {
    Lane* L = acquire_lane(A, false, next_finish == next_submit);  // block only while holding no lane
    if (!L) {
      rc = finish_one();
      continue;
    }
    Piece& q = pc[next_submit];
    q.lane = L;
    o1.resize((size_t)q.cnt + 1);
    for (uint32_t r = 0; r <= q.cnt; r++) o1[r] = offs[q.lo + r] - offs[q.lo];
    if (offs2) {
      o2.resize((size_t)q.cnt + 1);
      for (uint32_t r = 0; r <= q.cnt; r++) o2[r] = offs2[q.lo + r] - offs2[q.lo];
    }
    rc = stage(L, seqs + offs[q.lo], o1.data(), seqs2 ? seqs2 + offs2[q.lo] : nullptr, offs2 ? o2.data() : nullptr, q.cnt, !seqs2);
    if (rc == 0) rc = enqueue(db, A, L, p, true);
    if (timing) fprintf(stderr, "piece %u: enqueued at %.2f ms\n", next_submit, now() - t_begin);
    next_submit++;
    if (rc) {  // this piece never got as far as a completion event: drain and give its lane back, the older ones below
      const std::string keep = kmcpg_err_ref();
      (void)hipStreamSynchronize(A->up_stream);
      (void)hipStreamSynchronize(A->stream);
      if (A->stream2) (void)hipStreamSynchronize(A->stream2);
      (void)hipStreamSynchronize(A->copy_stream);
      release_lane(A, L, false);
      q.lane = nullptr;
      next_submit--;
      kmcpg_err_ref() = keep;
    }
  }
  while (rc == 0 && next_finish < next_submit) rc = finish_one();
  if (rc) {
    const std::string keep = kmcpg_err_ref();
    (void)hipStreamSynchronize(A->up_stream);
    (void)hipStreamSynchronize(A->stream);
    if (A->stream2) (void)hipStreamSynchronize(A->stream2);
    (void)hipStreamSynchronize(A->copy_stream);
    for (uint32_t j = next_finish; j < next_submit; j++)
      if (pc[j].lane) release_lane(A, pc[j].lane, false);
    result_owner_give(o);
    kmcpg_err_ref() = keep;
    return rc;
  }
  result_publish(o, n, p.k > 0 ? p.k : db->info.k, out);
  return 0;
}

extern "C" int kmcpg_search_batch(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n_reads,
                                  const kmcpg_params* params, kmcpg_result* out) {
  if (!out) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  memset(out, 0, sizeof *out);
  kmcpg_ticket* t = nullptr;
  int rc = 0;
  bool took = false;
  if (db && n_reads && seqs && offs && (seqs2 == nullptr) == (offs2 == nullptr)) {
    const kmcpg_params pp = params ? *params : default_params();
    if (pp.min_matched >= 1 && offs[0] == 0 && (!offs2 || offs2[0] == 0)) rc = search_batch_pieces(db, seqs, offs, seqs2, offs2, n_reads, pp, out, &took);
  }
  if (took && rc == 0) return 0;
  if (took) memset(out, 0, sizeof *out);
  if (!took || rc == KMCPG_ENOMEM) {
    rc = submit_checked(db, seqs, offs, seqs2, offs2, n_reads, params, true, &t);
    if (rc == 0) rc = kmcpg_wait(t, out);
  }
  if (rc != KMCPG_ENOMEM || n_reads < 2) return rc;
  // The batch's workspace (8-24 B per base next to the resident index) or its hit buffers did not fit: the two halves of the
  // batch one after the other, their results joined — a caller that sized its batches for an emptier GPU gets its answer
  // instead of an error (the reference has no such limit: it searches query by query).  The halves split again if they must.
  const std::string why = kmcpg_err_ref();
  const uint32_t h = n_reads / 2;
  kmcpg_result part[2];
  for (int i = 0; i < 2; i++) {
    const uint32_t lo = i ? h : 0, cnt = i ? n_reads - h : h;
    // offsets of a half start at its first read: a rebased copy (the sequences themselves are addressed in place)
    std::vector<uint64_t> o1((size_t)cnt + 1), o2;
    for (uint32_t r = 0; r <= cnt; r++) o1[r] = offs[lo + r] - offs[lo];
    if (offs2) {
      o2.resize((size_t)cnt + 1);
      for (uint32_t r = 0; r <= cnt; r++) o2[r] = offs2[lo + r] - offs2[lo];
    }
    rc = kmcpg_search_batch(db, seqs + offs[lo], o1.data(), seqs2 ? seqs2 + offs2[lo] : nullptr, offs2 ? o2.data() : nullptr, cnt, params, &part[i]);
    if (rc) {
      if (i) kmcpg_result_free(&part[0]);
      return rc;
    }
  }
  ResultOwner* a = (ResultOwner*)part[0].owner;
  const ResultOwner* b = (const ResultOwner*)part[1].owner;
  const uint64_t base = a->offs.back();
  a->qlen.insert(a->qlen.end(), b->qlen.begin(), b->qlen.end());
  a->qkmers.insert(a->qkmers.end(), b->qkmers.begin(), b->qkmers.end());
  a->ksize.insert(a->ksize.end(), b->ksize.begin(), b->ksize.end());
  for (size_t r = 1; r < b->offs.size(); r++) a->offs.push_back(base + b->offs[r]);
  if (a->pairs_mode != b->pairs_mode) {  // (one half took a path that collects records: retries, a paged pass)
    result_records_to_pairs(a);
    result_records_to_pairs(const_cast<ResultOwner*>(b));
  }
  if (a->pairs_mode) a->pairs.insert(a->pairs.end(), b->pairs.begin(), b->pairs.end());
  else a->matches.insert(a->matches.end(), b->matches.begin(), b->matches.end());
  *out = part[0];
  out->n_reads = n_reads;
  out->qlen = a->qlen.data();
  out->qkmers = a->qkmers.data();
  out->ksize = a->ksize.data();
  out->match_offs = a->offs.data();
  out->matches = a->matches.data();
  kmcpg_result_free(&part[1]);
  if (getenv("KMCPG_VERBOSE")) fprintf(stderr, "kmcpg_search_batch: %u queries searched as two halves (%s)\n", n_reads, why.c_str());
  return 0;
}

// ---- compact results: the record forms under tl_pairs_mode (results shaped meanwhile collect pairs, finalize.cpp), then whatever a
//      path without native pairs produced (retries, host-merged lists, paged passes) is converted
namespace {
struct PairsScope {
  bool on;
  explicit PairsScope(bool want) : on(want) {
    if (on) tl_pairs_mode = true;
  }
  ~PairsScope() {
    if (on) tl_pairs_mode = false;
  }
};

int publish_pairs(kmcpg_result* r, kmcpg_result_pairs* out) {
  ResultOwner* o = (ResultOwner*)r->owner;
  memset(out, 0, sizeof *out);
  if (!o) return 0;  // an empty result
  result_records_to_pairs(o);
  out->n_reads = r->n_reads;
  out->k = r->k;
  out->qlen = o->qlen.data();
  out->qkmers = o->qkmers.data();
  out->ksize = o->ksize.data();
  out->match_offs = o->offs.data();
  out->pairs = o->pairs.data();
  out->owner = o;
  memset(r, 0, sizeof *r);
  return 0;
}
}  // namespace

extern "C" int kmcpg_search_batch_pairs(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n_reads,
                                        const kmcpg_params* params, kmcpg_result_pairs* out) {
  if (!out) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  memset(out, 0, sizeof *out);
  // queries that may be searched again (--try-se on pairs, the smaller k of a multi-k database) have their sub-results spliced
  // record by record (retry_unmatched): those batches collect records and are converted at the end
  const bool native = db && !((params && params->try_se && seqs2) || ((!params || params->k <= 0) && db->ks_desc.size() > 1));
  kmcpg_result r{};
  int rc;
  {
    PairsScope scope(native);
    rc = kmcpg_search_batch(db, seqs, offs, seqs2, offs2, n_reads, params, &r);
  }
  if (rc) return rc;
  return publish_pairs(&r, out);
}

extern "C" int kmcpg_wait_pairs(kmcpg_ticket* t, kmcpg_result_pairs* out) {
  if (!t || !out) {
    if (t) drop_ticket(t);
    return kmcpg_fail(KMCPG_EINVAL, "null argument");
  }
  memset(out, 0, sizeof *out);
  const bool native = !((t->p.try_se && t->paired) || (t->p.k <= 0 && t->db->ks_desc.size() > 1));
  kmcpg_result r{};
  int rc;
  {
    PairsScope scope(native);
    rc = kmcpg_wait(t, &r);
  }
  if (rc) return rc;
  return publish_pairs(&r, out);
}

// How many bases a batch may hold on this handle so that its workspace fits beside the resident index: K1/K1d take up to 24 B
// per base (8 B of hashes, 16 B of sort scratch for queries above -u and for window sketches; query.cpp), the lanes another few
// bytes per base of staging.  A paged handle answers from the reserve plan_passes() kept free beside its largest shard, every
// other handle from the HBM free right now.  *max_bases = 0: unknown (the device would not say).
extern "C" int kmcpg_batch_hint(const kmcpg_db* db, uint64_t* max_bases) {
  if (!db || !max_bases) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  uint64_t room = 0;
  if (db->paged_passes > 1) room = db->paged_reserve;
  else if (db->opts.device >= 0) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    size_t fr = 0, tot = 0;
    if (hipSetDevice(db->opts.device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) room = fr;
    else (void)hipGetLastError();
    if (cur >= 0) (void)hipSetDevice(cur);
  }
  *max_bases = room / 40;  // 24 B/base of workspace + staging of the lanes in flight + headroom for the hit buffers
  return 0;
}

// ------------------------------------------------------------------------------------------------
// bench / parity support
// ------------------------------------------------------------------------------------------------
extern "C" int kmcpg_plant(kmcpg_db* db, uint32_t col, const uint64_t* hashes, uint64_t n) {
  if (!db || (!hashes && n)) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (col >= db->col_block.size()) return kmcpg_fail(KMCPG_EINVAL, "column out of range");
  const BlockMeta& b = db->blocks[db->col_block[col]];
  if (!b.local || n == 0) return 0;
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  DevBuf<uint64_t> d;  // released on every path
  if (d.ensure(n)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  hipError_t e = hipMemcpy(d.p, hashes, n * sizeof(uint64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_plant(db->h_blockdev[(size_t)b.local_idx], col - b.col_base, db->info.num_hashes, d.p, n, nullptr);
    e = hipDeviceSynchronize();
  }
  d.release();
  if (e != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "planting: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int kmcpg_read_rows(kmcpg_db* db, uint32_t block, const uint64_t* row_idx, uint64_t n_rows, uint8_t* out) {
  if (!db || block >= db->blocks.size() || (!row_idx && n_rows) || (!out && n_rows)) return kmcpg_fail(KMCPG_EINVAL, "bad argument");
  const BlockMeta& b = db->blocks[block];
  if (!b.local) return kmcpg_fail(KMCPG_EINVAL, "block %u is not resident on this rank", block);
  for (uint64_t i = 0; i < n_rows; i++)
    if (row_idx[i] >= b.h.num_sigs) return kmcpg_fail(KMCPG_EINVAL, "row out of range");
  if (n_rows == 0) return 0;
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  DevBuf<uint64_t> d_idx;  // released on every path
  DevBuf<uint8_t> d_out;
  if (d_idx.ensure(n_rows) || d_out.ensure(n_rows * b.h.row_bytes)) {
    d_idx.release();
    d_out.release();
    return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  }
  hipError_t e = hipMemcpy(d_idx.p, row_idx, n_rows * sizeof(uint64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_gather_rows(b.d_rows, b.stride, b.h.row_bytes, d_idx.p, 0, n_rows, d_out.p, nullptr);
    e = hipMemcpy(out, d_out.p, n_rows * b.h.row_bytes, hipMemcpyDeviceToHost);
  }
  d_idx.release();
  d_out.release();
  if (e != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "reading rows back: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int kmcpg_read_row_range(kmcpg_db* db, uint32_t block, uint64_t first_row, uint64_t n_rows, uint8_t* out) {
  if (!db || block >= db->blocks.size() || (!out && n_rows)) return kmcpg_fail(KMCPG_EINVAL, "bad argument");
  const BlockMeta& b = db->blocks[block];
  if (!b.local) return kmcpg_fail(KMCPG_EINVAL, "block %u is not resident on this rank", block);
  if (first_row > b.h.num_sigs || n_rows > b.h.num_sigs - first_row) return kmcpg_fail(KMCPG_EINVAL, "row out of range");
  if (n_rows == 0) return 0;
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  if (b.h.row_bytes >= 512 && b.h.row_bytes % 16 == 0) {  // wide rows of a DMA-friendly width: one strided copy (782-byte rows took 23 s for 2.7 GB this way)
    HIPCHK(hipMemcpy2D(out, b.h.row_bytes, b.d_rows + first_row * b.stride, b.stride, b.h.row_bytes, n_rows, hipMemcpyDeviceToHost));
    return 0;
  }
  // narrow rows (a 2-D copy would move them one by one): packed on the device first
  const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / b.h.row_bytes);
  uint8_t* d_out = nullptr;
  HIPCHK(hipMalloc((void**)&d_out, std::min(chunk, n_rows) * b.h.row_bytes));
  hipError_t e = hipSuccess;
  for (uint64_t r0 = 0; r0 < n_rows && e == hipSuccess; r0 += chunk) {
    const uint64_t nr = std::min(chunk, n_rows - r0);
    launch_gather_rows(b.d_rows, b.stride, b.h.row_bytes, nullptr, first_row + r0, nr, d_out, nullptr);
    e = hipMemcpy(out + r0 * b.h.row_bytes, d_out, nr * b.h.row_bytes, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_out);
  if (e != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "reading rows back: %s", hipGetErrorString(e));
  return 0;
}
