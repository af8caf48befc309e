// engine.hpp — internals shared by the host-side translation units of libkmcpgpu.so (engine.cpp: residency;
// query.cpp: GPU half; finalize.cpp: host half; host.cpp: the whole pipeline on host buffers).  Not part of the ABI.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kmcp_gpu.h"
#include "common.hpp"
#include "dbformat.hpp"
#include "fpr.hpp"

// error sink: sets the thread-local message behind kmcpg_last_error() and returns `code`
int kmcpg_fail(int code, const char* fmt, ...);
std::string& kmcpg_err_ref();  // the calling thread's message (workers hand theirs to the caller)

#define HIPCHK(expr)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// GPU work on a handle: refused for metadata-only handles (opts.device == -1)
#define KMCPG_USE_DEVICE(db)                                                                                      \
  do {                                                                                                            \
    if ((db)->opts.device < 0) return kmcpg_fail(KMCPG_EDEVICE, "metadata-only handle (device -1): no GPU work possible"); \
    HIPCHK(hipSetDevice((db)->opts.device));                                                                      \
  } while (0)

namespace kmcpg {

struct BlockMeta {
  std::string path;
  UnikiHeader h;
  uint32_t col_base = 0;
  bool local = false;
  int local_idx = -1;
  uint32_t stride = 0;          // row pitch of the group the block lives in
  uint8_t* d_rows = nullptr;    // first byte of the block inside its group's rows (not an allocation of its own)
  int group = -1;
  uint32_t byte_off = 0;        // offset of the block's bytes inside the group's row
};

// Resident blocks with the same NumSigs share one set of rows (common.hpp BlockDev): the allocation belongs to the group.
struct Group {
  uint8_t* d_rows = nullptr;
  uint64_t num_sigs = 0;
  uint32_t stride = 0, row_bytes = 0;
  std::vector<int> members;  // global block indices, in __db.yml `files` order
};

struct SlotClass {
  int lpr = 0;
  std::vector<Slot> slots;
  Slot* d_slots = nullptr;
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    if (hipMalloc((void**)&p, want * sizeof(T)) != hipSuccess) return -1;
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace kmcpg

namespace kmcpg {
struct Exchange;
// smallest count whose FPR(n, count) passes -f, for n = 0..n (query.cpp fpr_bound); h = pinned source of the upload
struct FprBoundTable {
  uint64_t key = 0;  // bits of max_fpr
  int n = 0;
  uint16_t* d = nullptr;
  uint16_t* h = nullptr;
};
void release_fpr_bounds(kmcpg_db* db);  // query.cpp
int plan_passes(kmcpg_db* front, int device, uint64_t* largest_shard_bytes, uint64_t* free_bytes, uint64_t* reserve_bytes = nullptr);  // engine.cpp
int open_like(const kmcpg_db* src, const kmcpg_opts* opts, kmcpg_db** out);                           // engine.cpp
struct AsyncState;
void async_release(kmcpg_db* db);  // host.cpp
int async_in_flight(kmcpg_db* db);
}  // namespace kmcpg

struct kmcpg_db {
  kmcpg_opts opts{};
  kmcpg_info info{};
  std::vector<kmcpg::BlockMeta> blocks;
  std::vector<int> local;  // global indices of resident blocks, in kmcpg::BlockDev order
  std::vector<kmcpg::BlockDev> h_blockdev;  // per resident block (planting / read-back helpers)
  kmcpg::BlockDev* d_blockdev = nullptr;
  std::vector<kmcpg::Group> groups;
  std::vector<kmcpg::BlockDev> h_groupdev;  // per group: what K2 gathers from
  kmcpg::BlockDev* d_groupdev = nullptr;
  std::vector<kmcpg::Seg> h_segs;
  kmcpg::Seg* d_segs = nullptr;
  std::vector<kmcpg::SlotClass> classes;
  std::vector<uint32_t> col_block;  // global column -> block index
  struct ColMeta {   // per global column, side by side for the host half (one cache line per hit instead of three)
    uint64_t size;   // Header.Sizes: k-mers of the column
    uint64_t gsize;  // Header.GSizes
    uint32_t tidx;   // Header.Indices: chunk index | #chunks << 16
    uint32_t pad;
  };
  std::vector<ColMeta> col_meta;
  std::unique_ptr<kmcpg::QueryFpr> fpr;
  std::mutex mu;      // serialises the enqueueing of GPU-half calls (kernels of one workspace slot follow each other in stream order;
                      // calls on different streams are ordered by the slot's event)
  std::vector<int> ks_desc;    // k-mer sizes of the database, descending (`ks` of __db.yml; one entry for most databases)
  kmcpg::AsyncState* async = nullptr;  // lanes + stream of kmcpg_submit/kmcpg_wait (host.cpp), created on first use
  // The k-mer workspace of kmcpg_query_device, twice: consecutive calls take the slots in turn, so that the k-mer kernels of batch
  // i + 1 (VALU-bound) may run on another stream BESIDE the COBS kernels of batch i (memory-bound) instead of behind them.  A call
  // waits for the previous user of ITS slot only (Workspace::ev); the COBS kernels themselves stay one batch at a time (cobs_ev:
  // two of them side by side would only share the memory system, and their HIP-event durations would stop meaning anything).
  // The second slot is used only while a second workspace of the batch's size fits beside the index (query.cpp pick_slot).
  struct Workspace {
    kmcpg::DevBuf<uint64_t> w_hashes, w_scratch;
    kmcpg::DevBuf<int32_t> w_nk_raw, w_nk1, w_seg_cnt;
    kmcpg::DevBuf<uint32_t> w_long_list, w_long_meta, w_long_counts;  // long-query (split) path
    kmcpg::DevBuf<uint64_t> w_huge_info;                             // whole-genome queries: (read, n, offset)
    kmcpg::DevBuf<uint8_t> w_huge_temp;                              // histogram table of the device-wide radix sort
    kmcpg::DevBuf<uint64_t> w_gathered;                              // profiling level 2: the row loads k2_cobs issued
    hipEvent_t ev = nullptr;  // recorded at the end of every call that used the slot: the slot's next user waits for it
    bool ev_valid = false;
    hipEvent_t in_ev = nullptr, k1_ev = nullptr;  // KMCPG_K1_STREAM: inputs ready on the caller's stream / k-mers ready on k1_stream
    void release() {
      w_hashes.release(); w_scratch.release(); w_nk_raw.release(); w_nk1.release(); w_seg_cnt.release(); w_long_list.release();
      w_long_meta.release(); w_long_counts.release(); w_huge_info.release(); w_huge_temp.release(); w_gathered.release();
      if (ev) (void)hipEventDestroy(ev);
      if (in_ev) (void)hipEventDestroy(in_ev);
      if (k1_ev) (void)hipEventDestroy(k1_ev);
      ev = in_ev = k1_ev = nullptr;
      ev_valid = false;
    }
  };
  Workspace ws[2];
  uint64_t ws_calls = 0;       // kmcpg_query_device calls so far: slot = ws_calls & 1 (when the second slot may be used)
  uint64_t k1_codes_direct = 0, k1_codes_expanded = 0;  // packed batches: k-mer kernels on the codes / on text expanded first (kmcpg_k1_codes_batches)
  int ws_last = 0;             // slot of the last kmcpg_query_device call (kmcpg_last_gathered_bytes reads its counters)
  hipStream_t k1_stream = nullptr;  // experiment (KMCPG_K1_STREAM=1): the k-mer kernels on a high-priority stream of the handle's own
  hipEvent_t cobs_ev = nullptr;  // end of the last call's COBS kernels: the next call's COBS kernels wait for it
  bool cobs_ev_valid = false;
  hipEvent_t fin_ev = nullptr;   // K3's scratch (w_fin_cnt, w_fin_sums) has one user at a time, whatever the k-mer slots do
  bool fin_ev_valid = false;
  bool synthetic = false;
  // in-process multi-GPU front handle (kmcpg_open_devices): metadata only itself, one resident shard handle per device
  std::vector<kmcpg_db*> shards;
  kmcpg::Exchange* exchange = nullptr;  // RCCL gather of the shards' hit lists (exchange.cpp); nullptr = host merge
  std::string exchange_why;             // why not, when nullptr
  // paged handle (kmcpg_open_paged): metadata only itself; every batch is searched against the index one shard at a time
  // (paged_passes shards, the same partition kmcpg_open makes for shard_count = paged_passes); the shard searched last stays
  // resident and is the first one of the next batch
  std::string db_dir;
  int paged_passes = 0, paged_device = 0, paged_rank = -1;
  kmcpg_db* paged_resident = nullptr;
  std::mutex paged_mu;
  uint64_t paged_uploads = 0;  // shards made resident so far (tests / logs)
  uint64_t paged_reserve = 0;  // HBM plan_passes() kept free beside the largest shard: what a batch's workspace may take
  // optional HIP-event timing of the last kmcpg_query_device call
  int profiling = 0;  // 1: HIP-event timing of the kernels; 2: + count the row loads k2_cobs issues
  // K3 (device half of finalize): Header.Sizes of every global column on the device, per-read counters and scan scratch
  uint64_t* d_col_size = nullptr;
  kmcpg::DevBuf<uint32_t> w_fin_cnt;
  kmcpg::DevBuf<uint64_t> w_fin_sums;
  std::vector<kmcpg::FprBoundTable> fpr_bounds;  // -f bound tables (query.cpp fpr_bound): one per (max_fpr, size), never rewritten
  hipEvent_t ev[16] = {};   // ring of 4 calls x (start, COBS start, COBS done, k-mers done)
  uint64_t ev_calls = 0;    // profiled calls so far
};


namespace kmcpg {

inline kmcpg_params default_params() {
  kmcpg_params p{};
  p.min_qlen = 30;
  p.min_matched = 10;
  p.min_qcov = 0.55;
  p.min_tcov = 0;
  p.max_fpr = 0.01;
  p.dedup_threshold = 256;
  return p;
}

// std::vector whose resize() leaves trivially-constructible elements uninitialised (no 60-MB memset per batch)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = NoInitAlloc<U>;
  };
  template <class U, class... A>
  void construct(U* q, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)q) U;
    else ::new ((void*)q) U(std::forward<A>(a)...);
  }
};
// ... and whose storage starts on a cache line.  A kmcpg_match is 56 bytes (seven 8-byte words, finalize.cpp asserts it), so
// records straddle lines; the host half writes them with 8-byte streaming stores, which only need 8-byte alignment — the
// aligned start merely keeps the first write-combining buffer of an array whole (finalize.cpp)
template <class T>
struct LineAlloc : NoInitAlloc<T> {
  template <class U>
  struct rebind {
    using other = LineAlloc<U>;
  };
  T* allocate(size_t n) {
    void* q = nullptr;
    if (posix_memalign(&q, 64, std::max<size_t>(64, n * sizeof(T))) != 0) throw std::bad_alloc();
    return (T*)q;
  }
  void deallocate(T* q, size_t) { free(q); }
};
typedef std::vector<kmcpg_match, LineAlloc<kmcpg_match>> MatchVec;

struct ResultOwner {
  std::vector<int32_t> qlen, qkmers, ksize;
  std::vector<uint64_t> offs;
  MatchVec matches;
  // compact results (kmcpg_search_batch_pairs / kmcpg_wait_pairs): the final (column, count) pairs of every read, in the order its
  // Match records would have, instead of the records themselves
  std::vector<kmcpg_pair, NoInitAlloc<kmcpg_pair>> pairs;
  bool pairs_mode = false;
};
// set for the duration of a kmcpg_*_pairs call on the calling thread: results shaped meanwhile (result_owner_shape) collect pairs
extern thread_local bool tl_pairs_mode;
// finalize.cpp: a result assembled piece by piece (kmcpg_search_batch cuts large batches into pieces, host.cpp)
ResultOwner* result_owner_take();
void result_owner_give(ResultOwner* o);
void result_owner_shape(ResultOwner* o, uint32_t n_reads);
// trusted: the list is what K2 + K3 made of this very batch with these very params (the library's own pipelines) — thresholds, -T, the
// -f bound of short queries and the order of segments up to K3_WG_CAP hold by construction
int finalize_grouped_into(const kmcpg_db* db, const kmcpg_pair* pairs, const uint64_t* read_offs, const int32_t* qkmers, const int32_t* qlen, uint32_t n_reads,
                          const kmcpg_params& p, ResultOwner* o, uint32_t read_base, uint64_t match_base, uint64_t* kept_out, bool trusted = false,
                          int32_t bound_n = 0);
int finalize_grouped_trusted(const kmcpg_db* db, const kmcpg_pair* pairs, const uint64_t* read_offs, const int32_t* qkmers, const int32_t* qlen, uint32_t n_reads,
                             const kmcpg_params& p, kmcpg_result* out, int32_t bound_n);
// bound_n: what the kmcpg_query_device call(s) that produced the list actually did — every query of up to bound_n k-mers had the -f bound
// applied on the device, whichever kernel form served it (0: no bound table in that call).  kmcpg_query_device leaves the value of its
// call in tl_query_bound_n on the calling thread; host.cpp keeps it with the batch (Lane::bound_n) and hands it to the finalizer, which
// takes a compact segment as final only for n <= bound_n — it does not look at the environment again.
extern thread_local int32_t tl_query_bound_n;
// batches of whole genomes (segment path of the k-mer stage): the ones whose k-mer kernels run beside the previous batch's COBS kernel by
// default — second workspace slot (query.cpp pick_slot) and second kernel stream (host.cpp enqueue; bench.py does the same with its streams)
bool whole_genome_batch(const kmcpg_db* db, uint32_t max_read_len, bool paired);
// KMCPG_FPR_BOUND (default on): K2 leaves out counts that cannot pass -f for queries of up to 512 (1024) k-mers (query.cpp fpr_bound)
inline bool fpr_bound_enabled() {
  const char* e = getenv("KMCPG_FPR_BOUND");
  return !(e && atoi(e) == 0);
}
constexpr int kFprBoundAlways = 512;  // queries of up to this many k-mers are covered by the bound table whatever the batch holds
void result_publish(ResultOwner* o, uint32_t n_reads, int k_used, kmcpg_result* out);
// query.cpp: kmcpg_query_device with a prologue run under the handle's enqueue lock, in front of the batch's first kernel
int query_device_after(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, const uint8_t* d_seqs2, const uint64_t* d_offs2, uint32_t n_reads,
                       uint64_t total_bases, uint32_t max_read_len, const kmcpg_params* params, kmcpg_hit* d_hits, uint64_t hit_cap, uint64_t* d_counters,
                       int32_t* d_qkmers, int32_t* d_qlen, void* stream, const std::function<int()>* prologue);
void result_records_to_pairs(ResultOwner* o);  // finalize.cpp: a result that holds records -> the pairs of a compact result

}  // namespace kmcpg
