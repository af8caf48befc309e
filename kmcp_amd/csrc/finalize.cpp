// finalize.cpp — the host half: float64 thresholds, FPR, Match values, sorting (util-db-search.go:7471-7489, :260-345).
#include <hip/hip_runtime.h>
#include <math.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
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

// ------------------------------------------------------------------------------------------------
// host half: thresholds that need float64, Match values, sorting (util-db-search.go:7471-7489, :260-345)
// ------------------------------------------------------------------------------------------------
namespace {

// Results are tens of MB per batch; handing freshly mmap'ed (page-faulting) vectors to every call costs more than filling
// them, so kmcpg_result_free parks a few owners here with their capacity and kmcpg_finalize takes them back.
std::mutex g_owner_mu;
std::vector<ResultOwner*> g_owner_pool;

ResultOwner* take_owner() {
  {
    std::lock_guard<std::mutex> g(g_owner_mu);
    if (!g_owner_pool.empty()) {
      ResultOwner* o = g_owner_pool.back();
      g_owner_pool.pop_back();
      o->pairs_mode = false;
      return o;
    }
  }
  return new ResultOwner();
}

void give_owner(ResultOwner* o) {
  const size_t bytes = o->matches.capacity() * sizeof(kmcpg_match) + o->pairs.capacity() * sizeof(kmcpg_pair) + (o->qlen.capacity() + o->qkmers.capacity() + o->ksize.capacity()) * 4 + o->offs.capacity() * 8;
  {
    std::lock_guard<std::mutex> g(g_owner_mu);
    // (a database full of close relatives returns hundreds of matches per read: 1.7 GB of records for a batch of 131 072 reads;
    // handing such a buffer back to the allocator means page-faulting it in again for the next batch, 0.2 s of every 0.4 s)
    if (g_owner_pool.size() < 4 && bytes <= (4ull << 30)) {
      g_owner_pool.push_back(o);
      return;
    }
  }
  delete o;
}

struct OwnerReturn {
  void operator()(ResultOwner* o) const { give_owner(o); }
};

// One record into the result array with streaming stores (seven 8-byte pieces; consecutive records fill whole cache lines in
// the write-combining buffers), so that the gigabyte of records a match-heavy batch produces is written once instead of read
// (for ownership) and written.
inline void store_record(kmcpg_match* dst, const kmcpg_match& src) {
#if defined(__x86_64__) && defined(__SSE2__)
  static_assert(sizeof(kmcpg_match) == 56 && alignof(kmcpg_match) == 8, "seven 8-byte pieces");
  long long w[7];
  memcpy(w, &src, sizeof w);
  long long* d = reinterpret_cast<long long*>(dst);
  for (int i = 0; i < 7; i++) _mm_stream_si64(d + i, w[i]);
#else
  *dst = src;
#endif
}
inline void records_visible() {
#if defined(__SSE2__)
  _mm_sfence();
#endif
}

inline bool is_tombstone(const kmcpg_hit& h) { return h.read == 0xffffffffu && h.col == 0xffffffffu; }

bool match_less(const kmcpg_match& x, const kmcpg_match& y, int sort_by) {
  double s1, s2, t1, t2;
  switch (sort_by) {  // Matches.Less / SortByTCov.Less / SortByJacc.Less (:105-145)
    case 1: s1 = x.tcov; s2 = y.tcov; t1 = x.mkmers; t2 = y.mkmers; break;
    case 2: s1 = x.jacc; s2 = y.jacc; t1 = x.mkmers; t2 = y.mkmers; break;
    default: s1 = x.qcov; s2 = y.qcov; t1 = x.tcov; t2 = y.tcov; break;
  }
  if (s1 != s2) return s1 > s2;
  if (t1 != t2) return t1 > t2;
  return x.col < y.col;  // deterministic tie-break; the reference's order among exact ties is arbitrary
}

// Host cores this process may use: the affinity mask capped by the cgroup CPU quota (a GPU box shows 256 cores and grants 16).
unsigned usable_cpus() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long long period = 0;
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1ll, atoll(q) / period));
    fclose(f);
  } else {
    long long quota = -1, period = 0;
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(g, "%lld", &quota) != 1) quota = -1;
      fclose(g);
    }
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(g, "%lld", &period) != 1) period = 0;
      fclose(g);
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1ll, quota / period));
  }
  return n;
}

// The workers of kmcpg_finalize: ONE set of threads per process, alive for its lifetime.  kmcpg_finalize is called from every
// waiter at once (the CLI's searcher threads, the shim's flushers) and has three parallel phases per call; starting fresh
// threads for each of them oversubscribed the host (16 new threads x 3 phases x every concurrent caller) and threw away the
// workers' thread-local scratch vectors with every call.  Here a call posts a job (a range of indices), the pool's threads and
// the caller itself take indices until none is left; concurrent callers share the same threads, so the number of runnable
// finalize threads stays at the pool size plus the callers whatever the load.
class WorkerPool {
 public:
  static WorkerPool& get() {
    static WorkerPool* p = new WorkerPool();  // never destroyed: its threads may outlive static destructors at exit
    return *p;
  }
  int width() const { return (int)threads_ + 1; }
  template <class F>
  void parallel_for(int n, F&& fn) {
    if (n <= 0) return;
    if (n == 1 || threads_ == 0) {
      for (int i = 0; i < n; i++) fn(i);
      return;
    }
    Job job;
    job.n = n;
    job.fn = [&fn](int i) { fn(i); };
    {
      std::lock_guard<std::mutex> g(mu_);
      jobs_.push_back(&job);
    }
    cv_.notify_all();
    work(&job);  // the caller takes indices too
    {
      std::lock_guard<std::mutex> g(mu_);  // nobody new may pick the job up
      jobs_.erase(std::remove(jobs_.begin(), jobs_.end(), &job), jobs_.end());
    }
    std::unique_lock<std::mutex> lk(job.mu);
    job.cv.wait(lk, [&] { return job.done == job.n && job.refs == 0; });
  }

 private:
  struct Job {
    int n = 0;
    std::function<void(int)> fn;
    std::atomic<int> next{0};
    std::mutex mu;
    std::condition_variable cv;
    int done = 0, refs = 0;  // under mu
  };
  WorkerPool() {
    unsigned t = std::min(16u, usable_cpus());
    if (const char* e = getenv("KMCPG_FINALIZE_POOL")) t = (unsigned)std::max(1, std::min(atoi(e), 64));
    threads_ = t > 1 ? t - 1 : 0;
    for (unsigned i = 0; i < threads_; i++) std::thread([this] { loop(); }).detach();
  }
  void work(Job* j) {
    int did = 0;
    for (;;) {
      const int i = j->next.fetch_add(1);
      if (i >= j->n) break;
      j->fn(i);
      did++;
    }
    if (did) {
      std::lock_guard<std::mutex> g(j->mu);
      j->done += did;
      if (j->done == j->n) j->cv.notify_all();
    }
  }
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      Job* j = nullptr;
      for (Job* c : jobs_)
        if (c->next.load() < c->n) {
          j = c;
          break;
        }
      if (!j) {
        cv_.wait(lk);
        continue;
      }
      {
        std::lock_guard<std::mutex> g(j->mu);
        j->refs++;
      }
      lk.unlock();
      work(j);
      {
        std::lock_guard<std::mutex> g(j->mu);
        j->refs--;
        j->cv.notify_all();  // last touch of the job: its owner may destroy it once we let go of j->mu
      }
      lk.lock();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Job*> jobs_;
  unsigned threads_ = 0;
};

}  // namespace

extern "C" int kmcpg_finalize(const kmcpg_db* db, const kmcpg_hit* hits, uint64_t n_hits, const int32_t* qkmers, const int32_t* qlen, uint32_t n_reads,
                              const kmcpg_params* params, kmcpg_result* out) {
  if (!db || !out || (!hits && n_hits) || (n_reads && (!qkmers || !qlen))) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  const kmcpg_params p = params ? *params : default_params();
  const bool timing = getenv("KMCPG_FIN_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_0 = now(), t_1 = 0, t_2 = 0, t_3 = 0, t_4 = 0;
  std::unique_ptr<ResultOwner, OwnerReturn> o(take_owner());
  o->qlen.assign(qlen, qlen + n_reads);
  o->qkmers.assign(qkmers, qkmers + n_reads);
  const int k_used = p.k > 0 ? p.k : db->info.k;
  o->ksize.assign(n_reads, k_used);
  // Reads are independent, so the batch is cut into R contiguous ranges of reads (up to four per worker):
  //  A. every worker takes a slice of the hit list (it arrives in no particular order) and counts its hits per range;
  //  B. ... and scatters them into the ranges' areas of `parted`;
  //  C. one range at a time per worker: counting sort of its hits by read, float64 thresholds, Match values, sort per read.  A hit
  //     yields at most one match, so the worker writes its matches straight into the result array from the position of its
  //     range's first hit on; the gaps the filters leave are closed afterwards.
  static thread_local std::vector<kmcpg_hit, NoInitAlloc<kmcpg_hit>> parted;
  static thread_local std::vector<uint64_t> per_read;
  WorkerPool& pool = WorkerPool::get();
  const uint64_t w_cap = (uint64_t)pool.width();
  int W = (int)std::max<uint64_t>(1, std::min<uint64_t>(w_cap, n_hits / 32768));
  if ((uint64_t)W > n_reads) W = n_reads ? (int)n_reads : 1;
  if (const char* e = getenv("KMCPG_FINALIZE_THREADS")) W = std::max(1, std::min(atoi(e), 64));
  if ((uint64_t)W > std::max<uint32_t>(1, n_reads)) W = (int)std::max<uint32_t>(1, n_reads);
  // ranges of 2^shift reads (a shift, not a division, per hit in phases A and B), up to four per worker: the pool hands them
  // out one at a time, so uneven ranges even out, and a range's hits (phase C sorts them by read) stay cache-sized
  uint64_t range_hits = 131072;
  if (const char* e = getenv("KMCPG_FIN_RANGE_HITS")) range_hits = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));
  const uint64_t r_max = W > 1 ? std::min<uint64_t>(2048, std::max<uint64_t>(4ull * (uint64_t)W, n_hits / range_hits)) : 1ull;
  int shift = 0;
  while (shift < 32 && (((uint64_t)n_reads + (1ull << shift) - 1) >> shift) > r_max) shift++;
  const int R = (int)std::max<uint64_t>(1, ((uint64_t)n_reads + (1ull << shift) - 1) >> shift);
  auto range_lo = [&](int w) { return (uint32_t)std::min<uint64_t>((uint64_t)w << shift, n_reads); };
  const size_t n_cols = db->col_block.size();
  parted.resize(n_hits);
  per_read.assign((size_t)n_reads, 0);
  // thread_local objects are per thread: the workers get at this thread's buffers through plain pointers
  kmcpg_hit* const parted_p = parted.data();
  o->matches.resize(n_hits);
  std::vector<uint64_t> cnt((size_t)W * R, 0);  // cnt[slice a][range b]
  std::atomic<int> bad{0};
  auto run = [&](auto&& fn) { pool.parallel_for(W, fn); };  // W pieces of work on the process-wide workers (+ this thread)
  auto slice = [&](int a, uint64_t* lo, uint64_t* hi) {
    *lo = n_hits * (uint64_t)a / (uint64_t)W;
    *hi = n_hits * (uint64_t)(a + 1) / (uint64_t)W;
  };
  const double t_00 = now();
  run([&](int a) {
    uint64_t lo, hi;
    slice(a, &lo, &hi);
    uint64_t* c = cnt.data() + (size_t)a * R;
    for (uint64_t i = lo; i < hi; i++) {
      if (hits[i].read >= n_reads || hits[i].col >= n_cols) {
        if (is_tombstone(hits[i])) continue;  // K2's stand-in for a set bit outside every column (k2_cobs.hip, epilogue): skipped
        bad.store(1);
        return;
      }
      c[(uint64_t)hits[i].read >> shift]++;
    }
  });
  if (bad.load()) {
    for (uint64_t i = 0; i < n_hits; i++) {
      if (is_tombstone(hits[i])) continue;
      if (hits[i].read >= n_reads) return kmcpg_fail(KMCPG_EINVAL, "hit %llu names read %u of %u", (unsigned long long)i, hits[i].read, n_reads);
      if (hits[i].col >= n_cols) return kmcpg_fail(KMCPG_EINVAL, "hit names column %u of %zu", hits[i].col, n_cols);
    }
  }
  const double t_a = now();
  std::vector<uint64_t> range_start((size_t)R + 1, 0), pos((size_t)W * R, 0);
  for (int b2 = 0; b2 < R; b2++) {
    uint64_t p0 = range_start[(size_t)b2];
    for (int a = 0; a < W; a++) {
      pos[(size_t)a * R + b2] = p0;
      p0 += cnt[(size_t)a * R + b2];
    }
    range_start[(size_t)b2 + 1] = p0;
  }
  run([&](int a) {
    uint64_t lo, hi;
    slice(a, &lo, &hi);
    uint64_t* q = pos.data() + (size_t)a * R;
    kmcpg_hit* dst = parted_p;
    for (uint64_t i = lo; i < hi; i++)
      if (!is_tombstone(hits[i])) dst[q[(uint64_t)hits[i].read >> shift]++] = hits[i];
  });
  t_1 = now();
  // FPR rows of the NumKmers values present in the batch (a handful for short reads; one O(n) pass each, kept by the database
  // handle), fetched once so that the workers below never lock.  (Looked up per read here, not per hit in phase A: a random
  // access into qkmers for every hit was a third of that phase.)
  QueryFpr* F = db->fpr.get();
  // (the shared_ptrs keep the rows alive for this call whatever other callers make the cache do; queries of up to 4 096 k-mers —
  // every short read — are found through a flat table, longer ones through the map)
  constexpr int kFlat = 4096;
  std::vector<const std::vector<double>*> flat_rows((size_t)kFlat + 1, nullptr);
  std::unordered_map<int, FprRow> fpr_rows;
  if (n_hits) {
    for (uint32_t r = 0; r < n_reads; r++) {
      const int n = qkmers[r];
      if (n <= 0) continue;
      if (n <= kFlat && flat_rows[(size_t)n]) continue;
      if (n > kFlat && fpr_rows.count(n)) continue;
      const auto it = fpr_rows.emplace(n, F->ensure_row(n)).first;
      if (n <= kFlat) flat_rows[(size_t)n] = it->second.get();
    }
  }
  t_2 = now();
  kmcpg_match* const mbase = o->matches.data();
  uint64_t* const per_read_p = per_read.data();
  std::vector<uint64_t> wcount((size_t)R, 0);
  const kmcpg_db::ColMeta* const col_meta = db->col_meta.data();
  pool.parallel_for(R, [&](int w) {
    const uint32_t lo = range_lo(w), hi = range_lo(w + 1);
    const uint64_t h0 = range_start[(size_t)w], h1 = range_start[(size_t)w + 1];
    if (hi <= lo) return;
    // counting sort of the range's hits by read
    static thread_local std::vector<uint64_t> start;
    static thread_local std::vector<kmcpg_hit, NoInitAlloc<kmcpg_hit>> sorted;
    start.assign((size_t)(hi - lo) + 1, 0);
    const kmcpg_hit* ph = parted_p;
    for (uint64_t i = h0; i < h1; i++) start[(size_t)(ph[i].read - lo) + 1]++;
    for (uint32_t r = 0; r < hi - lo; r++) start[(size_t)r + 1] += start[r];
    sorted.resize(h1 - h0);
    {
      static thread_local std::vector<uint64_t> cur;
      cur.assign(start.begin(), start.end() - 1);
      for (uint64_t i = h0; i < h1; i++) sorted[cur[ph[i].read - lo]++] = ph[i];
    }
    const uint64_t* const start_p = start.data();
    const kmcpg_hit* const sorted_p = sorted.data();
    const double tw1 = timing && w == 0 ? now() : 0;
    uint64_t pos2 = h0;
    int row_n = -1;
    const std::vector<double>* row_of_n = nullptr;
    for (uint32_t r = lo; r < hi; r++) {
      const uint64_t s0 = start_p[r - lo], s1 = start_p[r - lo + 1];
      if (s0 == s1) continue;
      const uint64_t first = pos2;
      const int n = qkmers[r];
      const double nh = (double)n;
      const double thr = nh * p.min_qcov;
      const std::vector<double>* row = nullptr;
      if (n > 0) {
        if (n != row_n) {
          row_n = n;
          row_of_n = n <= kFlat ? flat_rows[(size_t)n] : fpr_rows.find(n)->second.get();
        }
        row = row_of_n;
      }
      // A read with a handful of hits (the usual case) builds its matches in place and sorts the records; a read with many
      // (a database full of close relatives: hundreds) builds them in a scratch array, orders 4-byte indices and writes every
      // 56-byte record once, in its final position.
      const bool in_place = s1 - s0 <= 8;
      static thread_local std::vector<kmcpg_match> tmp;
      if (!in_place && tmp.size() < s1 - s0) tmp.resize(s1 - s0);
      kmcpg_match* const dst = in_place ? mbase + first : tmp.data();
      uint64_t cnt2 = 0;
      int c_lo = INT32_MAX, c_hi = 0;
      for (uint64_t i = s0; i < s1; i++) {
        const kmcpg_hit& h = sorted_p[i];
        const int count = (int)h.count;
        if (count < p.min_matched) continue;
        const double c = (double)count;
        if (!(c > thr)) continue;
        const kmcpg_db::ColMeta cm = col_meta[h.col];
        const double nt = (double)cm.size;
        const double T = c / nt;
        if (!(T >= p.min_tcov)) continue;
        const double fpr = row ? QueryFpr::value(*row, n, count) : F->get(n, count);
        if (!(fpr <= p.max_fpr)) continue;
        kmcpg_match m{};
        m.col = h.col;
        m.target_idx = cm.tidx;
        m.gsize = cm.gsize;
        m.mkmers = count;
        m.fpr = fpr;
        m.qcov = c / nh;
        m.tcov = T;
        m.jacc = c / (nh + nt - c);
        dst[cnt2++] = m;
        c_lo = std::min(c_lo, count);
        c_hi = std::max(c_hi, count);
      }
      pos2 = first + cnt2;
      if (in_place) {
        if (cnt2 > 1) {
          if (!p.do_not_sort) {
            const int sb = p.sort_by;
            std::sort(mbase + first, mbase + pos2, [sb](const kmcpg_match& x, const kmcpg_match& y) { return match_less(x, y, sb); });
          } else {
            std::sort(mbase + first, mbase + pos2, [](const kmcpg_match& x, const kmcpg_match& y) { return x.col < y.col; });
          }
        }
      } else if (cnt2 > 0) {
        static thread_local std::vector<uint32_t> order, bucket;
        if (order.size() < cnt2) order.resize(cnt2);
        uint32_t* const ord = order.data();
        const kmcpg_match* const t = tmp.data();
        if (!p.do_not_sort && p.sort_by == 0 && (uint64_t)(c_hi - c_lo) < 4 * cnt2 + 64) {
          // -s qcov (the default): qcov = mkmers / n with one n per read, so the primary key is the integer mkmers (two
          // counts that differ give quotients more than an ulp apart) — a counting sort over the counts present, descending;
          // equal counts (a few per bucket) are then ordered by tcov descending, column ascending, as match_less does.
          const size_t nb = (size_t)(c_hi - c_lo) + 1;
          bucket.assign(nb + 1, 0);
          for (uint64_t i = 0; i < cnt2; i++) bucket[(size_t)(c_hi - t[i].mkmers) + 1]++;
          for (size_t b = 0; b < nb; b++) bucket[b + 1] += bucket[b];
          for (uint64_t i = 0; i < cnt2; i++) ord[bucket[(size_t)(c_hi - t[i].mkmers)]++] = (uint32_t)i;
          // bucket[b] is now the END of bucket b
          uint32_t b0 = 0;
          for (size_t b = 0; b < nb; b++) {
            const uint32_t b1 = bucket[b];
            auto before = [t](uint32_t x, uint32_t y) {
              if (t[x].tcov != t[y].tcov) return t[x].tcov > t[y].tcov;
              return t[x].col < t[y].col;
            };
            if (b1 - b0 > 16) {
              std::sort(ord + b0, ord + b1, before);
            } else {  // the usual bucket: a handful of equal counts
              for (uint32_t i = b0 + 1; i < b1; i++) {
                const uint32_t v = ord[i];
                uint32_t j = i;
                for (; j > b0 && before(v, ord[j - 1]); j--) ord[j] = ord[j - 1];
                ord[j] = v;
              }
            }
            b0 = b1;
          }
        } else {
          for (uint64_t i = 0; i < cnt2; i++) ord[i] = (uint32_t)i;
          if (!p.do_not_sort) {
            const int sb = p.sort_by;
            std::sort(ord, ord + cnt2, [t, sb](uint32_t x, uint32_t y) { return match_less(t[x], t[y], sb); });
          } else {
            std::sort(ord, ord + cnt2, [t](uint32_t x, uint32_t y) { return t[x].col < t[y].col; });
          }
        }
        for (uint64_t i = 0; i < cnt2; i++) store_record(mbase + first + i, t[ord[i]]);
      }
      if (cnt2 > 0 && p.top_n_scores > 0 && !p.do_not_sort) {  // --keep-top-scores (:285-311), including its [:i+1]
        int nn = 0;
        uint64_t i = 0;
        double pscore = 1024;
        for (; i < cnt2; i++) {
          const kmcpg_match& m = mbase[first + i];
          const double score = p.sort_by == 1 ? m.tcov : (p.sort_by == 2 ? m.jacc : m.qcov);
          if (score < pscore) {
            nn++;
            if (nn > p.top_n_scores) break;
            pscore = score;
          }
        }
        if (i >= cnt2) i = cnt2 - 1;
        pos2 = first + i + 1;
      }
      per_read_p[r] = pos2 - first;
    }
    records_visible();
    wcount[(size_t)w] = pos2 - h0;
    if (timing && w == 0) fprintf(stderr, "finalize worker 0: counting sort %.2f, matches %.2f ms (%llu hits)\n", tw1 - t_2, now() - tw1, (unsigned long long)(h1 - h0));
  });
  t_3 = now();
  uint64_t total = 0;
  for (int w = 0; w < R; w++) {  // close the gaps the filters left between the ranges
    if (range_start[(size_t)w] != total && wcount[(size_t)w]) memmove(mbase + total, mbase + range_start[(size_t)w], wcount[(size_t)w] * sizeof(kmcpg_match));
    total += wcount[(size_t)w];
  }
  o->matches.resize(total);
  o->offs.resize((size_t)n_reads + 1);
  o->offs[0] = 0;
  for (uint32_t r = 0; r < n_reads; r++) o->offs[r + 1] = o->offs[r] + per_read[r];
  t_4 = now();
  if (timing) fprintf(stderr, "finalize: prep %.2f count %.2f scatter %.2f (W %d R %d)\n", t_00 - t_0, t_a - t_00, t_1 - t_a, W, R);
  if (timing) fprintf(stderr, "finalize: bucket %.2f fprrows %.2f workers %.2f close %.2f ms\n", t_1 - t_0, t_2 - t_1, t_3 - t_2, t_4 - t_3);
  out->n_reads = n_reads;
  out->k = k_used;
  out->qlen = o->qlen.data();
  out->qkmers = o->qkmers.data();
  out->ksize = o->ksize.data();
  out->match_offs = o->offs.data();
  out->matches = o->matches.data();
  out->owner = o.release();
  return 0;
}

// The host half behind K3 (k3_finalize.hip): the hit list arrives grouped by read, filtered by -T and in final order, 8 bytes per
// match; what is left is the float64 arithmetic of a Match (util-db-search.go:7487-7489), the FPR column with its -f test
// (:7474-7478; for queries of up to 1 024 k-mers K2 has applied it already through the bound table — the running value of
// util-fpr.go:32-50 only ever falls, so "count >= smallest passing count" IS the test), --keep-top-scores (:285-311) and the
// columns' metadata.  Every threshold is applied again (a no-op on lists K2 + K3 produced) and the order of every segment is checked
// while it streams: a list from elsewhere is finalized correctly too — segments longer than K3 orders (K3_WG_CAP) and segments
// found out of order are sorted here.
namespace kmcpg {
thread_local bool tl_pairs_mode = false;
thread_local int32_t tl_query_bound_n = 0;
ResultOwner* result_owner_take() { return take_owner(); }
void result_owner_give(ResultOwner* o) { give_owner(o); }

// The work of kmcpg_finalize_grouped for the reads [read_base, read_base + n_reads) of a result that is being assembled in `o`:
// their records go to o->matches[match_base ...] (the vector grows here), their match COUNTS to o->offs[read_base + 1 + r] (the caller
// turns counts into offsets once every piece is in), qlen / qkmers / ksize to their places.  *kept = records written.
// kmcpg_search_batch cuts a large batch into pieces that follow each other through the GPU and lands them here one after the other,
// so that the copy and the expansion of a piece overlap the kernels of the next (host.cpp).
int finalize_grouped_into(const kmcpg_db* db, const kmcpg_pair* pairs, const uint64_t* read_offs, const int32_t* qkmers, const int32_t* qlen, uint32_t n_reads,
                          const kmcpg_params& p, ResultOwner* o, uint32_t read_base, uint64_t match_base, uint64_t* kept_out, bool trusted, int32_t bound_n) {
  const uint64_t n_pairs = read_offs[n_reads];
  const int final_n = trusted ? std::min<int>(bound_n, kFprBoundAlways) : 0;  // segments of queries up to this size are final as they stand
  if (n_pairs && !pairs) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (read_offs[0] != 0) return kmcpg_fail(KMCPG_EINVAL, "read_offs[0] must be 0");
  if (read_offs[(size_t)n_reads + 1] != 0)
    return kmcpg_fail(KMCPG_EINVAL, "%llu hit(s) named a read or column that does not exist", (unsigned long long)read_offs[(size_t)n_reads + 1]);
  const bool timing = getenv("KMCPG_FIN_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_0 = now();
  const int k_used = p.k > 0 ? p.k : db->info.k;
  if (n_reads) {
    memcpy(o->qlen.data() + read_base, qlen, (size_t)n_reads * sizeof(int32_t));
    memcpy(o->qkmers.data() + read_base, qkmers, (size_t)n_reads * sizeof(int32_t));
    std::fill(o->ksize.begin() + read_base, o->ksize.begin() + read_base + n_reads, k_used);
  }
  const bool as_pairs = o->pairs_mode;  // compact result: the surviving pairs themselves, no records
  if (as_pairs) o->pairs.resize(match_base + n_pairs);
  else o->matches.resize(match_base + n_pairs);
  const double t_1 = now();
  WorkerPool& pool = WorkerPool::get();
  // contiguous ranges of reads holding ~64 k matches each (the offsets are at hand: a binary search per boundary)
  const uint64_t per_range = 65536;
  const int R = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n_reads, 4096), (n_pairs + per_range - 1) / per_range));
  std::vector<uint32_t> lo_of((size_t)R + 1, n_reads);
  lo_of[0] = 0;
  for (int w = 1; w < R; w++) {
    const uint64_t target = n_pairs * (uint64_t)w / (uint64_t)R;
    lo_of[(size_t)w] = std::max<uint32_t>(lo_of[(size_t)w - 1], (uint32_t)(std::lower_bound(read_offs, read_offs + n_reads, target) - read_offs));
  }
  QueryFpr* F = db->fpr.get();
  const size_t n_cols = db->col_meta.size();
  const kmcpg_db::ColMeta* const col_meta = db->col_meta.data();
  kmcpg_match* const mbase = as_pairs ? nullptr : o->matches.data() + match_base;
  kmcpg_pair* const pbase = as_pairs ? o->pairs.data() + match_base : nullptr;
  uint64_t* const per_read = o->offs.data() + 1 + read_base;  // counts; the caller makes offsets of them
  std::vector<uint64_t> wcount((size_t)R, 0);
  std::atomic<int> bad{0};
  pool.parallel_for(R, [&](int w) {
    const uint32_t lo = lo_of[(size_t)w], hi = lo_of[(size_t)w + 1];
    uint64_t pos = lo < n_reads ? read_offs[lo] : n_pairs;
    const uint64_t pos0 = pos;
    int row_n = -1;
    FprRow row;  // held while in use: the cache may drop its own reference at any time
    static thread_local std::vector<kmcpg_match> tmp;
    for (uint32_t r = lo; r < hi; r++) {
      const uint64_t s0 = read_offs[r], s1 = read_offs[r + 1];
      per_read[r] = 0;
      if (s1 == s0) continue;
      if (s1 < s0 || s1 > n_pairs) {
        bad.store(1);
        return;
      }
      const int n = qkmers[r];
      const double nh = (double)n, thr = nh * p.min_qcov;
      if (n > 0 && n != row_n) {
        row_n = n;
        row = F->ensure_row(n);
      }
      const uint64_t m = s1 - s0;
      // K3 leaves segments above K3_WG_CAP unordered; a list from elsewhere (the public kmcpg_finalize_grouped) may hold shorter
      // ones out of order too: the records are checked against their predecessor while they stream and the segment is done again
      // through the host sort if one is out of place (never, on K3's output)
      bool host_sort = m > (uint64_t)K3_WG_CAP;
      // The library's own list of a short query, collected as pairs: every test below is a no-op on it (K2 applied -c, -t and the -f bound
      // with these params, K3 applied -T and the order) — the segment is final as it stands
      // (final_n is what the query call reported for THIS batch: the device applied the bound table to every query of up to that many
      // k-mers, plain or chunked form — not what the environment says now)
      if (as_pairs && !host_sort && n > 0 && n <= final_n && !(p.top_n_scores > 0 && !p.do_not_sort)) {
        memcpy(pbase + pos, pairs + s0, (size_t)m * sizeof(kmcpg_pair));
        per_read[r] = m;
        pos += m;
        continue;
      }
     again:
      // A handful of matches (the usual read): plain stores, straight into the result.  Many: the records are built in a scratch
      // array and then written in one tight loop of streaming stores — streaming stores interleaved with the loads and divisions
      // of the loop below leave half-filled write-combining buffers behind, which the memory system pays for with partial
      // writes (measured: 0.5 us per record instead of 0.03).
      const bool via_tmp = host_sort || (m > 8 && !as_pairs);  // (8-byte pairs go straight to their place)
      if (via_tmp && tmp.size() < m) tmp.resize(m);
      const uint64_t first = pos;
      uint64_t kept = 0;
      // --keep-top-scores while the records go out: matches arrive by descending score
      int nn = 0;
      double pscore = 1024;
      const bool top = p.top_n_scores > 0 && !p.do_not_sort;
      bool cut = false, out_of_order = false;
      kmcpg_match prev{};
      bool have_prev = false;
      // (after the --keep-top-scores cut the rest of an ordered segment is only looked at — the cut is right only if ALL of it is in order)
      for (uint64_t i = s0; i < s1; i++) {
        const kmcpg_pair h = pairs[i];
        if (h.col >= n_cols) {
          bad.store(1);
          return;
        }
        const int count = (int)h.count;
        if (count < p.min_matched) continue;
        const double c = (double)count;
        if (!(c > thr)) continue;
        const kmcpg_db::ColMeta cm = col_meta[h.col];
        const double nt = (double)cm.size;
        const double T = c / nt;
        if (!(T >= p.min_tcov)) continue;
        const double fpr = n > 0 ? QueryFpr::value(*row, n, count) : 1.0;
        if (!(fpr <= p.max_fpr)) continue;
        kmcpg_match mm{};
        mm.col = h.col;
        mm.target_idx = cm.tidx;
        mm.gsize = cm.gsize;
        mm.mkmers = count;
        mm.fpr = fpr;
        mm.qcov = c / nh;
        mm.tcov = T;
        mm.jacc = c / (nh + nt - c);
        if (host_sort) {
          tmp[kept++] = mm;
          continue;
        }
        if (!trusted) {  // (K3's own segments are in order by construction: the library's pipelines skip the comparison)
          if (have_prev && (p.do_not_sort ? mm.col < prev.col : match_less(mm, prev, p.sort_by))) {
            out_of_order = true;
            break;
          }
          prev = mm;
          have_prev = true;
        } else if (cut) {
          break;
        }
        if (cut) continue;
        if (via_tmp) tmp[kept] = mm;
        else if (as_pairs) pbase[first + kept] = h;
        else mbase[first + kept] = mm;
        kept++;
        if (top) {
          const double score = p.sort_by == 1 ? mm.tcov : (p.sort_by == 2 ? mm.jacc : mm.qcov);
          if (score < pscore) {
            nn++;
            if (nn > p.top_n_scores) cut = true;  // the reference keeps this one too (its [:i+1], :305-309) and stops
            pscore = score;
          }
        }
      }
      if (out_of_order) {
        host_sort = true;
        goto again;
      }
      if (via_tmp && !host_sort)
        for (uint64_t i = 0; i < kept; i++) store_record(mbase + first + i, tmp[i]);
      if (host_sort && kept) {
        const int sb = p.sort_by;
        if (!p.do_not_sort) std::sort(tmp.begin(), tmp.begin() + (ptrdiff_t)kept, [sb](const kmcpg_match& x, const kmcpg_match& y) { return match_less(x, y, sb); });
        else std::sort(tmp.begin(), tmp.begin() + (ptrdiff_t)kept, [](const kmcpg_match& x, const kmcpg_match& y) { return x.col < y.col; });
        uint64_t keep = kept;
        if (top) {
          uint64_t i = 0;
          for (; i < kept; i++) {
            const kmcpg_match& x = tmp[i];
            const double score = p.sort_by == 1 ? x.tcov : (p.sort_by == 2 ? x.jacc : x.qcov);
            if (score < pscore) {
              nn++;
              if (nn > p.top_n_scores) break;
              pscore = score;
            }
          }
          if (i >= kept) i = kept - 1;
          keep = i + 1;
        }
        if (as_pairs)
          for (uint64_t i = 0; i < keep; i++) pbase[first + i] = kmcpg_pair{tmp[i].col, (uint32_t)tmp[i].mkmers};
        else
          for (uint64_t i = 0; i < keep; i++) store_record(mbase + first + i, tmp[i]);
        kept = keep;
      }
      per_read[r] = kept;
      pos = first + kept;
    }
    records_visible();
    wcount[(size_t)w] = pos - pos0;
  });
  const double t_2 = now();
  if (timing) fprintf(stderr, "finalize_grouped: prep %.2f workers %.2f ms (R %d, %llu pairs)\n", t_1 - t_0, t_2 - t_1, R, (unsigned long long)n_pairs);
  if (bad.load()) return kmcpg_fail(KMCPG_EINVAL, "grouped hit list: offsets or columns out of range");
  uint64_t total = 0;
  for (int w = 0; w < R; w++) {  // close the gaps the filters / --keep-top-scores left between the ranges
    const uint64_t start = lo_of[(size_t)w] < n_reads ? read_offs[lo_of[(size_t)w]] : n_pairs;
    if (start != total && wcount[(size_t)w]) {
      if (as_pairs) memmove(pbase + total, pbase + start, wcount[(size_t)w] * sizeof(kmcpg_pair));
      else memmove(mbase + total, mbase + start, wcount[(size_t)w] * sizeof(kmcpg_match));
    }
    total += wcount[(size_t)w];
  }
  if (as_pairs) o->pairs.resize(match_base + total);
  else o->matches.resize(match_base + total);
  *kept_out = total;
  return 0;
}

// counts in o->offs[1 ..] -> offsets, and the result's pointers
void result_publish(ResultOwner* o, uint32_t n_reads, int k_used, kmcpg_result* out) {
  o->offs[0] = 0;
  for (uint32_t r = 0; r < n_reads; r++) o->offs[(size_t)r + 1] += o->offs[r];
  out->n_reads = n_reads;
  out->k = k_used;
  out->qlen = o->qlen.data();
  out->qkmers = o->qkmers.data();
  out->ksize = o->ksize.data();
  out->match_offs = o->offs.data();
  out->matches = o->matches.data();
  out->owner = o;
}

void result_owner_shape(ResultOwner* o, uint32_t n_reads) {
  o->pairs_mode = tl_pairs_mode;
  o->pairs.clear();
  o->qlen.resize(n_reads);
  o->qkmers.resize(n_reads);
  o->ksize.resize(n_reads);
  o->offs.assign((size_t)n_reads + 1, 0);
  o->matches.clear();
}
}  // namespace kmcpg

namespace kmcpg {
int finalize_grouped_trusted(const kmcpg_db* db, const kmcpg_pair* pairs, const uint64_t* read_offs, const int32_t* qkmers, const int32_t* qlen, uint32_t n_reads,
                             const kmcpg_params& p, kmcpg_result* out, int32_t bound_n) {
  std::unique_ptr<ResultOwner, OwnerReturn> o(take_owner());
  result_owner_shape(o.get(), n_reads);
  uint64_t kept = 0;
  if (int rc = finalize_grouped_into(db, pairs, read_offs, qkmers, qlen, n_reads, p, o.get(), 0, 0, &kept, true, bound_n)) return rc;
  result_publish(o.release(), n_reads, p.k > 0 ? p.k : db->info.k, out);
  return 0;
}
}  // namespace kmcpg

extern "C" int kmcpg_finalize_grouped(const kmcpg_db* db, const kmcpg_pair* pairs, const uint64_t* read_offs, const int32_t* qkmers, const int32_t* qlen,
                                      uint32_t n_reads, const kmcpg_params* params, kmcpg_result* out) {
  if (!db || !out || !read_offs || (n_reads && (!qkmers || !qlen))) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  const kmcpg_params p = params ? *params : default_params();
  std::unique_ptr<ResultOwner, OwnerReturn> o(take_owner());
  result_owner_shape(o.get(), n_reads);
  uint64_t kept = 0;
  if (int rc = finalize_grouped_into(db, pairs, read_offs, qkmers, qlen, n_reads, p, o.get(), 0, 0, &kept)) return rc;
  result_publish(o.release(), n_reads, p.k > 0 ? p.k : db->info.k, out);
  return 0;
}

// a result that holds records (a path that does not collect pairs natively: retries, host-merged lists, paged handles) -> pairs
namespace kmcpg {
void result_records_to_pairs(ResultOwner* o) {
  if (o->pairs_mode) return;
  const size_t n = o->matches.size();
  o->pairs.resize(n);
  for (size_t i = 0; i < n; i++) o->pairs[i] = kmcpg_pair{o->matches[i].col, (uint32_t)o->matches[i].mkmers};
  o->pairs_mode = true;
  // the records are dead from here on: a compact result must not carry a 56-byte copy of every match along (nor hand a consumer
  // records whose offsets mean pairs)
  MatchVec().swap(o->matches);
}
}  // namespace kmcpg

// The Match records of one query's pairs: the float64 values exactly as kmcpg_finalize_grouped writes them (util-db-search.go:7487-7489,
// util-fpr.go:32-50).  No threshold is applied: the pairs of a kmcpg_result_pairs have passed all of them.
extern "C" int kmcpg_expand_pairs(const kmcpg_db* db, int32_t qkmers, const kmcpg_pair* pairs, uint64_t n, kmcpg_match* out) {
  if (!db || (n && (!pairs || !out))) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (n == 0) return 0;
  const size_t n_cols = db->col_meta.size();
  const kmcpg_db::ColMeta* const col_meta = db->col_meta.data();
  const double nh = (double)qkmers;
  // the row of this NumKmers stays with the thread: consecutive queries of a batch mostly share it (and a formatter thread asks for
  // a few hundred thousand queries per second)
  // (keyed on the FPR table's process-unique id, not on the handle's address: a database opened later may live where a closed one did)
  static thread_local uint64_t row_of = 0;
  static thread_local int row_n = -1;
  static thread_local FprRow row;
  if (qkmers > 0 && (row_of != db->fpr->id() || row_n != qkmers)) {
    row = db->fpr->ensure_row(qkmers);
    row_of = db->fpr->id();
    row_n = qkmers;
  }
  for (uint64_t i = 0; i < n; i++) {
    const kmcpg_pair h = pairs[i];
    if (h.col >= n_cols) return kmcpg_fail(KMCPG_EINVAL, "pair %llu names column %u of %zu", (unsigned long long)i, h.col, n_cols);
    const kmcpg_db::ColMeta cm = col_meta[h.col];
    const double c = (double)h.count, nt = (double)cm.size;
    kmcpg_match mm{};
    mm.col = h.col;
    mm.target_idx = cm.tidx;
    mm.gsize = cm.gsize;
    mm.mkmers = (int32_t)h.count;
    mm.fpr = qkmers > 0 ? QueryFpr::value(*row, qkmers, (int)h.count) : 1.0;
    mm.qcov = c / nh;
    mm.tcov = c / nt;
    mm.jacc = c / (nh + nt - c);
    out[i] = mm;
  }
  return 0;
}

extern "C" void kmcpg_result_pairs_free(kmcpg_result_pairs* r) {
  if (!r || !r->owner) return;
  give_owner((ResultOwner*)r->owner);
  memset(r, 0, sizeof *r);
}

extern "C" void kmcpg_result_free(kmcpg_result* r) {
  if (!r || !r->owner) return;
  give_owner((ResultOwner*)r->owner);
  memset(r, 0, sizeof *r);
}

