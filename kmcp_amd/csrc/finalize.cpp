// finalize.cpp — the host half: float64 thresholds, FPR, Match values, sorting (util-db-search.go:7471-7489, :260-345).
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
      return o;
    }
  }
  return new ResultOwner();
}

void give_owner(ResultOwner* o) {
  const size_t bytes = o->matches.capacity() * sizeof(kmcpg_match) + (o->qlen.capacity() + o->qkmers.capacity() + o->ksize.capacity()) * 4 + o->offs.capacity() * 8;
  {
    std::lock_guard<std::mutex> g(g_owner_mu);
    if (g_owner_pool.size() < 4 && bytes <= (1ull << 30)) {
      g_owner_pool.push_back(o);
      return;
    }
  }
  delete o;
}

struct OwnerReturn {
  void operator()(ResultOwner* o) const { give_owner(o); }
};

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

}  // namespace

extern "C" int kmcpg_finalize(const kmcpg_db* db, const kmcpg_hit* hits, uint64_t n_hits, const int32_t* qkmers, const int32_t* qlen, uint32_t n_reads,
                              const kmcpg_params* params, kmcpg_result* out) {
  if (!db || !out || (!hits && n_hits) || (n_reads && (!qkmers || !qlen))) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  const kmcpg_params p = params ? *params : default_params();
  std::unique_ptr<ResultOwner, OwnerReturn> o(take_owner());
  o->qlen.assign(qlen, qlen + n_reads);
  o->qkmers.assign(qkmers, qkmers + n_reads);
  const int k_used = p.k > 0 ? p.k : db->info.k;
  o->ksize.assign(n_reads, k_used);
  // scratch of this thread, kept between calls (a caller thread finalizes batch after batch)
  static thread_local std::vector<uint64_t> start, cur, per_read;
  static thread_local std::vector<kmcpg_hit, NoInitAlloc<kmcpg_hit>> sorted;
  // bucket hits by read (counting sort), then order each bucket by column
  start.assign((size_t)n_reads + 1, 0);
  for (uint64_t i = 0; i < n_hits; i++) {
    if (hits[i].read >= n_reads) return kmcpg_fail(KMCPG_EINVAL, "hit %llu names read %u of %u", (unsigned long long)i, hits[i].read, n_reads);
    if (hits[i].col >= db->col_block.size()) return kmcpg_fail(KMCPG_EINVAL, "hit names column %u of %zu", hits[i].col, db->col_block.size());
    start[hits[i].read + 1]++;
  }
  for (uint32_t r = 0; r < n_reads; r++) start[r + 1] += start[r];
  sorted.resize(n_hits);
  cur.assign(start.begin(), start.end() - 1);
  for (uint64_t i = 0; i < n_hits; i++) sorted[cur[hits[i].read]++] = hits[i];
  // FPR rows of the NumKmers values present (a handful for short reads), fetched once so that the workers below never lock
  QueryFpr* F = db->fpr.get();
  std::unordered_map<int, const std::vector<double>*> fpr_rows;
  int last_n = -1;  // reads of one batch mostly share their NumKmers: skip the map for runs of the same value
  for (uint32_t r = 0; r < n_reads; r++) {
    const int n = qkmers[r];
    if (n == last_n || n <= 0 || n > QueryFpr::kCachedMaxN || start[r + 1] == start[r]) continue;
    last_n = n;
    if (!fpr_rows.count(n)) fpr_rows.emplace(n, F->ensure_row(n));
  }
  // Reads are independent: contiguous ranges of reads per worker thread.  A hit yields at most one match, so worker w writes
  // its matches straight into the result array from position start[lo_w] on; the ranges are closed up afterwards.
  const int workers = (int)std::max<uint64_t>(1, std::min<uint64_t>(8, n_hits / 32768));
  o->matches.resize(n_hits);
  kmcpg_match* const mbase = o->matches.data();
  per_read.assign((size_t)n_reads, 0);
  uint64_t* const per_read_p = per_read.data();
  const uint64_t* const start_p = start.data();
  const kmcpg_hit* const sorted_p = sorted.data();
  std::vector<uint64_t> wcount((size_t)workers, 0);
  auto work = [&, mbase, per_read_p, start_p, sorted_p](int w) {
    const uint32_t lo = (uint32_t)((uint64_t)n_reads * w / workers), hi = (uint32_t)((uint64_t)n_reads * (w + 1) / workers);
    uint64_t pos = start_p[lo];
    int row_n = -1;
    const std::vector<double>* row_of_n = nullptr;
    for (uint32_t r = lo; r < hi; r++) {
      const uint64_t first = pos;
      const int n = qkmers[r];
      const double nh = (double)n;
      const double thr = nh * p.min_qcov;
      const std::vector<double>* row = nullptr;
      if (start_p[r + 1] > start_p[r] && n > 0 && n <= QueryFpr::kCachedMaxN) {
        if (n != row_n) {
          row_n = n;
          row_of_n = fpr_rows.find(n)->second;
        }
        row = row_of_n;
      }
      for (uint64_t i = start_p[r]; i < start_p[r + 1]; i++) {
        const kmcpg_hit& h = sorted_p[i];
        const int count = (int)h.count;
        if (count < p.min_matched) continue;
        const double c = (double)count;
        if (!(c > thr)) continue;
        const BlockMeta& b = db->blocks[db->col_block[h.col]];
        const uint32_t ci = h.col - b.col_base;
        const double nt = (double)b.h.sizes[ci];
        const double T = c / nt;
        if (!(T >= p.min_tcov)) continue;
        const double fpr = row ? (*row)[(size_t)std::min(count, n)] : F->get(n, count);
        if (!(fpr <= p.max_fpr)) continue;
        kmcpg_match m{};
        m.col = h.col;
        m.target_idx = b.h.indices[ci];
        m.gsize = b.h.gsizes[ci];
        m.mkmers = count;
        m.fpr = fpr;
        m.qcov = c / nh;
        m.tcov = T;
        m.jacc = c / (nh + nt - c);
        mbase[pos++] = m;
      }
      uint64_t cnt = pos - first;
      if (cnt > 1 && !p.do_not_sort) {
        const int sb = p.sort_by;
        std::sort(mbase + first, mbase + pos, [sb](const kmcpg_match& x, const kmcpg_match& y) { return match_less(x, y, sb); });
      } else if (cnt > 1) {
        std::sort(mbase + first, mbase + pos, [](const kmcpg_match& x, const kmcpg_match& y) { return x.col < y.col; });
      }
      if (cnt > 0 && p.top_n_scores > 0 && !p.do_not_sort) {  // --keep-top-scores (:285-311), including its [:i+1]
        int nn = 0;
        uint64_t i = 0;
        double pscore = 1024;
        for (; i < cnt; i++) {
          const kmcpg_match& m = mbase[first + i];
          const double score = p.sort_by == 1 ? m.tcov : (p.sort_by == 2 ? m.jacc : m.qcov);
          if (score < pscore) {
            nn++;
            if (nn > p.top_n_scores) break;
            pscore = score;
          }
        }
        if (i >= cnt) i = cnt - 1;
        pos = first + i + 1;
      }
      per_read_p[r] = pos - first;
    }
    wcount[(size_t)w] = pos - start_p[lo];
  };
  if (workers == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int w = 0; w < workers; w++) th.emplace_back(work, w);
    for (auto& t : th) t.join();
  }
  uint64_t total = 0;
  for (int w = 0; w < workers; w++) {  // close the gaps the filters left between the workers' ranges
    const uint32_t lo = (uint32_t)((uint64_t)n_reads * w / workers);
    if (start[lo] != total && wcount[(size_t)w]) memmove(mbase + total, mbase + start[lo], wcount[(size_t)w] * sizeof(kmcpg_match));
    total += wcount[(size_t)w];
  }
  o->matches.resize(total);
  o->offs.resize((size_t)n_reads + 1);
  o->offs[0] = 0;
  for (uint32_t r = 0; r < n_reads; r++) o->offs[r + 1] = o->offs[r] + per_read[r];
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

extern "C" void kmcpg_result_free(kmcpg_result* r) {
  if (!r || !r->owner) return;
  give_owner((ResultOwner*)r->owner);
  memset(r, 0, sizeof *r);
}

