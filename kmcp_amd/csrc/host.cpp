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
// whole pipeline on host buffers
// ------------------------------------------------------------------------------------------------
namespace {

struct RawBatch {
  std::vector<kmcpg_hit> hits;
  std::vector<int32_t> qk, ql;
};

int run_raw(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n, const kmcpg_params& p,
            RawBatch* rb) {
  rb->hits.clear();
  rb->qk.assign(n, 0);
  rb->ql.assign(n, 0);
  if (n == 0) return 0;
  const uint64_t tb1 = offs[n] - offs[0], tb2 = seqs2 ? offs2[n] - offs2[0] : 0;
  if (offs[0] != 0 || (seqs2 && offs2[0] != 0)) return kmcpg_fail(KMCPG_EINVAL, "offs[0] must be 0");
  uint32_t maxlen = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint64_t l = offs[i + 1] - offs[i];
    if (seqs2) l = std::max<uint64_t>(l, offs2[i + 1] - offs2[i]);
    if (l > 0x7fffffffULL) return kmcpg_fail(KMCPG_EUNSUPPORTED, "query longer than 2^31-1 bases");
    maxlen = std::max<uint32_t>(maxlen, (uint32_t)l);
  }
  {
    std::lock_guard<std::mutex> g(db->mu);
    KMCPG_USE_DEVICE(db);
    if (db->s_seqs.ensure(tb1 + 16) || db->s_offs.ensure(n + 1) || db->s_counter.ensure(2) || db->s_qk.ensure(n) || db->s_ql.ensure(n))
      return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    if (seqs2 && (db->s_seqs2.ensure(tb2 + 16) || db->s_offs2.ensure(n + 1))) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    HIPCHK(hipMemcpy(db->s_seqs.p, seqs, tb1, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db->s_offs.p, offs, (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (seqs2) {
      HIPCHK(hipMemcpy(db->s_seqs2.p, seqs2, tb2, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(db->s_offs2.p, offs2, (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
  }
  uint64_t cap = std::max<uint64_t>(db->s_hits.cap, (uint64_t)n * 8 + 1024);
  for (int attempt = 0; attempt < 3; attempt++) {
    {
      std::lock_guard<std::mutex> g(db->mu);
      if (db->s_hits.ensure(cap)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
    }
    int rc = kmcpg_query_device(db, db->s_seqs.p, db->s_offs.p, seqs2 ? db->s_seqs2.p : nullptr, seqs2 ? db->s_offs2.p : nullptr, n, tb1 + tb2, maxlen, &p,
                                db->s_hits.p, db->s_hits.cap, db->s_counter.p, db->s_qk.p, db->s_ql.p, nullptr);
    if (rc) return rc;
    uint64_t cnt = 0;
    HIPCHK(hipMemcpy(&cnt, db->s_counter.p, sizeof cnt, hipMemcpyDeviceToHost));  // synchronises the default stream
    if (cnt <= db->s_hits.cap) {
      rb->hits.resize(cnt);
      if (cnt) HIPCHK(hipMemcpy(rb->hits.data(), db->s_hits.p, cnt * sizeof(kmcpg_hit), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(rb->qk.data(), db->s_qk.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(rb->ql.data(), db->s_ql.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
      return 0;
    }
    cap = cnt + cnt / 4;  // buffer was too small: rerun with room for every hit
  }
  return kmcpg_fail(KMCPG_ENOMEM, "hit buffer overflow");
}

// all resident shards of a multi-device handle search the batch concurrently (one host thread per GPU); the hit lists are
// concatenated exactly as the reference concatenates the replies of its per-block workers (:946-964)
int run_raw_any(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n, const kmcpg_params& p,
                RawBatch* rb) {
  if (db->shards.empty()) return run_raw(db, seqs, offs, seqs2, offs2, n, p, rb);
  const size_t S = db->shards.size();
  std::vector<RawBatch> parts(S);
  std::vector<int> rcs(S, 0);
  std::vector<std::string> errs(S);
  std::vector<std::thread> th;
  for (size_t i = 0; i < S; i++)
    th.emplace_back([&, i] {
      rcs[i] = run_raw(db->shards[i], seqs, offs, seqs2, offs2, n, p, &parts[i]);
      if (rcs[i]) errs[i] = kmcpg_err_ref();  // thread-local message of the worker
    });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < S; i++)
    if (rcs[i]) return kmcpg_fail(rcs[i], "device %d: %s", db->shards[i]->opts.device, errs[i].c_str());
  rb->qk = parts[0].qk;  // every shard generates the same k-mers
  rb->ql = parts[0].ql;
  rb->hits.clear();
  for (auto& pt : parts) rb->hits.insert(rb->hits.end(), pt.hits.begin(), pt.hits.end());
  return 0;
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
    rc = kmcpg_open(db_dir, &so, &sh);
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
  *out = front;
  return 0;
}

extern "C" int kmcpg_search_batch(kmcpg_db* db, const uint8_t* seqs, const uint64_t* offs, const uint8_t* seqs2, const uint64_t* offs2, uint32_t n_reads,
                                  const kmcpg_params* params, kmcpg_result* out) {
  if (!db || !out || (n_reads && (!seqs || !offs))) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if ((seqs2 == nullptr) != (offs2 == nullptr)) return kmcpg_fail(KMCPG_EINVAL, "seqs2 and offs2 must be given together");
  if (db->opts.shard_count != 1)
    return kmcpg_fail(KMCPG_EINVAL, "kmcpg_search_batch needs the whole database: open it on one GPU or with kmcpg_open_devices; use kmcpg_query_device + kmcpg_finalize per shard");
  kmcpg_params p = params ? *params : default_params();
  memset(out, 0, sizeof *out);
  RawBatch rb;
  int rc;
  {
    std::lock_guard<std::mutex> api_guard(db->api_mu);
    rc = run_raw_any(db, seqs, offs, seqs2, offs2, n_reads, p, &rb);
  }
  if (rc) return rc;
  rc = kmcpg_finalize(db, rb.hits.data(), rb.hits.size(), rb.qk.data(), rb.ql.data(), n_reads, &p, out);
  if (rc) return rc;
  if (!(p.try_se && seqs2)) return 0;

  // --try-se (:831-850, :1001-1014): paired-end queries without a match are searched again with read 1, then read 2.
  // The retries skip the length gate (it is applied once, before k-mer generation) and reuse the mates' own k-mers.
  ResultOwner* o = (ResultOwner*)out->owner;
  for (int mate = 0; mate < 2; mate++) {
    std::vector<uint32_t> todo;
    for (uint32_t r = 0; r < n_reads; r++)
      if (o->offs[r + 1] == o->offs[r] && o->qkmers[r] > 0) todo.push_back(r);  // searched (>= MinMatched k-mers) but nothing found
    if (todo.empty()) break;
    const uint8_t* S = mate == 0 ? seqs : seqs2;
    const uint64_t* O = mate == 0 ? offs : offs2;
    std::vector<uint8_t> sub;
    std::vector<uint64_t> so(1, 0);
    for (uint32_t r : todo) {
      sub.insert(sub.end(), S + O[r], S + O[r + 1]);
      so.push_back(sub.size());
    }
    kmcpg_params q = p;
    q.min_qlen = 0;
    q.try_se = 0;
    RawBatch rb2;
    {
      std::lock_guard<std::mutex> api_guard(db->api_mu);
      rc = run_raw_any(db, sub.data(), so.data(), nullptr, nullptr, (uint32_t)todo.size(), q, &rb2);
    }
    if (rc) return rc;
    kmcpg_result r2;
    rc = kmcpg_finalize(db, rb2.hits.data(), rb2.hits.size(), rb2.qk.data(), rb2.ql.data(), (uint32_t)todo.size(), &q, &r2);
    if (rc) return rc;
    // splice the retried queries back in
    std::vector<uint64_t> noffs((size_t)n_reads + 1, 0);
    MatchVec nm;
    size_t t = 0;
    std::vector<char> stop(n_reads, 0);
    for (uint32_t r = 0; r < n_reads; r++) {
      if (t < todo.size() && todo[t] == r) {
        o->qlen[r] = r2.qlen[t];
        if (r2.qkmers[t] > 0) o->qkmers[r] = r2.qkmers[t];
        else stop[r] = 1;  // fewer than MinMatched k-mers in this mate: the reference returns here (:854-869)
        nm.insert(nm.end(), r2.matches + r2.match_offs[t], r2.matches + r2.match_offs[t + 1]);
        t++;
      } else {
        nm.insert(nm.end(), o->matches.begin() + (ptrdiff_t)o->offs[r], o->matches.begin() + (ptrdiff_t)o->offs[r + 1]);
      }
      noffs[r + 1] = nm.size();
    }
    kmcpg_result_free(&r2);
    o->matches.swap(nm);
    o->offs.swap(noffs);
    if (mate == 0)
      for (uint32_t r = 0; r < n_reads; r++)
        if (stop[r]) o->qkmers[r] = -o->qkmers[r] - 1;  // park: not retried with read 2
    out->matches = o->matches.data();
    out->match_offs = o->offs.data();
  }
  for (uint32_t r = 0; r < n_reads; r++)
    if (o->qkmers[r] < 0) o->qkmers[r] = -(o->qkmers[r] + 1);
  out->qlen = o->qlen.data();
  out->qkmers = o->qkmers.data();
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
  uint64_t* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, n * sizeof(uint64_t)));
  HIPCHK(hipMemcpy(d, hashes, n * sizeof(uint64_t), hipMemcpyHostToDevice));
  launch_plant(db->h_blockdev[(size_t)b.local_idx], col - b.col_base, db->info.num_hashes, d, n, nullptr);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipFree(d));
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
  uint64_t* d_idx = nullptr;
  uint8_t* d_out = nullptr;
  HIPCHK(hipMalloc((void**)&d_idx, n_rows * sizeof(uint64_t)));
  HIPCHK(hipMalloc((void**)&d_out, n_rows * b.h.row_bytes));
  HIPCHK(hipMemcpy(d_idx, row_idx, n_rows * sizeof(uint64_t), hipMemcpyHostToDevice));
  launch_gather_rows(b.d_rows, b.stride, b.h.row_bytes, d_idx, n_rows, d_out, nullptr);
  HIPCHK(hipMemcpy(out, d_out, n_rows * b.h.row_bytes, hipMemcpyDeviceToHost));
  HIPCHK(hipFree(d_idx));
  HIPCHK(hipFree(d_out));
  return 0;
}
