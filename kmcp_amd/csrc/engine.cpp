// engine.cpp — host side of libkmcpgpu.so: database residency in HBM, the batched query pipeline and
// the float64 post-processing, behind the C ABI of include/kmcp_gpu.h.
//
// Reference counterparts (kmcp/cmd/): NewUnikIndexDB / NewUnikIndex (util-db-search.go:648-743,
// 1196-1280), handleQuery (:763-1025), threshold+Match (:7415-7733), handleQuerySingleDB (:260-345).
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
#include "fpr.hpp"
#include "kernels.hpp"

using namespace kmcpg;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(KMCPG_EDEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// GPU work on a handle: refused for metadata-only handles (opts.device == -1)
#define KMCPG_USE_DEVICE(db)                                                                                      \
  do {                                                                                                            \
    if ((db)->opts.device < 0) return fail(KMCPG_EDEVICE, "metadata-only handle (device -1): no GPU work possible"); \
    HIPCHK(hipSetDevice((db)->opts.device));                                                                      \
  } while (0)

extern "C" const char* kmcpg_last_error(void) { return g_err.c_str(); }

// error sink shared with build.cpp
int kmcpg_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// ------------------------------------------------------------------------------------------------
// database object
// ------------------------------------------------------------------------------------------------
namespace {

struct BlockMeta {
  std::string path;
  UnikiHeader h;
  uint32_t col_base = 0;
  bool local = false;
  int local_idx = -1;
  uint32_t stride = 0;
  uint8_t* d_rows = nullptr;
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

}  // namespace

struct kmcpg_db {
  kmcpg_opts opts{};
  kmcpg_info info{};
  std::vector<BlockMeta> blocks;
  std::vector<int> local;  // global indices of resident blocks, in BlockDev order
  std::vector<BlockDev> h_blockdev;
  BlockDev* d_blockdev = nullptr;
  std::vector<SlotClass> classes;
  std::vector<uint32_t> col_block;  // global column -> block index
  std::unique_ptr<QueryFpr> fpr;
  std::mutex mu;      // guards the device workspace of one GPU-half call
  std::mutex api_mu;  // serialises the GPU halves of kmcpg_search_batch callers (they share the staging buffers); the host
                      // half (kmcpg_finalize) runs outside it, so two callers overlap one's finalize with the other's kernels
  // workspace of kmcpg_query_device
  DevBuf<uint64_t> w_hashes, w_scratch;
  DevBuf<int32_t> w_nk_raw, w_nk1, w_seg_cnt;
  DevBuf<uint32_t> w_long_list, w_long_meta, w_long_counts;  // long-query (split) path
  DevBuf<uint64_t> w_huge_info;                             // whole-genome queries: (read, n, offset)
  DevBuf<uint8_t> w_huge_temp;                              // hipCUB temporary storage
  // workspace of kmcpg_search_batch
  DevBuf<uint8_t> s_seqs, s_seqs2;
  DevBuf<uint64_t> s_offs, s_offs2, s_counter;
  DevBuf<kmcpg_hit> s_hits;
  DevBuf<int32_t> s_qk, s_ql;
  bool synthetic = false;
  // in-process multi-GPU front handle (kmcpg_open_devices): metadata only itself, one resident shard handle per device
  std::vector<kmcpg_db*> shards;
  // optional HIP-event timing of the last kmcpg_query_device call
  bool profiling = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ev_valid = false;
};

namespace {

uint32_t device_stride(uint32_t row_bytes) {
  // rows are padded so a row never straddles more memory lines than it must: powers of two up to 64 B,
  // multiples of 64 B above (1872 -> 1920).  The disk format is untouched (serialization.go:288-300).
  if (row_bytes <= 16) return 16;
  if (row_bytes <= 32) return 32;
  if (row_bytes <= 64) return 64;
  return (row_bytes + 63) / 64 * 64;
}

int lpr_for_stride(uint32_t stride) { return stride <= 64 ? 4 : (stride <= 256 ? 16 : 64); }

void magic_for(uint64_t d, uint64_t* hi, uint64_t* lo) {
  unsigned __int128 m = (~(unsigned __int128)0) / d + 1;  // wraps to 0 for d == 1: then x % 1 == 0 falls out
  *hi = (uint64_t)(m >> 64);
  *lo = (uint64_t)m;
}

// blocks are independent: greedy partition by bytes, largest first
void assign_shards(kmcpg_db* db) {
  const int S = db->opts.shard_count;
  std::vector<int> order(db->blocks.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  auto bytes = [&](int i) { return db->blocks[i].h.num_sigs * (uint64_t)db->blocks[i].h.row_bytes; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bytes(a) > bytes(b); });
  std::vector<uint64_t> load((size_t)S, 0);
  for (int i : order) {
    int best = 0;
    for (int s = 1; s < S; s++)
      if (load[s] < load[best]) best = s;
    load[best] += bytes(i);
    db->blocks[i].local = best == db->opts.shard_rank;
  }
}

int finish_open(kmcpg_db* db) {
  // BlockDev table + slot classes
  db->local.clear();
  db->h_blockdev.clear();
  db->classes.clear();
  db->info.n_blocks_local = 0;
  db->info.matrix_bytes_local = 0;
  db->info.row_bytes_sum_local = 0;
  for (size_t i = 0; i < db->blocks.size(); i++) {
    BlockMeta& b = db->blocks[i];
    if (!b.local) continue;
    b.local_idx = (int)db->local.size();
    db->local.push_back((int)i);
    BlockDev bd{};
    bd.rows = b.d_rows;
    bd.num_sigs = b.h.num_sigs;
    magic_for(b.h.num_sigs, &bd.magic_hi, &bd.magic_lo);
    bd.stride = b.stride;
    bd.row_bytes = b.h.row_bytes;
    bd.ncols = (uint32_t)b.h.names.size();
    bd.col_base = b.col_base;
    db->h_blockdev.push_back(bd);
    db->info.n_blocks_local++;
    db->info.matrix_bytes_local += b.h.num_sigs * (uint64_t)b.h.row_bytes;
    db->info.row_bytes_sum_local += b.h.row_bytes;
    const int lpr = lpr_for_stride(b.stride);
    SlotClass* cls = nullptr;
    for (auto& c : db->classes)
      if (c.lpr == lpr) cls = &c;
    if (!cls) {
      db->classes.push_back(SlotClass{});
      cls = &db->classes.back();
      cls->lpr = lpr;
    }
    const uint32_t tile_bytes = (uint32_t)lpr * 16u;
    const uint32_t tiles = (b.stride + tile_bytes - 1) / tile_bytes;
    for (uint32_t t = 0; t < tiles; t++) cls->slots.push_back(Slot{(uint32_t)b.local_idx, t});
  }
  if (db->opts.device >= 0 && !db->h_blockdev.empty()) {
    HIPCHK(hipMalloc((void**)&db->d_blockdev, db->h_blockdev.size() * sizeof(BlockDev)));
    HIPCHK(hipMemcpy(db->d_blockdev, db->h_blockdev.data(), db->h_blockdev.size() * sizeof(BlockDev), hipMemcpyHostToDevice));
  }
  for (auto& c : db->classes) {
    if (db->opts.device < 0) break;
    HIPCHK(hipMalloc((void**)&c.d_slots, c.slots.size() * sizeof(Slot)));
    HIPCHK(hipMemcpy(c.d_slots, c.slots.data(), c.slots.size() * sizeof(Slot), hipMemcpyHostToDevice));
  }
  db->col_block.clear();
  for (size_t i = 0; i < db->blocks.size(); i++)
    for (size_t c = 0; c < db->blocks[i].h.names.size(); c++) db->col_block.push_back((uint32_t)i);
  db->fpr.reset(new QueryFpr(db->info.fpr));
  return 0;
}

// Upload of the local blocks: file rows -> pinned staging -> temp device buffer -> padded HBM rows (k_repack).
// The matrix is cut into 64 MB chunks handed to a few loader threads; each owns a pinned buffer, a device staging buffer and
// a stream, so that preads (page cache or disk), PCIe copies and repacks of different chunks overlap.
struct UploadChunk {
  BlockMeta* b;
  uint64_t r0, nr;
};

int upload_blocks(kmcpg_db* db) {
  const uint64_t kChunkBytes = 64ull << 20;
  std::vector<UploadChunk> chunks;
  uint64_t cap = 1;
  for (auto& b : db->blocks) {
    if (!b.local) continue;
    const uint64_t ns = b.h.num_sigs;
    const uint32_t rb = b.h.row_bytes;
    b.stride = device_stride(rb);
    // the kernel addresses rows in 16-byte units with 32 bits
    if ((ns + 1) * (uint64_t)(b.stride >> 4) > 0xffffffffULL)
      return fail(KMCPG_EUNSUPPORTED, "%s: block larger than 64 GB in HBM (NumSigs %llu x %u B)", b.path.c_str(), (unsigned long long)ns, b.stride);
    HIPCHK(hipMalloc((void**)&b.d_rows, (ns + 1) * (uint64_t)b.stride));
    HIPCHK(hipMemset(b.d_rows + ns * b.stride, 0, b.stride));  // the all-zero row
    const uint64_t chunk_rows = std::max<uint64_t>(1, kChunkBytes / rb);
    for (uint64_t r0 = 0; r0 < ns; r0 += chunk_rows) {
      chunks.push_back(UploadChunk{&b, r0, std::min(chunk_rows, ns - r0)});
      cap = std::max(cap, chunks.back().nr * rb);
    }
  }
  if (chunks.empty()) return 0;
  int T = (int)std::min<size_t>(std::max(1u, std::min(8u, std::thread::hardware_concurrency())), chunks.size());
  if (const char* e = getenv("KMCPG_LOAD_THREADS")) T = std::max(1, std::min(atoi(e), 64));
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  std::mutex emu;
  int ecode = 0;
  std::string emsg;
  auto set_err = [&](int code, const std::string& m) {
    std::lock_guard<std::mutex> g(emu);
    if (!failed.exchange(1)) {
      ecode = code;
      emsg = m;
    }
  };
  const int dev = db->opts.device;
  auto worker = [&]() {
    uint8_t *h_buf = nullptr, *d_tmp = nullptr;
    hipStream_t st = nullptr;
    int fd = -1;
    const BlockMeta* fd_of = nullptr;
    hipError_t he = hipSetDevice(dev);
    if (he == hipSuccess) he = hipHostMalloc((void**)&h_buf, cap, hipHostMallocDefault);
    if (he == hipSuccess) he = hipMalloc((void**)&d_tmp, cap);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (he != hipSuccess) set_err(KMCPG_EDEVICE, std::string("upload staging: ") + hipGetErrorString(he));
    while (!failed.load()) {
      const size_t i = next.fetch_add(1);
      if (i >= chunks.size()) break;
      const UploadChunk& c = chunks[i];
      const uint32_t rb = c.b->h.row_bytes;
      if (fd_of != c.b) {
        if (fd >= 0) close(fd);
        fd = open(c.b->path.c_str(), O_RDONLY);
        fd_of = c.b;
        if (fd < 0) {
          set_err(KMCPG_EIO, "kmcp index file missing: " + c.b->path);
          break;
        }
      }
      he = hipStreamSynchronize(st);  // the previous chunk has left the staging buffers
      if (he != hipSuccess) {
        set_err(KMCPG_EDEVICE, "uploading " + c.b->path + ": " + hipGetErrorString(he));
        break;
      }
      const uint64_t want = c.nr * rb;
      uint64_t got = 0;
      while (got < want) {
        ssize_t n = pread(fd, h_buf + got, want - got, (off_t)(c.b->h.offset0 + c.r0 * rb + got));
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) break;
        got += (uint64_t)n;
      }
      if (got != want) {
        set_err(KMCPG_EFORMAT, "kmcp: truncated index file: " + c.b->path);
        break;
      }
      he = hipMemcpyAsync(d_tmp, h_buf, want, hipMemcpyHostToDevice, st);
      if (he != hipSuccess) {
        set_err(KMCPG_EDEVICE, "uploading " + c.b->path + ": " + hipGetErrorString(he));
        break;
      }
      launch_repack(d_tmp, c.b->d_rows + c.r0 * c.b->stride, c.nr, rb, c.b->stride, st);
    }
    if (st) {
      he = hipStreamSynchronize(st);
      if (he != hipSuccess) set_err(KMCPG_EDEVICE, std::string("upload: ") + hipGetErrorString(he));
      (void)hipStreamDestroy(st);
    }
    if (fd >= 0) close(fd);
    if (d_tmp) (void)hipFree(d_tmp);
    if (h_buf) (void)hipHostFree(h_buf);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back(worker);
  worker();
  for (auto& t : th) t.join();
  if (failed.load()) return fail(ecode, "%s", emsg.c_str());
  return 0;
}

// releases device memory too when an open fails half-way
struct DbCloser {
  void operator()(kmcpg_db* d) const { kmcpg_close(d); }
};
typedef std::unique_ptr<kmcpg_db, DbCloser> DbPtr;

int check_opts(const kmcpg_opts* o, kmcpg_opts* out) {
  kmcpg_opts d{};
  d.device = 0;
  d.shard_rank = 0;
  d.shard_count = 1;
  if (o) d = *o;
  if (d.shard_count < 1 || d.shard_rank < 0 || d.shard_rank >= d.shard_count) return fail(KMCPG_EINVAL, "bad shard_rank/shard_count");
  if (d.device == -1) {  // metadata only: headers parsed, nothing resident, every GPU entry point refuses
    *out = d;
    return 0;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(KMCPG_EDEVICE, "no HIP device available: libkmcpgpu has no CPU fallback");
  if (d.device < 0 || d.device >= ndev) return fail(KMCPG_EINVAL, "device %d out of range (%d devices)", d.device, ndev);
  *out = d;
  return 0;
}

}  // namespace

extern "C" int kmcpg_open(const char* db_dir, const kmcpg_opts* opts, kmcpg_db** out) {
  if (!db_dir || !out) return fail(KMCPG_EINVAL, "null argument");
  *out = nullptr;
  DbPtr db(new kmcpg_db());
  int rc = check_opts(opts, &db->opts);
  if (rc) return rc;
  const bool meta_only = db->opts.device < 0;
  if (!meta_only) HIPCHK(hipSetDevice(db->opts.device));
  const std::string dir(db_dir);
  DbYml y;
  std::string e = read_db_yml(dir + "/__db.yml", &y);
  if (!e.empty()) return fail(e.find("open") != std::string::npos ? KMCPG_EIO : KMCPG_EFORMAT, "%s", e.c_str());
  int k = y.ks.empty() ? y.k : *std::max_element(y.ks.begin(), y.ks.end());
  kmcpg_info& I = db->info;
  I.k = k;
  I.canonical = y.canonical;
  I.num_hashes = y.num_hashes;
  I.scaled = y.scaled;
  I.scale = y.scale;
  I.minimizer = y.minimizer;
  I.minimizer_w = y.minimizer_w;
  I.syncmer = y.syncmer;
  I.syncmer_s = y.syncmer_s;
  I.fpr = y.fpr;
  uint32_t base = 0;
  for (const auto& fn : y.files) {
    BlockMeta b;
    b.path = dir + "/" + fn;
    e = read_uniki_header(b.path, &b.h);
    if (!e.empty()) return fail(e.find("missing") != std::string::npos ? KMCPG_EIO : KMCPG_EFORMAT, "%s", e.c_str());
    // compatibility checks of NewUnikIndexDB (:689-695) and Header.Compatible (serialization.go:90-99)
    if (b.h.k != k || b.h.canonical != (bool)y.canonical || b.h.num_hashes != y.num_hashes || (y.uniki_version >= 0 && y.uniki_version != b.h.version))
      return fail(KMCPG_EFORMAT, "index files not compatible");
    b.col_base = base;
    base += (uint32_t)b.h.names.size();
    I.matrix_bytes += b.h.num_sigs * (uint64_t)b.h.row_bytes;
    db->blocks.push_back(std::move(b));
  }
  I.n_blocks = (int32_t)db->blocks.size();
  I.n_cols = base;
  if (I.num_hashes < 1 || I.num_hashes > 4) return fail(KMCPG_EUNSUPPORTED, "hashes=%d (kmcp index allows 1..4)", I.num_hashes);
  assign_shards(db.get());
  if (!meta_only) {
    rc = upload_blocks(db.get());
    if (rc) return rc;
  } else {
    for (auto& b : db->blocks)
      if (b.local) b.stride = device_stride(b.h.row_bytes);
  }
  rc = finish_open(db.get());
  if (rc) return rc;
  *out = db.release();
  return 0;
}

extern "C" int kmcpg_open_synthetic(const kmcpg_synth_spec* s, const kmcpg_opts* opts, kmcpg_db** out) {
  if (!s || !out) return fail(KMCPG_EINVAL, "null argument");
  *out = nullptr;
  if (s->n_blocks == 0 || s->cols_per_block == 0 || s->num_sigs == 0 || s->num_hashes < 1 || s->num_hashes > 4)
    return fail(KMCPG_EINVAL, "bad synthetic spec");
  DbPtr db(new kmcpg_db());
  int rc = check_opts(opts, &db->opts);
  if (rc) return rc;
  KMCPG_USE_DEVICE(db);  // a synthetic index only exists in HBM
  db->synthetic = true;
  kmcpg_info& I = db->info;
  I.k = s->k;
  I.canonical = 1;
  I.num_hashes = s->num_hashes;
  I.scale = s->scale > 1 ? s->scale : 1;
  I.scaled = s->scale > 1;
  I.syncmer = s->syncmer_s > 0;
  I.syncmer_s = s->syncmer_s;
  I.minimizer = s->minimizer_w > 0;
  I.minimizer_w = s->minimizer_w;
  I.fpr = s->fpr;
  uint32_t base = 0;
  char name[64];
  for (uint32_t i = 0; i < s->n_blocks; i++) {
    BlockMeta b;
    snprintf(name, sizeof name, "<synthetic block %u>", i);
    b.path = name;
    b.h.version = 4;
    b.h.k = s->k;
    b.h.canonical = true;
    b.h.compact = true;
    b.h.num_hashes = s->num_hashes;
    b.h.num_sigs = s->num_sigs;
    b.h.row_bytes = (s->cols_per_block + 7) / 8;
    for (uint32_t c = 0; c < s->cols_per_block; c++) {
      snprintf(name, sizeof name, "syn%u", base + c);
      b.h.names.push_back(name);
      b.h.gsizes.push_back(4000000);
      b.h.indices.push_back((c % 10) | (10u << 16));
      b.h.sizes.push_back(s->kmers_per_col);
    }
    b.col_base = base;
    base += s->cols_per_block;
    I.matrix_bytes += b.h.num_sigs * (uint64_t)b.h.row_bytes;
    db->blocks.push_back(std::move(b));
  }
  I.n_blocks = (int32_t)db->blocks.size();
  I.n_cols = base;
  assign_shards(db.get());
  // density of one Bloom filter holding kmers_per_col of num_sigs slots with h hashes, at 8-bit resolution
  const double dens = 1.0 - exp(-(double)s->num_hashes * (double)s->kmers_per_col / (double)s->num_sigs);
  uint32_t p8 = (uint32_t)llround(dens * 256.0);
  if (p8 > 255) p8 = 255;
  for (size_t i = 0; i < db->blocks.size(); i++) {
    BlockMeta& b = db->blocks[i];
    if (!b.local) continue;
    b.stride = device_stride(b.h.row_bytes);
    if ((b.h.num_sigs + 1) * (uint64_t)(b.stride >> 4) > 0xffffffffULL) return fail(KMCPG_EUNSUPPORTED, "synthetic block larger than 64 GB");
    HIPCHK(hipMalloc((void**)&b.d_rows, (b.h.num_sigs + 1) * (uint64_t)b.stride));
    HIPCHK(hipMemset(b.d_rows + b.h.num_sigs * b.stride, 0, b.stride));
    launch_synth_fill(b.d_rows, b.h.num_sigs, b.stride, (uint32_t)b.h.names.size(), s->seed * 0x9e3779b97f4a7c15ULL + i * 0x632be59bd9b4e019ULL + 1, p8,
                      nullptr);
  }
  HIPCHK(hipDeviceSynchronize());
  rc = finish_open(db.get());
  if (rc) return rc;
  *out = db.release();
  return 0;
}

extern "C" int kmcpg_close(kmcpg_db* db) {
  if (!db) return 0;
  for (kmcpg_db* sh : db->shards) kmcpg_close(sh);
  db->shards.clear();
  if (db->opts.device >= 0) (void)hipSetDevice(db->opts.device);
  for (auto& b : db->blocks)
    if (b.d_rows) (void)hipFree(b.d_rows);
  if (db->d_blockdev) (void)hipFree(db->d_blockdev);
  for (auto& c : db->classes)
    if (c.d_slots) (void)hipFree(c.d_slots);
  db->w_hashes.release();
  db->w_scratch.release();
  db->w_nk_raw.release();
  db->w_nk1.release();
  db->w_seg_cnt.release();
  db->w_long_list.release();
  db->w_long_meta.release();
  db->w_long_counts.release();
  db->w_huge_info.release();
  db->w_huge_temp.release();
  db->s_seqs.release();
  db->s_seqs2.release();
  db->s_offs.release();
  db->s_offs2.release();
  db->s_counter.release();
  db->s_hits.release();
  db->s_qk.release();
  db->s_ql.release();
  for (auto& ev : db->ev)
    if (ev) (void)hipEventDestroy(ev);
  delete db;
  return 0;
}

extern "C" int kmcpg_db_info(const kmcpg_db* db, kmcpg_info* info) {
  if (!db || !info) return fail(KMCPG_EINVAL, "null argument");
  *info = db->info;
  return 0;
}

extern "C" int kmcpg_col_info(const kmcpg_db* db, uint32_t col, const char** name, uint32_t* target_idx, uint64_t* gsize, uint64_t* size) {
  if (!db) return fail(KMCPG_EINVAL, "null argument");
  if (col >= db->col_block.size()) return fail(KMCPG_EINVAL, "column %u out of range", col);
  const BlockMeta& b = db->blocks[db->col_block[col]];
  const uint32_t c = col - b.col_base;
  if (name) *name = b.h.names[c].c_str();
  if (target_idx) *target_idx = b.h.indices[c];
  if (gsize) *gsize = b.h.gsizes[c];
  if (size) *size = b.h.sizes[c];
  return 0;
}

extern "C" int kmcpg_block_info(const kmcpg_db* db, uint32_t block, uint64_t* num_sigs, uint32_t* n_cols, uint32_t* row_bytes, uint32_t* dev_stride,
                                int32_t* is_local, uint32_t* col_base) {
  if (!db || block >= db->blocks.size()) return fail(KMCPG_EINVAL, "bad block");
  const BlockMeta& b = db->blocks[block];
  if (num_sigs) *num_sigs = b.h.num_sigs;
  if (n_cols) *n_cols = (uint32_t)b.h.names.size();
  if (row_bytes) *row_bytes = b.h.row_bytes;
  if (dev_stride) *dev_stride = b.stride;
  if (is_local) *is_local = b.local ? 1 : 0;
  if (col_base) *col_base = b.col_base;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GPU half
// ------------------------------------------------------------------------------------------------
namespace {

kmcpg_params default_params() {
  kmcpg_params p{};
  p.min_qlen = 30;
  p.min_matched = 10;
  p.min_qcov = 0.55;
  p.min_tcov = 0;
  p.max_fpr = 0.01;
  p.dedup_threshold = 256;
  return p;
}

uint64_t max_hash_for(uint32_t scale) {
  // uint64(float64(^uint64(0)) / float64(scale))  (util-db-search.go:1040-1043)
  const double d = 18446744073709551616.0 / (double)scale;
  if (d >= 18446744073709551616.0) return ~0ULL;
  return (uint64_t)d;
}

// K1 (+K1d): hashes of read i end up at d_hashes[offs[i] + offs2[i] ...], NumKmers in d_nk_search
int run_kmers(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, const uint8_t* d_seqs2, const uint64_t* d_offs2, uint32_t n_reads,
              uint32_t max_read_len, const kmcpg_params& p, uint64_t* d_hashes, uint64_t* d_scratch, uint64_t scratch_half, int32_t* d_nk_raw, int32_t* d_nk1,
              int32_t* d_nk_search, int32_t* d_qlen, hipStream_t st, uint64_t* max_n_out) {
  const kmcpg_info& I = db->info;
  if (!I.canonical) return fail(KMCPG_EUNSUPPORTED, "non-canonical index");
  K1Args a{};
  a.seqs = d_seqs;
  a.offs = d_offs;
  a.seqs2 = d_seqs2;
  a.offs2 = d_offs2;
  a.n_reads = n_reads;
  a.k = I.k;
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
  // whole genomes (single-end, plain or FracMinHash k-mers): segments of a read on their own workgroups
  const uint32_t segs = (max_read_len + (uint32_t)k1_segment_len() - 1) / (uint32_t)k1_segment_len();
  if (a.mode == 0 && !d_seqs2 && segs > 1 && d_scratch && (uint64_t)n_reads * segs <= (1ull << 21)) {  // one workgroup of 1024 threads per segment, < 2^32 threads per launch
    if (db->w_seg_cnt.ensure((size_t)n_reads * segs)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
    a.seg_cnt = db->w_seg_cnt.p;
    a.segs_max = segs;
  }
  launch_k1(a, max_read_len, st);
  uint64_t ub = max_read_len >= (uint32_t)I.k ? (uint64_t)(max_read_len - I.k + 1) : 0;
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
    launch_dedup(d, ub, st);
    if (ub > HUGE_MIN) {
      // whole-genome queries: which ones they are is only known on the device -> one small read-back, then a device-wide
      // sort + unique per such query
      const int32_t thr = std::max<int32_t>((int32_t)HUGE_MIN, p.dedup_threshold);
      uint32_t meta[2] = {0, 0};
      if (db->w_long_list.ensure(n_reads + 1) || db->w_long_meta.ensure(2)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
      HIPCHK(hipMemsetAsync(db->w_long_meta.p, 0, 2 * sizeof(uint32_t), st));
      launch_list_long(d_nk_raw, n_reads, thr, db->w_long_list.p, db->w_long_meta.p, st);
      HIPCHK(hipMemcpyAsync(meta, db->w_long_meta.p, sizeof meta, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (meta[0]) {
        const size_t tb = huge_dedup_temp_bytes(meta[1]);
        if (db->w_huge_info.ensure(3 * (size_t)meta[0] + 1) || db->w_huge_temp.ensure(tb + 64)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
        launch_gather_huge(db->w_long_list.p, meta[0], d_nk_raw, d_offs, d_offs2, db->w_huge_info.p, st);
        std::vector<uint64_t> hinfo(3 * (size_t)meta[0]);
        HIPCHK(hipMemcpyAsync(hinfo.data(), db->w_huge_info.p, hinfo.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int* d_num = (int*)db->w_huge_temp.p;  // first 64 bytes: the unique count
        for (uint32_t i = 0; i < meta[0]; i++) {
          const uint32_t r = (uint32_t)hinfo[3 * i], n = (uint32_t)hinfo[3 * i + 1];
          const uint64_t koff = hinfo[3 * i + 2];
          if (huge_dedup(d_hashes + koff, d_scratch + koff, n, d_num, db->w_huge_temp.p + 64, tb, d_nk_search, r, p.min_matched, st) != 0)
            return fail(KMCPG_EDEVICE, "device-wide sort of a %u-k-mer query failed", n);
        }
      }
    }
  } else {
    launch_nk_simple(d_nk_raw, d_nk_search, n_reads, p.min_matched, st);
  }
  return 0;
}

}  // namespace

extern "C" int kmcpg_kmers_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, uint32_t n_reads, uint64_t total_bases,
                                  uint32_t max_read_len, const kmcpg_params* params, uint64_t* d_hashes, uint64_t hashes_cap, uint64_t* d_koff,
                                  int32_t* d_nk, void* stream) {
  if (!db || !d_seqs || !d_offs || !d_hashes || !d_nk) return fail(KMCPG_EINVAL, "null argument");
  if (hashes_cap < total_bases) return fail(KMCPG_EINVAL, "hashes_cap must be >= total_bases");
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  const kmcpg_params p = params ? *params : default_params();
  hipStream_t st = (hipStream_t)stream;
  if (db->w_scratch.ensure(2 * total_bases + 2) || db->w_nk_raw.ensure(n_reads + 1) || db->w_nk1.ensure(n_reads + 1)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
  DevBuf<int32_t> ql;
  if (ql.ensure(n_reads + 1)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
  uint64_t maxn = 0;
  int rc = run_kmers(db, d_seqs, d_offs, nullptr, nullptr, n_reads, max_read_len, p, d_hashes, db->w_scratch.p, total_bases + 1, db->w_nk_raw.p, db->w_nk1.p,
                     d_nk, ql.p, st, &maxn);
  if (rc == 0 && d_koff) HIPCHK(hipMemcpyAsync(d_koff, d_offs, (size_t)n_reads * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
  hipError_t e = hipStreamSynchronize(st);
  ql.release();
  if (rc) return rc;
  if (e != hipSuccess) return fail(KMCPG_EDEVICE, "k-mer kernel failed: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int kmcpg_query_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, const uint8_t* d_seqs2, const uint64_t* d_offs2,
                                  uint32_t n_reads, uint64_t total_bases, uint32_t max_read_len, const kmcpg_params* params, kmcpg_hit* d_hits,
                                  uint64_t hit_cap, uint64_t* d_counters, int32_t* d_qkmers, int32_t* d_qlen, void* stream) {
  if (!db || !d_seqs || !d_offs || !d_counters || !d_qkmers || !d_qlen || (!d_hits && hit_cap)) return fail(KMCPG_EINVAL, "null argument");
  if ((d_seqs2 == nullptr) != (d_offs2 == nullptr)) return fail(KMCPG_EINVAL, "seqs2 and offs2 must be given together");
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  const kmcpg_params p = params ? *params : default_params();
  if (p.min_matched < 1) return fail(KMCPG_EINVAL, "min_matched must be >= 1");  // getFlagPositiveInt (search.go:165)
  hipStream_t st = (hipStream_t)stream;
  if (db->w_hashes.ensure(total_bases + 1) || db->w_nk_raw.ensure(n_reads + 1) || db->w_nk1.ensure(n_reads + 1)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
  uint64_t ub = max_read_len >= (uint32_t)db->info.k ? (uint64_t)(max_read_len - db->info.k + 1) : 0;
  if (d_seqs2) ub *= 2;
  const bool window_sketch = db->info.syncmer || db->info.minimizer;
  if ((ub > (uint64_t)p.dedup_threshold || window_sketch) && db->w_scratch.ensure(2 * total_bases + 2)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
  uint64_t maxn = 0;
  if (db->profiling) {
    for (auto& ev : db->ev)
      if (!ev) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipEventRecord(db->ev[0], st));
  }
  int rc = run_kmers(db, d_seqs, d_offs, d_seqs2, d_offs2, n_reads, max_read_len, p, db->w_hashes.p, db->w_scratch.p, total_bases + 1, db->w_nk_raw.p,
                     db->w_nk1.p, d_qkmers, d_qlen, st, &maxn);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(d_counters, 0, sizeof(uint64_t), st));
  if (db->profiling) HIPCHK(hipEventRecord(db->ev[1], st));
  // long queries (whole genomes, -g) are split into chunks of k-mers so that they spread over the chip; short ones keep
  // the one-wave-per-(query, slot) kernel.  Which queries are long is only known on the device: one small D2H read.
  const char* sm_env = getenv("KMCPG_SPLIT_MIN");
  const int32_t split_min = sm_env ? atoi(sm_env) : 2048;
  uint32_t long_meta[2] = {0, 0};
  if (split_min > 0 && maxn > (uint64_t)split_min) {
    if (db->w_long_list.ensure(n_reads + 1) || db->w_long_meta.ensure(2)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
    HIPCHK(hipMemsetAsync(db->w_long_meta.p, 0, 2 * sizeof(uint32_t), st));
    launch_list_long(d_qkmers, n_reads, split_min, db->w_long_list.p, db->w_long_meta.p, st);
    HIPCHK(hipMemcpyAsync(long_meta, db->w_long_meta.p, sizeof long_meta, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  uint32_t n_long = long_meta[0];
  size_t total_slots = 0;
  for (const auto& c : db->classes) total_slots += c.slots.size();
  // splitting pays when the long queries alone would leave the chip idle (few (query, slot) pairs) or need more than 16
  // counter planes; a batch of thousands of 10-kb reads already fills it and keeps the plain kernel (unless forced by env)
  if (n_long && !sm_env && (uint64_t)n_long * total_slots > 16384 && long_meta[1] <= 65535) n_long = 0;
  // largest NumKmers the plain kernel will meet: bounded by the read length, and exactly known once the long ones were listed
  uint64_t max_short = maxn;
  if (split_min > 0 && maxn > (uint64_t)split_min)
    max_short = n_long ? (uint64_t)split_min : std::max<uint64_t>(long_meta[1], (uint64_t)split_min);
  const int npl = max_short <= 255 ? 8 : (max_short <= 65535 ? 16 : (max_short <= 16777215 ? 24 : 0));
  if (!npl) return fail(KMCPG_EUNSUPPORTED, "queries with more than 16777215 k-mers need KMCPG_SPLIT_MIN > 0");
  K2Args a{};
  a.blocks = db->d_blockdev;
  a.n_reads = n_reads;
  a.hashes = db->w_hashes.p;
  a.offs = d_offs;
  a.offs2 = d_offs2;
  a.nk = d_qkmers;
  a.min_qcov = p.min_qcov;
  a.min_matched = p.min_matched;
  a.num_hashes = db->info.num_hashes;
  a.nt_loads = getenv("KMCPG_NT_LOADS") ? atoi(getenv("KMCPG_NT_LOADS")) : 1;
  a.prune = getenv("KMCPG_PRUNE") ? atoi(getenv("KMCPG_PRUNE")) : 1;
  a.split_min = n_long ? split_min : 0;
  a.hits = d_hits;
  a.hit_cap = hit_cap;
  a.counter = (unsigned long long*)d_counters;
  for (const auto& c : db->classes) {
    a.slots = c.d_slots;
    a.nslots = (uint32_t)c.slots.size();
    if (launch_k2(a, c.lpr, npl, st) != 0) return fail(KMCPG_EINVAL, "batch too large for one launch: split it");
  }
  if (n_long) {
    a.ncols_total = (uint32_t)db->info.n_cols;
    // ~64 chunks for the largest query, 1024..8192 k-mers each (at most 8192: the chunk's counts fit 16 planes)
    uint32_t chk = 1024;
    while (chk < 8192 && (uint64_t)chk * 64 < long_meta[1]) chk <<= 1;
    a.split_chk = chk;
    a.split_chunks = (long_meta[1] + chk - 1) / chk;
    // count arrays of at most ~2 GB at a time
    const uint32_t group = (uint32_t)std::max<uint64_t>(1, (2ull << 30) / ((uint64_t)a.ncols_total * 4));
    if (db->w_long_counts.ensure((size_t)std::min<uint32_t>(group, n_long) * a.ncols_total)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
    a.long_counts = db->w_long_counts.p;
    for (uint32_t g0 = 0; g0 < n_long; g0 += group) {
      a.long_list = db->w_long_list.p + g0;
      a.n_long = std::min<uint32_t>(group, n_long - g0);
      HIPCHK(hipMemsetAsync(a.long_counts, 0, (size_t)a.n_long * a.ncols_total * sizeof(uint32_t), st));
      for (const auto& c : db->classes) {
        a.slots = c.d_slots;
        a.nslots = (uint32_t)c.slots.size();
        if (launch_k2_split(a, c.lpr, st) != 0) return fail(KMCPG_EINVAL, "batch too large for one launch: split it");
      }
      launch_threshold_long(a, st);
    }
  }
  if (db->profiling) {
    HIPCHK(hipEventRecord(db->ev[2], st));
    db->ev_valid = true;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int kmcpg_set_profiling(kmcpg_db* db, int enable) {
  if (!db) return fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  db->profiling = enable != 0;
  db->ev_valid = false;
  return 0;
}

extern "C" int kmcpg_last_timing(kmcpg_db* db, float* kmers_ms, float* cobs_ms) {
  if (!db) return fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  if (!db->profiling || !db->ev_valid) return fail(KMCPG_EINVAL, "no profiled kmcpg_query_device call yet");
  KMCPG_USE_DEVICE(db);
  HIPCHK(hipEventSynchronize(db->ev[2]));
  float a = 0, b = 0;
  HIPCHK(hipEventElapsedTime(&a, db->ev[0], db->ev[1]));
  HIPCHK(hipEventElapsedTime(&b, db->ev[1], db->ev[2]));
  if (kmers_ms) *kmers_ms = a;
  if (cobs_ms) *cobs_ms = b;
  return 0;
}

extern "C" int kmcpg_plant_reads_device(kmcpg_db* db, const uint8_t* d_seqs, const uint64_t* d_offs, uint32_t n_reads, uint64_t total_bases,
                                        uint32_t max_read_len, const uint32_t* d_cols, void* stream) {
  if (!db || !d_seqs || !d_offs || !d_cols) return fail(KMCPG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(db->mu);
  KMCPG_USE_DEVICE(db);
  hipStream_t st = (hipStream_t)stream;
  if (db->w_hashes.ensure(total_bases + 1) || db->w_nk_raw.ensure(n_reads + 1) || db->w_nk1.ensure(n_reads + 1)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
  DevBuf<int32_t> tmp;
  if (tmp.ensure(2 * (size_t)n_reads + 2)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
  kmcpg_params p = default_params();
  p.min_qlen = 0;
  p.min_matched = 1;
  p.dedup_threshold = 0x7fffffff;  // plant every k-mer occurrence (idempotent)
  uint64_t maxn = 0;
  if ((db->info.syncmer || db->info.minimizer) && db->w_scratch.ensure(2 * total_bases + 2)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
  int rc = run_kmers(db, d_seqs, d_offs, nullptr, nullptr, n_reads, max_read_len, p, db->w_hashes.p, db->w_scratch.p, total_bases + 1, db->w_nk_raw.p,
                     db->w_nk1.p, tmp.p, tmp.p + n_reads + 1, st, &maxn);
  if (rc == 0)
    launch_plant_reads(db->d_blockdev, (uint32_t)db->h_blockdev.size(), db->info.num_hashes, db->w_hashes.p, d_offs, db->w_nk_raw.p, d_cols, n_reads, st);
  hipError_t e = hipStreamSynchronize(st);
  tmp.release();
  if (rc) return rc;
  if (e != hipSuccess) return fail(KMCPG_EDEVICE, "plant kernel failed: %s", hipGetErrorString(e));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// host half: thresholds that need float64, Match values, sorting (util-db-search.go:7471-7489, :260-345)
// ------------------------------------------------------------------------------------------------
namespace {

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
typedef std::vector<kmcpg_match, NoInitAlloc<kmcpg_match>> MatchVec;

struct ResultOwner {
  std::vector<int32_t> qlen, qkmers;
  std::vector<uint64_t> offs;
  MatchVec matches;
};

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
  const size_t bytes = o->matches.capacity() * sizeof(kmcpg_match) + (o->qlen.capacity() + o->qkmers.capacity()) * 4 + o->offs.capacity() * 8;
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
  if (!db || !out || (!hits && n_hits) || !qkmers || !qlen) return fail(KMCPG_EINVAL, "null argument");
  const kmcpg_params p = params ? *params : default_params();
  std::unique_ptr<ResultOwner, OwnerReturn> o(take_owner());
  o->qlen.assign(qlen, qlen + n_reads);
  o->qkmers.assign(qkmers, qkmers + n_reads);
  // scratch of this thread, kept between calls (a caller thread finalizes batch after batch)
  static thread_local std::vector<uint64_t> start, cur, per_read;
  static thread_local std::vector<kmcpg_hit, NoInitAlloc<kmcpg_hit>> sorted;
  // bucket hits by read (counting sort), then order each bucket by column
  start.assign((size_t)n_reads + 1, 0);
  for (uint64_t i = 0; i < n_hits; i++) {
    if (hits[i].read >= n_reads) return fail(KMCPG_EINVAL, "hit %llu names read %u of %u", (unsigned long long)i, hits[i].read, n_reads);
    if (hits[i].col >= db->col_block.size()) return fail(KMCPG_EINVAL, "hit names column %u of %zu", hits[i].col, db->col_block.size());
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
  out->k = db->info.k;
  out->qlen = o->qlen.data();
  out->qkmers = o->qkmers.data();
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
  if (offs[0] != 0 || (seqs2 && offs2[0] != 0)) return fail(KMCPG_EINVAL, "offs[0] must be 0");
  uint32_t maxlen = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint64_t l = offs[i + 1] - offs[i];
    if (seqs2) l = std::max<uint64_t>(l, offs2[i + 1] - offs2[i]);
    if (l > 0x7fffffffULL) return fail(KMCPG_EUNSUPPORTED, "query longer than 2^31-1 bases");
    maxlen = std::max<uint32_t>(maxlen, (uint32_t)l);
  }
  {
    std::lock_guard<std::mutex> g(db->mu);
    KMCPG_USE_DEVICE(db);
    if (db->s_seqs.ensure(tb1 + 16) || db->s_offs.ensure(n + 1) || db->s_counter.ensure(2) || db->s_qk.ensure(n) || db->s_ql.ensure(n))
      return fail(KMCPG_ENOMEM, "hipMalloc failed");
    if (seqs2 && (db->s_seqs2.ensure(tb2 + 16) || db->s_offs2.ensure(n + 1))) return fail(KMCPG_ENOMEM, "hipMalloc failed");
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
      if (db->s_hits.ensure(cap)) return fail(KMCPG_ENOMEM, "hipMalloc failed");
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
  return fail(KMCPG_ENOMEM, "hit buffer overflow");
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
      if (rcs[i]) errs[i] = g_err;  // thread-local message of the worker
    });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < S; i++)
    if (rcs[i]) return fail(rcs[i], "device %d: %s", db->shards[i]->opts.device, errs[i].c_str());
  rb->qk = parts[0].qk;  // every shard generates the same k-mers
  rb->ql = parts[0].ql;
  rb->hits.clear();
  for (auto& pt : parts) rb->hits.insert(rb->hits.end(), pt.hits.begin(), pt.hits.end());
  return 0;
}

}  // namespace

extern "C" int kmcpg_open_devices(const char* db_dir, const int32_t* devices, int32_t n_devices, kmcpg_db** out) {
  if (!db_dir || !devices || !out || n_devices < 1) return fail(KMCPG_EINVAL, "bad argument");
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
      std::string keep = g_err;
      kmcpg_close(front);
      g_err = keep;
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
  if (!db || !out || (n_reads && (!seqs || !offs))) return fail(KMCPG_EINVAL, "null argument");
  if ((seqs2 == nullptr) != (offs2 == nullptr)) return fail(KMCPG_EINVAL, "seqs2 and offs2 must be given together");
  if (db->opts.shard_count != 1)
    return fail(KMCPG_EINVAL, "kmcpg_search_batch needs the whole database: open it on one GPU or with kmcpg_open_devices; use kmcpg_query_device + kmcpg_finalize per shard");
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
  if (!db || (!hashes && n)) return fail(KMCPG_EINVAL, "null argument");
  if (col >= db->col_block.size()) return fail(KMCPG_EINVAL, "column out of range");
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
  if (!db || block >= db->blocks.size() || (!row_idx && n_rows) || (!out && n_rows)) return fail(KMCPG_EINVAL, "bad argument");
  const BlockMeta& b = db->blocks[block];
  if (!b.local) return fail(KMCPG_EINVAL, "block %u is not resident on this rank", block);
  for (uint64_t i = 0; i < n_rows; i++)
    if (row_idx[i] >= b.h.num_sigs) return fail(KMCPG_EINVAL, "row out of range");
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
