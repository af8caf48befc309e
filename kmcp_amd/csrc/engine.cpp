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
#include <chrono>
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
#include "fastmod.hpp"
#include "exchange.hpp"
#include "fpr.hpp"
#include "kernels.hpp"

using namespace kmcpg;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

extern "C" const char* kmcpg_last_error(void) { return g_err.c_str(); }
std::string& kmcpg_err_ref() { return g_err; }

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

namespace {

uint32_t device_stride(uint32_t row_bytes, uint32_t align = 64) {
  // rows are padded so a row never straddles more memory lines than it must: powers of two up to 64 B,
  // multiples of 64 B above (1872 -> 1920).  The disk format is untouched (serialization.go:288-300).
  if (row_bytes <= 16) return 16;
  if (row_bytes <= 32) return 32;
  if (row_bytes <= 64) return 64;
  if (row_bytes > 128 && align == 128) return (row_bytes + 127) / 128 * 128;
  return (row_bytes + 63) / 64 * 64;
}

// Row pitch of a database's groups: multiples of 128 bytes for a database whose queries are long (sketches: FracMinHash, minimizer,
// syncmer; several hash functions), of 64 bytes otherwise.  Rows of an odd number of 64-byte halves (782 -> 832 bytes) on a 64-byte pitch
// start in the middle of a 128-byte memory line every other row: the same 7 lines per whole row either way, but a single live sector —
// what long queries gather for the last 40 % of their k-mers, in tail mode (k2_cobs.hip) — is then two lines for half of the rows.
// Genome search, same box: K2 4.69 -> 4.54 ms per 256 genomes.  Short reads on plain k-mer databases read whole rows almost to the
// end and lose 1-6 % to the wider pitch (391-byte rows at 512 instead of 448: profiles/r06_tail_mode.txt), so they keep 64.
// KMCPG_ROW_ALIGN=64 / 128 overrides (read at every open).
uint32_t row_align_for(const kmcpg_info& info) {
  if (const char* e = getenv("KMCPG_ROW_ALIGN")) {
    const int v = atoi(e);
    if (v == 64 || v == 128) return (uint32_t)v;
  }
  return (info.scaled || info.minimizer || info.syncmer || info.num_hashes > 1) ? 128u : 64u;
}

// lanes per row tile (16 B each): the narrowest form that covers the row, so that no lane of a wave idles (a 128-byte row on the
// 16-lane form left half of every wave without a row to load)
int lpr_for_stride(uint32_t stride) {
  if (const char* e = getenv("KMCPG_LPR8"))
    if (atoi(e) == 0) return stride <= 64 ? 4 : (stride <= 256 ? 16 : 64);
  // 257..512 bytes: the 32-lane form, two units per wave (KMCPG_LPR32=0: the 64-lane form with half of its lanes idle, as before round 5)
  const bool lpr32 = !(getenv("KMCPG_LPR32") && atoi(getenv("KMCPG_LPR32")) == 0);
  return stride <= 64 ? 4 : (stride <= 128 ? 8 : (stride <= 256 ? 16 : (stride <= 512 && lpr32 ? 32 : 64)));
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

// Resident blocks with the same NumSigs go side by side into one group (a k-mer has the same row index in all of them), as
// long as the group's rows stay addressable in 16-byte units with 32 bits.  KMCPG_FUSE=0 keeps every block on its own.
void form_groups(kmcpg_db* db) {
  const bool fuse = !(getenv("KMCPG_FUSE") && atoi(getenv("KMCPG_FUSE")) == 0);
  const uint32_t align = row_align_for(db->info);
  db->groups.clear();
  for (size_t i = 0; i < db->blocks.size(); i++) {
    BlockMeta& b = db->blocks[i];
    b.group = -1;
    if (!b.local) continue;
    int gi = -1;
    if (fuse)
      for (size_t g = 0; g < db->groups.size(); g++) {
        const Group& G = db->groups[g];
        if (G.num_sigs != b.h.num_sigs) continue;
        const uint64_t st = device_stride(G.row_bytes + b.h.row_bytes, align);
        if ((G.num_sigs + 1) * (st >> 4) <= 0xffffffffULL) gi = (int)g;
      }
    if (gi < 0) {
      db->groups.push_back(Group{});
      gi = (int)db->groups.size() - 1;
      db->groups[(size_t)gi].num_sigs = b.h.num_sigs;
    }
    Group& G = db->groups[(size_t)gi];
    b.group = gi;
    b.byte_off = G.row_bytes;
    G.row_bytes += b.h.row_bytes;
    G.members.push_back((int)i);
  }
  for (auto& G : db->groups) {
    G.stride = device_stride(G.row_bytes, align);
    for (int m : G.members) db->blocks[(size_t)m].stride = G.stride;
  }
}

// HBM this process may still take on the current device.  KMCPG_HBM_LIMIT_MB caps it (tests of the larger-than-HBM path; a
// host that shares the GPU).
// ~0 = the device would not say: callers skip their fit checks and let hipMalloc decide.
uint64_t hbm_free_bytes() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return ~0ull;
  }
  uint64_t f = free_b;
  if (const char* e = getenv("KMCPG_HBM_LIMIT_MB")) f = std::min<uint64_t>(f, (uint64_t)std::max(0ll, atoll(e)) << 20);
  return f;
}

// what this rank's blocks take in HBM with the layout form_groups chose, plus what the query path needs beside them
// (k-mer workspace, lanes, hit buffers of the first batches)
uint64_t resident_bytes(const kmcpg_db* db) {
  uint64_t need = 0;
  for (const auto& G : db->groups) need += (G.num_sigs + 1) * (uint64_t)G.stride;
  return need;
}
uint64_t workspace_reserve() {
  if (const char* e = getenv("KMCPG_WORKSPACE_RESERVE_MB")) return (uint64_t)std::max(0ll, atoll(e)) << 20;
  return 1ull << 30;
}

// rows of every group: zero-filled (row padding, the all-zero row at index NumSigs, the target of the OR-ing repack)
int alloc_groups(kmcpg_db* db) {
  // The reference searches a database of any size through mmap / --low-mem (util-db-search.go:1238-1280, :6975-7335); here the
  // local blocks must be resident.  Say so with numbers before the first allocation fails half-way: the caller can then split
  // the index over more GPUs (shard_count) or search it in passes (kmcpg_open_paged).
  const uint64_t need = resident_bytes(db), free_b = hbm_free_bytes(), kWorkspaceReserve = workspace_reserve();
  if (free_b != ~0ull && need + kWorkspaceReserve > free_b)
    return kmcpg_fail(KMCPG_ENOMEM, "index does not fit in HBM: shard %d/%d needs %.2f GB for its %zu block group(s) + %.1f GB of workspace, %.2f GB free on device %d "
                      "(more GPUs: kmcpg_open_devices / --gpus; one GPU: kmcpg_open_paged / --gpu-passes)",
                      db->opts.shard_rank, db->opts.shard_count, need / 1e9, db->groups.size(), kWorkspaceReserve / 1e9, free_b / 1e9, db->opts.device);
  for (auto& G : db->groups) {
    // the kernel addresses rows in 16-byte units with 32 bits
    if ((G.num_sigs + 1) * (uint64_t)(G.stride >> 4) > 0xffffffffULL)
      return kmcpg_fail(KMCPG_EUNSUPPORTED, "%s: block larger than 64 GB in HBM (NumSigs %llu x %u B)", db->blocks[(size_t)G.members[0]].path.c_str(),
                        (unsigned long long)G.num_sigs, G.stride);
    const uint64_t bytes = (G.num_sigs + 1) * (uint64_t)G.stride;
    if (hipError_t e = hipMalloc((void**)&G.d_rows, bytes); e != hipSuccess) {
      (void)hipGetLastError();
      return kmcpg_fail(e == hipErrorOutOfMemory ? KMCPG_ENOMEM : KMCPG_EDEVICE, "hipMalloc of %.2f GB for the rows of %s failed: %s", bytes / 1e9,
                        db->blocks[(size_t)G.members[0]].path.c_str(), hipGetErrorString(e));
    }
    HIPCHK(hipMemsetAsync(G.d_rows, 0, bytes, nullptr));
    for (int m : G.members) db->blocks[(size_t)m].d_rows = G.d_rows + db->blocks[(size_t)m].byte_off;
  }
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

int finish_open(kmcpg_db* db) {
  // per-block and per-group device tables + slot classes
  db->local.clear();
  db->h_blockdev.clear();
  db->h_groupdev.clear();
  db->h_segs.clear();
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
    bd.magic_hi = fastmod_magic(b.h.num_sigs);
    bd.stride = b.stride;
    bd.row_bytes = b.h.row_bytes;
    bd.ncols = (uint32_t)b.h.names.size();
    bd.col_base = b.col_base;
    db->h_blockdev.push_back(bd);
    db->info.n_blocks_local++;
    db->info.matrix_bytes_local += b.h.num_sigs * (uint64_t)b.h.row_bytes;
    db->info.row_bytes_sum_local += b.h.row_bytes;
  }
  auto add_slot = [&](int lpr, uint32_t group, uint32_t byte0) {
    SlotClass* cls = nullptr;
    for (auto& c : db->classes)
      if (c.lpr == lpr) cls = &c;
    if (!cls) {
      db->classes.push_back(SlotClass{});
      cls = &db->classes.back();
      cls->lpr = lpr;
    }
    cls->slots.push_back(Slot{group, byte0 / ((uint32_t)lpr * 16u)});
  };
  for (size_t g = 0; g < db->groups.size(); g++) {
    const Group& G = db->groups[g];
    BlockDev gd{};
    gd.rows = G.d_rows;
    gd.num_sigs = G.num_sigs;
    gd.magic_hi = fastmod_magic(G.num_sigs);
    gd.stride = G.stride;
    gd.row_bytes = G.row_bytes;
    gd.col_base = db->blocks[(size_t)G.members[0]].col_base;
    gd.seg0 = (uint32_t)db->h_segs.size();
    gd.nsegs = (uint32_t)G.members.size();
    for (int m : G.members) {
      const BlockMeta& b = db->blocks[(size_t)m];
      db->h_segs.push_back(Seg{b.byte_off, b.byte_off + b.h.row_bytes, b.col_base, (uint32_t)b.h.names.size()});
      gd.ncols += (uint32_t)b.h.names.size();
    }
    db->h_groupdev.push_back(gd);
    // whole 1-KB tiles go to full waves (64 lanes x 16 B); what is left of the row to the narrowest lane group that covers it
    const uint32_t full = G.stride / 1024u, rem = G.stride % 1024u;
    for (uint32_t t = 0; t < full; t++) add_slot(64, (uint32_t)g, t * 1024u);
    // EXPERIMENT, off (KMCPG_SPLIT_TILES=1: multi-hash databases, 2: all): a remainder of 257..896 bytes on the 64-lane form leaves
    // 8..47 lanes of every wave without a row to load; cut into power-of-two tiles that fill their waves — 832 = 512 (32 lanes, two
    // units per wave) + 256 (16 lanes) + 64 (4 lanes), each aligned to its own tile size — the genome search's K2 took 9.35 ms instead
    // of 4.9 ms (same bytes moved): three launches read the hashes and compute the row indices three times, and the narrow parts run
    // on the fabric's request rate.  The idle lanes were never the cost (profiles/r05_split_tiles.txt).
    // Round 6 (profiles/r06_lpr_640.txt, single-hash index, short reads, 23-27 GB): a remainder of 640 bytes as 512 (32-lane form) + 128
    // (8-lane form) is 11-17 % faster than one 64-lane tile with 40 lanes busy (385 -> 329-347 ms per 1 M reads) and is now what a
    // single-hash database gets; 576 = 512 + 64 gains 4.6 % and 768 = 512 + 256 gains 6.8 %: below the 10 % bar, left as one tile.
    const int split_tiles = getenv("KMCPG_SPLIT_TILES") ? atoi(getenv("KMCPG_SPLIT_TILES")) : -1;  // (read at every open: tests flip it; -1 = the rule above)
    const bool split = rem > 256 && rem <= 896 && __builtin_popcount(rem / 64u) <= 3 &&
                       (split_tiles == 2 || (split_tiles == 1 && db->info.num_hashes > 1) || (split_tiles < 0 && rem == 640 && db->info.num_hashes == 1));
    if (rem && split) {
      uint32_t at = full * 1024u, left = rem;
      for (uint32_t part = 512; part >= 64 && left; part >>= 1)
        if (left >= part) {
          add_slot((int)(part / 16u), (uint32_t)g, at);
          at += part;
          left -= part;
        }
    } else if (rem) {
      add_slot(lpr_for_stride(rem), (uint32_t)g, full * 1024u);
    }
  }
  if (db->opts.device >= 0 && !db->h_blockdev.empty()) {
    HIPCHK(hipMalloc((void**)&db->d_blockdev, db->h_blockdev.size() * sizeof(BlockDev)));
    HIPCHK(hipMemcpy(db->d_blockdev, db->h_blockdev.data(), db->h_blockdev.size() * sizeof(BlockDev), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&db->d_groupdev, db->h_groupdev.size() * sizeof(BlockDev)));
    HIPCHK(hipMemcpy(db->d_groupdev, db->h_groupdev.data(), db->h_groupdev.size() * sizeof(BlockDev), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&db->d_segs, db->h_segs.size() * sizeof(Seg)));
    HIPCHK(hipMemcpy(db->d_segs, db->h_segs.data(), db->h_segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
  }
  for (auto& c : db->classes) {
    if (db->opts.device < 0) break;
    HIPCHK(hipMalloc((void**)&c.d_slots, c.slots.size() * sizeof(Slot)));
    HIPCHK(hipMemcpy(c.d_slots, c.slots.data(), c.slots.size() * sizeof(Slot), hipMemcpyHostToDevice));
  }
  db->col_block.clear();
  db->col_meta.clear();
  for (size_t i = 0; i < db->blocks.size(); i++)
    for (size_t c = 0; c < db->blocks[i].h.names.size(); c++) {
      db->col_block.push_back((uint32_t)i);
      db->col_meta.push_back(kmcpg_db::ColMeta{db->blocks[i].h.sizes[c], db->blocks[i].h.gsizes[c], db->blocks[i].h.indices[c], 0});
    }
  db->fpr.reset(new QueryFpr(db->info.fpr));
  if (db->opts.device >= 0 && !db->col_meta.empty()) {  // K3 reads the columns' k-mer counts (-T, the tcov / jacc sort keys)
    std::vector<uint64_t> sz(db->col_meta.size());
    for (size_t c = 0; c < sz.size(); c++) sz[c] = db->col_meta[c].size;
    HIPCHK(hipMalloc((void**)&db->d_col_size, sz.size() * sizeof(uint64_t)));
    HIPCHK(hipMemcpy(db->d_col_size, sz.data(), sz.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
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
  // 64-MB chunks for indexes of many GB; a small index (configs[1]: 1.4 GB) is cut finer — every loader thread pins a staging buffer of
  // one chunk before it can start (0.17 ms per MB, serialised inside the runtime: 8 x 64 MB were 100 ms of a 200-ms upload,
  // profiles/r06_cli_e2e.txt)
  uint64_t local_bytes = 0;
  for (auto& b : db->blocks)
    if (b.local) local_bytes += b.h.num_sigs * (uint64_t)b.h.row_bytes;
  const uint64_t kChunkBytes = local_bytes < (8ull << 30) ? (16ull << 20) : (64ull << 20);
  std::vector<UploadChunk> chunks;
  uint64_t cap = 1;
  for (auto& b : db->blocks) {
    if (!b.local) continue;
    const uint64_t ns = b.h.num_sigs;
    const uint32_t rb = b.h.row_bytes;
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
      launch_repack(d_tmp, db->groups[(size_t)c.b->group].d_rows + c.r0 * c.b->stride, c.nr, rb, c.b->stride, c.b->byte_off, (uint32_t)c.b->h.names.size(), st);
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
  if (failed.load()) return kmcpg_fail(ecode, "%s", emsg.c_str());
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
  if (d.shard_count < 1 || d.shard_rank < 0 || d.shard_rank >= d.shard_count) return kmcpg_fail(KMCPG_EINVAL, "bad shard_rank/shard_count");
  if (d.device == -1) {  // metadata only: headers parsed, nothing resident, every GPU entry point refuses
    *out = d;
    return 0;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return kmcpg_fail(KMCPG_EDEVICE, "no HIP device available: libkmcpgpu has no CPU fallback");
  if (d.device < 0 || d.device >= ndev) return kmcpg_fail(KMCPG_EINVAL, "device %d out of range (%d devices)", d.device, ndev);
  *out = d;
  return 0;
}

}  // namespace

namespace kmcpg {
// Passes a paged handle needs (kmcpg_open_paged): the smallest number S of shards (same byte-balanced partition as
// kmcpg_open with shard_count = S) whose largest shard fits the free HBM of `device` next to the workspace.  `front` is a
// metadata-only handle of the whole database; its own partition is restored before returning.  0 = not even one block fits.
int plan_passes(kmcpg_db* front, int device, uint64_t* largest_shard_bytes, uint64_t* free_bytes, uint64_t* reserve_bytes) {
  if (hipSetDevice(device) != hipSuccess) return 0;
  // a paged handle lives on large batches (every batch pays passes - 1 uploads): room for the k-mer workspace of a few million
  // reads (8-25 B per base) is kept free beside the resident shard
  uint64_t free_b = hbm_free_bytes();
  if (free_b == ~0ull) free_b = 0;  // paging needs a number: an unknown device plans nothing
  const uint64_t reserve = getenv("KMCPG_WORKSPACE_RESERVE_MB") ? workspace_reserve() : std::max<uint64_t>(workspace_reserve(), std::min<uint64_t>(free_b / 8, 24ull << 30));
  if (free_bytes) *free_bytes = free_b;
  if (reserve_bytes) *reserve_bytes = reserve;
  const kmcpg_opts keep = front->opts;
  int found = 0;
  const int nb = (int)front->blocks.size();
  for (int S = 1; S <= std::max(1, nb) && !found; S++) {
    uint64_t worst = 0;
    for (int r = 0; r < S; r++) {
      front->opts.shard_count = S;
      front->opts.shard_rank = r;
      assign_shards(front);
      form_groups(front);
      worst = std::max(worst, resident_bytes(front));
    }
    if (largest_shard_bytes) *largest_shard_bytes = worst;
    if (worst + reserve <= free_b) found = S;
  }
  front->opts = keep;
  assign_shards(front);
  form_groups(front);
  return found;
}
}  // namespace kmcpg

namespace kmcpg {
// kmcpg_open for another shard of a database whose headers `src` has already parsed (the shards of an in-process multi-GPU
// handle, the passes of a paged one): __db.yml and the .uniki headers — hundreds of thousands of reference names at GTDB
// scale — are read once, every further handle copies the parsed metadata and only makes its own blocks resident.
int open_like(const kmcpg_db* src, const kmcpg_opts* opts, kmcpg_db** out) {
  *out = nullptr;
  DbPtr db(new kmcpg_db());
  int rc = check_opts(opts, &db->opts);
  if (rc) return rc;
  const bool meta_only = db->opts.device < 0;
  if (!meta_only) HIPCHK(hipSetDevice(db->opts.device));
  db->db_dir = src->db_dir;
  db->ks_desc = src->ks_desc;
  db->info = src->info;
  db->blocks = src->blocks;
  for (auto& b : db->blocks) {
    b.local = false;
    b.local_idx = -1;
    b.stride = 0;
    b.d_rows = nullptr;
    b.group = -1;
    b.byte_off = 0;
  }
  assign_shards(db.get());
  form_groups(db.get());
  if (!meta_only) {
    rc = alloc_groups(db.get());
    if (rc) return rc;
    rc = upload_blocks(db.get());
    if (rc) return rc;
  }
  rc = finish_open(db.get());
  if (rc) return rc;
  *out = db.release();
  return 0;
}
}  // namespace kmcpg

extern "C" int kmcpg_open(const char* db_dir, const kmcpg_opts* opts, kmcpg_db** out) {
  if (!db_dir || !out) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  *out = nullptr;
  DbPtr db(new kmcpg_db());
  int rc = check_opts(opts, &db->opts);
  if (rc) return rc;
  const bool meta_only = db->opts.device < 0;
  // KMCPG_OPEN_TIMING=1: where the time of an open goes (stderr; profiles/r06_cli_e2e.txt)
  const bool timing = getenv("KMCPG_OPEN_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const double t = now();
    fprintf(stderr, "kmcpg_open: %-28s %8.2f ms\n", what, t - t_prev);
    t_prev = t;
  };
  if (!meta_only) HIPCHK(hipSetDevice(db->opts.device));
  lap("hipSetDevice (runtime init)");
  const std::string dir(db_dir);
  db->db_dir = dir;
  DbYml y;
  std::string e = read_db_yml(dir + "/__db.yml", &y);
  if (!e.empty()) return kmcpg_fail(e.find("open") != std::string::npos ? KMCPG_EIO : KMCPG_EFORMAT, "%s", e.c_str());
  // a database may hold several k-mer sizes (`ks`); queries start with the largest and fall back to smaller ones
  // (util-db-search.go:752-758, :1016-1022); the .uniki headers carry the largest (:690)
  db->ks_desc = y.ks.empty() ? std::vector<int>{y.k} : y.ks;
  std::sort(db->ks_desc.begin(), db->ks_desc.end(), [](int a, int b) { return a > b; });
  db->ks_desc.erase(std::unique(db->ks_desc.begin(), db->ks_desc.end()), db->ks_desc.end());
  int k = db->ks_desc[0];
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
    if (!e.empty()) return kmcpg_fail(e.find("missing") != std::string::npos ? KMCPG_EIO : KMCPG_EFORMAT, "%s", e.c_str());
    // compatibility checks of NewUnikIndexDB (:689-695) and Header.Compatible (serialization.go:90-99)
    if (b.h.k != k || b.h.canonical != (bool)y.canonical || b.h.num_hashes != y.num_hashes || (y.uniki_version >= 0 && y.uniki_version != b.h.version))
      return kmcpg_fail(KMCPG_EFORMAT, "index files not compatible");
    b.col_base = base;
    base += (uint32_t)b.h.names.size();
    I.matrix_bytes += b.h.num_sigs * (uint64_t)b.h.row_bytes;
    db->blocks.push_back(std::move(b));
  }
  I.n_blocks = (int32_t)db->blocks.size();
  I.n_cols = base;
  if (I.num_hashes < 1 || I.num_hashes > 4) return kmcpg_fail(KMCPG_EUNSUPPORTED, "hashes=%d (kmcp index allows 1..4)", I.num_hashes);
  lap("__db.yml + .uniki headers");
  assign_shards(db.get());
  form_groups(db.get());
  if (!meta_only) {
    rc = alloc_groups(db.get());
    if (rc) return rc;
    lap("alloc_groups (hipMalloc)");
    rc = upload_blocks(db.get());
    if (rc) return rc;
    lap("upload_blocks");
  }
  rc = finish_open(db.get());
  if (rc) return rc;
  lap("finish_open (tables)");
  *out = db.release();
  return 0;
}

extern "C" int kmcpg_open_synthetic(const kmcpg_synth_spec* s, const kmcpg_opts* opts, kmcpg_db** out) {
  if (!s || !out) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  *out = nullptr;
  if (s->n_blocks == 0 || s->cols_per_block == 0 || s->num_sigs == 0 || s->num_hashes < 1 || s->num_hashes > 4)
    return kmcpg_fail(KMCPG_EINVAL, "bad synthetic spec");
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
  db->ks_desc.assign(1, s->k);
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
    b.h.num_sigs = s->num_sigs + (uint64_t)i * s->sigs_step;
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
  form_groups(db.get());
  rc = alloc_groups(db.get());
  if (rc) return rc;
  for (size_t i = 0; i < db->blocks.size(); i++) {
    BlockMeta& b = db->blocks[i];
    if (!b.local) continue;
    const double dens = 1.0 - exp(-(double)s->num_hashes * (double)s->kmers_per_col / (double)b.h.num_sigs);
    uint32_t p8 = (uint32_t)llround(dens * 256.0);
    if (p8 > 255) p8 = 255;
    launch_synth_fill(b.d_rows, b.h.num_sigs, b.stride, device_stride(b.h.row_bytes), (uint32_t)b.h.names.size(),
                      s->seed * 0x9e3779b97f4a7c15ULL + i * 0x632be59bd9b4e019ULL + 1, p8, nullptr);
  }
  HIPCHK(hipDeviceSynchronize());
  rc = finish_open(db.get());
  if (rc) return rc;
  *out = db.release();
  return 0;
}

extern "C" int kmcpg_close(kmcpg_db* db) {
  if (!db) return 0;
  if (int n = kmcpg::async_in_flight(db)) return kmcpg_fail(KMCPG_EBUSY, "%d batch(es) still between kmcpg_submit and kmcpg_wait: wait for every ticket before closing", n);
  if (db->exchange) kmcpg::exchange_destroy(db->exchange);
  db->exchange = nullptr;
  for (kmcpg_db* sh : db->shards) kmcpg_close(sh);
  db->shards.clear();
  if (db->paged_resident) kmcpg_close(db->paged_resident);
  db->paged_resident = nullptr;
  if (db->opts.device >= 0) (void)hipSetDevice(db->opts.device);
  for (auto& G : db->groups)
    if (G.d_rows) (void)hipFree(G.d_rows);
  if (db->d_blockdev) (void)hipFree(db->d_blockdev);
  if (db->d_groupdev) (void)hipFree(db->d_groupdev);
  if (db->d_segs) (void)hipFree(db->d_segs);
  for (auto& c : db->classes)
    if (c.d_slots) (void)hipFree(c.d_slots);
  for (auto& w : db->ws) w.release();
  db->w_fin_cnt.release();
  db->w_fin_sums.release();
  if (db->d_col_size) (void)hipFree(db->d_col_size);
  kmcpg::release_fpr_bounds(db);
  kmcpg::async_release(db);
  if (db->k1_stream) (void)hipStreamDestroy(db->k1_stream);
  if (db->cobs_ev) (void)hipEventDestroy(db->cobs_ev);
  if (db->fin_ev) (void)hipEventDestroy(db->fin_ev);
  for (auto& ev : db->ev)
    if (ev) (void)hipEventDestroy(ev);
  delete db;
  return 0;
}

extern "C" int kmcpg_db_info(const kmcpg_db* db, kmcpg_info* info) {
  if (!db || !info) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  *info = db->info;
  return 0;
}

extern "C" int kmcpg_db_ks(const kmcpg_db* db, int32_t* ks, int32_t cap, int32_t* n) {
  if (!db || !n || (cap > 0 && !ks)) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  *n = (int32_t)db->ks_desc.size();
  for (int32_t i = 0; i < cap && i < *n; i++) ks[i] = db->ks_desc[(size_t)i];
  return 0;
}

extern "C" int kmcpg_col_info(const kmcpg_db* db, uint32_t col, const char** name, uint32_t* target_idx, uint64_t* gsize, uint64_t* size) {
  if (!db) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  if (col >= db->col_block.size()) return kmcpg_fail(KMCPG_EINVAL, "column %u out of range", col);
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
  if (!db || block >= db->blocks.size()) return kmcpg_fail(KMCPG_EINVAL, "bad block");
  const BlockMeta& b = db->blocks[block];
  if (num_sigs) *num_sigs = b.h.num_sigs;
  if (n_cols) *n_cols = (uint32_t)b.h.names.size();
  if (row_bytes) *row_bytes = b.h.row_bytes;
  if (dev_stride) *dev_stride = b.stride;
  if (is_local) *is_local = b.local ? 1 : 0;
  if (col_base) *col_base = b.col_base;
  return 0;
}

