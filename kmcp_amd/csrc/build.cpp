// build.cpp — kmcpg_build_db: `kmcp index` on the GPU from lists of k-mer hashes.
// Host part: block layout (kmcp/cmd/index.go:657-682), signature size (util-hash.go:46-50), .uniki header
// (index/serialization.go:159-300), __db.yml (util-db-info.go:46-79), __name_mapping.tsv (index.go:1375-1393).
// Device part: the Bloom-column scatter (index.go:1107-1309) as one atomicOr per (k-mer, hash).
#include <errno.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <string>
#include <vector>

#include "fpr.hpp"
#include "engine.hpp"
#include "fastmod.hpp"
#include "kernels.hpp"

using namespace kmcpg;



namespace {

void be32(FILE* f, uint32_t v) {
  uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
  fwrite(b, 1, 4, f);
}
void be64(FILE* f, uint64_t v) {
  be32(f, (uint32_t)(v >> 32));
  be32(f, (uint32_t)v);
}

// CalcSignatureSize (util-hash.go:46-50)
uint64_t signature_size(uint64_t n, int h, double fpr) {
  const double ratio = (double)(-h) / log(1.0 - go_pow(fpr, 1.0 / (double)h));
  return (uint64_t)ceil((double)n * ratio);
}

int mkdirs(const std::string& d) {
  std::string cur;
  for (size_t i = 0; i <= d.size(); i++) {
    if (i == d.size() || d[i] == '/') {
      if (!cur.empty() && mkdir(cur.c_str(), 0755) != 0 && errno != EEXIST) return -1;
    }
    if (i < d.size()) cur.push_back(d[i]);
  }
  return 0;
}

#define BHIP(expr)                                                                                        \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      rc = kmcpg_fail(KMCPG_EDEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      goto done;                                                                                          \
    }                                                                                                     \
  } while (0)

// NumSigs of a block as `kmcp index` sizes it: from its fullest column (index.go:936-946, :1023)
uint64_t block_num_sigs(const kmcpg_build_cfg& cfg, const std::vector<const kmcpg_build_col*>& cols) {
  uint64_t max_elems = 0;
  for (auto* c : cols) max_elems = std::max(max_elems, c->n_hashes);
  return signature_size(max_elems, cfg.num_hashes, cfg.fpr);
}

int build_block(const std::string& path, const kmcpg_build_cfg& cfg, const std::vector<const kmcpg_build_col*>& cols, uint64_t num_sigs) {
  int rc = 0;
  const uint32_t n = (uint32_t)cols.size();
  uint64_t max_elems = 0, total = 0;
  for (auto* c : cols) {
    max_elems = std::max(max_elems, c->n_hashes);
    total += c->n_hashes;
  }
  const uint32_t row_bytes = (n + 7) / 8;
  const uint64_t bytes = num_sigs * (uint64_t)row_bytes;
  uint8_t* d_sigs = nullptr;
  uint64_t *d_hashes = nullptr, *d_off = nullptr;
  std::vector<uint8_t> host;
  FILE* f = nullptr;
  const uint64_t chunk_cap = std::max<uint64_t>(max_elems, 64ull << 20);  // hashes per upload
  std::vector<uint64_t> stage, off;
  BHIP(hipMalloc((void**)&d_sigs, bytes + 8));
  BHIP(hipMemset(d_sigs, 0, bytes + 8));
  BHIP(hipMalloc((void**)&d_hashes, chunk_cap * sizeof(uint64_t)));
  BHIP(hipMalloc((void**)&d_off, ((size_t)n + 1) * sizeof(uint64_t)));
  for (uint32_t c0 = 0; c0 < n;) {  // groups of consecutive columns that fit one upload
    stage.clear();
    off.assign(1, 0);
    uint32_t c1 = c0;
    while (c1 < n && (c1 == c0 || stage.size() + cols[c1]->n_hashes <= chunk_cap)) {
      stage.insert(stage.end(), cols[c1]->hashes, cols[c1]->hashes + cols[c1]->n_hashes);
      off.push_back(stage.size());
      c1++;
    }
    if (!stage.empty()) {
      BHIP(hipMemcpy(d_hashes, stage.data(), stage.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
      BHIP(hipMemcpy(d_off, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
      launch_build_scatter(d_sigs, num_sigs, fastmod_magic(num_sigs), row_bytes, cfg.num_hashes, d_hashes, d_off, c0, c1 - c0, stage.size(), nullptr);
      BHIP(hipDeviceSynchronize());
    }
    c0 = c1;
  }
  host.resize(bytes);
  BHIP(hipMemcpy(host.data(), d_sigs, bytes, hipMemcpyDeviceToHost));
  f = fopen(path.c_str(), "wb");
  if (!f) {
    rc = kmcpg_fail(KMCPG_EIO, "cannot write %s: %s", path.c_str(), strerror(errno));
    goto done;
  }
  {
    fwrite(".kmcpidx", 1, 8, f);
    const uint8_t meta[4] = {4, (uint8_t)cfg.k, (uint8_t)((cfg.canonical ? 1 : 0) | 2 /* COMPACT = !faster (index.go:207) */), (uint8_t)cfg.num_hashes};
    fwrite(meta, 1, 4, f);
    be64(f, num_sigs);
    be32(f, n);
    for (auto* c : cols) {
      be32(f, (uint32_t)strlen(c->name) + 1);
      fwrite(c->name, 1, strlen(c->name), f);
      fputc('\n', f);
    }
    be32(f, n);
    for (auto* c : cols) {
      be32(f, 1);
      be64(f, c->gsize);
    }
    be32(f, n);
    for (auto* c : cols) {
      be32(f, 1);
      be32(f, c->chunk_idx + (c->chunks << 16));  // index.go:1096
    }
    for (auto* c : cols) be64(f, c->n_hashes);
    if (fwrite(host.data(), 1, bytes, f) != bytes) rc = kmcpg_fail(KMCPG_EIO, "short write on %s", path.c_str());
  }
done:
  if (f) fclose(f);
  if (d_sigs) (void)hipFree(d_sigs);
  if (d_hashes) (void)hipFree(d_hashes);
  if (d_off) (void)hipFree(d_off);
  (void)total;
  return rc;
}

}  // namespace

extern "C" int kmcpg_build_db(const char* out_dir, const kmcpg_build_cfg* cfg, const kmcpg_build_col* cols, uint32_t n_cols, int32_t device) {
  if (!out_dir || !cfg || !cols || n_cols == 0) return kmcpg_fail(KMCPG_EINVAL, "bad argument");
  if (cfg->num_hashes < 1 || cfg->num_hashes > 4 || !(cfg->fpr > 0 && cfg->fpr < 1) || cfg->k < 1 || cfg->k > 255)
    return kmcpg_fail(KMCPG_EINVAL, "bad build configuration");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return kmcpg_fail(KMCPG_EDEVICE, "no HIP device available: libkmcpgpu has no CPU fallback");
  if (device < 0 || device >= ndev) return kmcpg_fail(KMCPG_EINVAL, "device %d out of range", device);
  if (hipSetDevice(device) != hipSuccess) return kmcpg_fail(KMCPG_EDEVICE, "hipSetDevice failed");
  std::vector<const kmcpg_build_col*> order;
  uint64_t total = 0;
  for (uint32_t i = 0; i < n_cols; i++) {
    if (!cols[i].name || (!cols[i].hashes && cols[i].n_hashes)) return kmcpg_fail(KMCPG_EINVAL, "column %u: null name or hashes", i);
    order.push_back(&cols[i]);
    total += cols[i].n_hashes;
  }
  // files sorted by #k-mers ascending (index.go:667); the reference's parallel quicksort is unstable, input order breaks ties here
  std::stable_sort(order.begin(), order.end(), [](const kmcpg_build_col* a, const kmcpg_build_col* b) { return a->n_hashes < b->n_hashes; });
  int sblock = cfg->block_size > 0 ? cfg->block_size : ((int)((double)n_cols / (double)std::max(1, cfg->threads)) + 7) / 8 * 8;  // index.go:671-682
  if (sblock > (int)n_cols) sblock = (int)n_cols;
  if (sblock < 8) sblock = 8;
  const std::string dir = std::string(out_dir) + "/R001";
  if (mkdirs(dir) != 0) return kmcpg_fail(KMCPG_EIO, "cannot create %s: %s", dir.c_str(), strerror(errno));
  // Block layout (index.go:787-894).  The reference walks the ascending list with a small state machine; its effect is that the
  // columns fall into size tiers — up to -x k-mers, up to -8, up to -1, above — and every tier is cut into blocks of its own size
  // (-b, -X, 8, 1), a tier change closing the open block.  When -X >= -b the -x tier does not exist (index.go:684-689) and the
  // columns between -8 and -1 keep blocks of -b columns, still separated from the smaller ones.
  const uint64_t thr_x = cfg->kmers_x ? cfg->kmers_x : 10ull << 20, thr_8 = cfg->kmers_8 ? cfg->kmers_8 : 20ull << 20,
                 thr_1 = cfg->kmers_1 ? cfg->kmers_1 : 200ull << 20;
  if (!(thr_x < thr_8 && thr_8 < thr_1)) return kmcpg_fail(KMCPG_EINVAL, "block thresholds must satisfy -x < -8 < -1");  // index.go:242-257
  int size_x = cfg->block_size_x ? cfg->block_size_x : 256;
  if (size_x <= 8 || size_x % 8) return kmcpg_fail(KMCPG_EINVAL, "-X/--block-sizeX should be a multiple of 8 greater than 8: %d", size_x);  // :225-230
  const bool skip_x = size_x >= sblock;
  auto tier = [&](uint64_t km) { return km > thr_1 ? 3 : km > thr_8 ? 2 : (!skip_x && km > thr_x) ? 1 : 0; };
  const int tier_size[4] = {sblock, size_x, skip_x ? sblock : 8, 1};
  std::vector<std::string> files;
  struct Planned {
    std::vector<const kmcpg_build_col*> cols;
    int tier;
    uint64_t num_sigs;
  };
  std::vector<Planned> plan;
  for (size_t i = 0; i < order.size();) {
    if (order[i]->n_hashes == 0) {  // empty inputs are skipped (index.go:799-801)
      i++;
      continue;
    }
    const int t = tier(order[i]->n_hashes);
    Planned b;
    b.tier = t;
    while ((int)b.cols.size() < tier_size[t] && i < order.size() && tier(order[i]->n_hashes) == t) b.cols.push_back(order[i++]);
    b.num_sigs = block_num_sigs(*cfg, b.cols);
    plan.push_back(std::move(b));
  }
  // uniform_sigs (not in the reference): blocks with EQUAL NumSigs share their row addresses (h % NumSigs), so libkmcpgpu lays them
  // side by side in HBM and serves them with one wide gather — 46 M reads/s instead of 15 M on a 10 k-chunk `-j 32` index whose
  // blocks are 39 bytes wide.  `kmcp index` gives every block the size its fullest column asks for, and since
  // the columns are sorted by k-mer count before they are cut into blocks, (almost) no two blocks agree.  A larger filter only
  // lowers a block's false-positive rate below the database's `fpr` (the reader takes any per-block NumSigs,
  // index/serialization.go:383-593), so rounding NumSigs UP is always safe; it costs file size.
  //   1: every block of a tier gets the tier's largest NumSigs (one group per tier; size cost = how uneven the columns are);
  //   2: NumSigs is rounded up to a geometric ladder of ratio 5/4 above the tier's smallest (a few groups per tier, < 25 % larger).
  if (cfg->uniform_sigs == 1 || cfg->uniform_sigs == 2) {
    for (int t = 0; t < 4; t++) {
      uint64_t lo = ~0ull, hi = 0;
      for (const auto& b : plan)
        if (b.tier == t) {
          lo = std::min(lo, b.num_sigs);
          hi = std::max(hi, b.num_sigs);
        }
      if (hi == 0) continue;
      for (auto& b : plan) {
        if (b.tier != t) continue;
        if (cfg->uniform_sigs == 1) b.num_sigs = hi;
        else {
          uint64_t step = lo;
          while (step < b.num_sigs) step = step + step / 4 + 1;
          b.num_sigs = std::min(step, std::max(hi, b.num_sigs));
        }
      }
    }
  } else if (cfg->uniform_sigs != 0) {
    return kmcpg_fail(KMCPG_EINVAL, "uniform_sigs must be 0, 1 or 2");
  }
  for (const auto& b : plan) {
    char name[64];
    snprintf(name, sizeof name, "_block%03zu.uniki", files.size() + 1);  // index.go:1283-1285
    int rc = build_block(dir + "/" + name, *cfg, b.cols, b.num_sigs);
    if (rc) return rc;
    files.push_back(name);
  }
  FILE* f = fopen((dir + "/__db.yml").c_str(), "w");
  if (!f) return kmcpg_fail(KMCPG_EIO, "cannot write %s/__db.yml", dir.c_str());
  auto b = [](int v) { return v ? "true" : "false"; };
  fprintf(f, "version: 4\nunikiVersion: 4\nalias: %s\nk: %d\nks:\n- %d\nhashed: true\ncanonical: %s\n", cfg->alias ? cfg->alias : "kmcp-gpu-db", cfg->k, cfg->k,
          b(cfg->canonical));
  fprintf(f, "scaled: %s\nscale: %u\nminimizer: %s\nminimizer-w: %u\nsyncmer: %s\nsyncmer-s: %u\n", b(cfg->scale > 1), cfg->scale > 1 ? cfg->scale : 1,
          b(cfg->minimizer_w > 0), cfg->minimizer_w, b(cfg->syncmer_s > 0), cfg->syncmer_s);
  fprintf(f, "split-seq: %s\nsplit-size: %d\nsplit-num: %d\nsplit-overlap: %d\ncompact-size: true\n", b(cfg->split_seq), cfg->split_size, cfg->split_num,
          cfg->split_overlap);
  fprintf(f, "hashes: %d\nfpr: %.17g\nnumNameGroups: %u\nblocksize: %d\ntotalKmers: %llu\nfiles:\n", cfg->num_hashes, cfg->fpr, n_cols, sblock,
          (unsigned long long)total);
  for (const auto& fn : files) fprintf(f, "- %s\n", fn.c_str());
  fclose(f);
  f = fopen((dir + "/__name_mapping.tsv").c_str(), "w");
  if (f) {
    for (uint32_t i = 0; i < n_cols; i++) fprintf(f, "%s\t%s\n", cols[i].name, cols[i].name);
    fclose(f);
  }
  return 0;
}

// The resident database back to disk in the reference's format (benchmarking support, no counterpart in the reference): a synthetic
// index generated in HBM (kmcpg_open_synthetic + kmcpg_plant_reads_device) becomes <out_dir>/R001/{_blockNNN.uniki, __db.yml,
// __name_mapping.tsv}, which kmcp-search — or `kmcp search` — opens like any database `kmcp index` wrote (index/serialization.go:159-300,
// util-db-info.go:46-79).  bench.py's end-to-end leg uses it to put BASELINE configs[1] on /dev/shm.  Every block must be resident.
extern "C" int kmcpg_save_db(kmcpg_db* db, const char* out_dir) {
  if (!db || !out_dir) return kmcpg_fail(KMCPG_EINVAL, "null argument");
  for (const auto& b : db->blocks)
    if (!b.local) return kmcpg_fail(KMCPG_EINVAL, "kmcpg_save_db needs every block resident on this handle (one GPU, one shard)");
  const std::string dir = std::string(out_dir) + "/R001";
  if (mkdirs(dir) != 0) return kmcpg_fail(KMCPG_EIO, "cannot create %s: %s", dir.c_str(), strerror(errno));
  std::vector<std::string> files;
  std::vector<uint8_t> host;
  uint64_t total_kmers = 0;
  for (size_t bi = 0; bi < db->blocks.size(); bi++) {
    const kmcpg::UnikiHeader& h = db->blocks[bi].h;
    const uint32_t n = (uint32_t)h.names.size();
    char name[64];
    snprintf(name, sizeof name, "_block%03zu.uniki", bi + 1);
    const std::string path = dir + "/" + name;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return kmcpg_fail(KMCPG_EIO, "cannot write %s: %s", path.c_str(), strerror(errno));
    fwrite(".kmcpidx", 1, 8, f);
    const uint8_t meta[4] = {4, (uint8_t)h.k, (uint8_t)((h.canonical ? 1 : 0) | 2), (uint8_t)h.num_hashes};
    fwrite(meta, 1, 4, f);
    be64(f, h.num_sigs);
    be32(f, n);
    for (const auto& nm : h.names) {
      be32(f, (uint32_t)nm.size() + 1);
      fwrite(nm.data(), 1, nm.size(), f);
      fputc('\n', f);
    }
    be32(f, n);
    for (uint32_t c = 0; c < n; c++) {
      be32(f, 1);
      be64(f, h.gsizes[c]);
    }
    be32(f, n);
    for (uint32_t c = 0; c < n; c++) {
      be32(f, 1);
      be32(f, h.indices[c]);
    }
    for (uint32_t c = 0; c < n; c++) {
      be64(f, h.sizes[c]);
      total_kmers += h.sizes[c];
    }
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / h.row_bytes);
    host.resize((size_t)(std::min(chunk, h.num_sigs) * h.row_bytes));
    int rc = 0;
    for (uint64_t r0 = 0; r0 < h.num_sigs && rc == 0; r0 += chunk) {
      const uint64_t nr = std::min(chunk, h.num_sigs - r0);
      rc = kmcpg_read_row_range(db, (uint32_t)bi, r0, nr, host.data());
      if (rc == 0 && fwrite(host.data(), 1, (size_t)(nr * h.row_bytes), f) != (size_t)(nr * h.row_bytes)) rc = kmcpg_fail(KMCPG_EIO, "short write on %s", path.c_str());
    }
    fclose(f);
    if (rc) return rc;
    files.push_back(name);
  }
  const kmcpg_info& I = db->info;
  FILE* f = fopen((dir + "/__db.yml").c_str(), "w");
  if (!f) return kmcpg_fail(KMCPG_EIO, "cannot write %s/__db.yml", dir.c_str());
  auto b = [](int v) { return v ? "true" : "false"; };
  fprintf(f, "version: 4\nunikiVersion: 4\nalias: kmcp-gpu-saved\nk: %d\nks:\n", I.k);
  std::vector<int> ks(db->ks_desc.rbegin(), db->ks_desc.rend());  // ascending, as `kmcp index` lists them
  if (ks.empty()) ks.push_back(I.k);
  for (int k : ks) fprintf(f, "- %d\n", k);
  fprintf(f, "hashed: true\ncanonical: %s\n", b(I.canonical));
  fprintf(f, "scaled: %s\nscale: %u\nminimizer: %s\nminimizer-w: %u\nsyncmer: %s\nsyncmer-s: %u\n", b(I.scaled), I.scaled ? I.scale : 1, b(I.minimizer),
          I.minimizer_w, b(I.syncmer), I.syncmer_s);
  fprintf(f, "split-seq: false\nsplit-size: 0\nsplit-num: 0\nsplit-overlap: 0\ncompact-size: true\n");
  fprintf(f, "hashes: %d\nfpr: %.17g\nnumNameGroups: %llu\nblocksize: %u\ntotalKmers: %llu\nfiles:\n", I.num_hashes, I.fpr, (unsigned long long)I.n_cols,
          db->blocks.empty() ? 0u : (uint32_t)db->blocks[0].h.names.size(), (unsigned long long)total_kmers);
  for (const auto& fn : files) fprintf(f, "- %s\n", fn.c_str());
  fclose(f);
  f = fopen((dir + "/__name_mapping.tsv").c_str(), "w");
  if (f) {
    for (const auto& bl : db->blocks)
      for (const auto& nm : bl.h.names) fprintf(f, "%s\t%s\n", nm.c_str(), nm.c_str());
    fclose(f);
  }
  return 0;
}
