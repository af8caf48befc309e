// kmcp-search — `kmcp search` on MI355X: same flags, same 15-column TSV, same trailer (kmcp/cmd/search.go),
// with the per-query work done by libkmcpgpu.so (include/kmcp_gpu.h) instead of the Go search engine.
//
// Mirrors: flags search.go:1031-1107 + root.go:62-82; input handling :793-1000 (single-end, -1/-2 paired-end,
// -g whole file incl. the k-1 N's appended after records 2..m, :899-914); output :436-438 (header), :448-588
// (rows), :1022-1025 (trailer); fatal errors as checkError (util-cli.go:35-40: message + exit status 255).
// Three threads: reader (FASTA/Q, gz via zlib) -> GPU batches -> ordered writer.
#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include <dirent.h>
#include <fcntl.h>
#include <sched.h>
#include <unistd.h>
#include <sys/stat.h>

#include "../include/kmcp_gpu.h"

static const char* VERSION = "0.9.5-mi355x";
static bool g_quiet = false;
static FILE* g_log = nullptr;

static void logf(const char* level, const char* fmt, va_list ap) {
  char buf[2048];
  vsnprintf(buf, sizeof buf, fmt, ap);
  time_t t = time(nullptr);
  struct tm tmv;
  localtime_r(&t, &tmv);
  char ts[32];
  strftime(ts, sizeof ts, "%H:%M:%S.000", &tmv);
  fprintf(stderr, "%s [%s] %s\n", ts, level, buf);
  if (g_log) fprintf(g_log, "%s [%s] %s\n", ts, level, buf);
}
static void info(const char* fmt, ...) {
  if (g_quiet && !g_log) return;
  va_list ap;
  va_start(ap, fmt);
  logf("INFO", fmt, ap);
  va_end(ap);
}
static void warn(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  logf("WARN", fmt, ap);
  va_end(ap);
}
[[noreturn]] void die(const char* fmt, ...) {  // checkError: log + os.Exit(-1)
  va_list ap;
  va_start(ap, fmt);
  logf("ERRO", fmt, ap);
  va_end(ap);
  exit(255);
}

// ------------------------------------------------------------------------------------------------
// options
// ------------------------------------------------------------------------------------------------
struct Options {
  std::string db_dir, out_file = "-", read1, read2, query_id, sort_by = "qcov", infile_list, log_file;
  std::vector<std::string> name_maps, files;
  int min_qlen = 30, min_kmers = 10, dedup = 256, top_scores = 0, threads = 0, device = 0, batch = 131072, gpus = 1, gpu_passes = -1;
  bool batch_given = false;
  std::vector<int32_t> gpu_ids;
  double min_qcov = 0.55, min_tcov = 0, max_fpr = 0.01;
  bool load_whole = false, low_mem = false, whole_file = false, use_filename = false, keep_unmatched = false, no_header = false,
       do_not_sort = false, default_name_map = false, try_se = false, quiet = false, parse_only = false;
};

static void usage() {
  fputs(
      "kmcp-search: search sequences against a kmcp database on an AMD MI355X GPU\n\n"
      "Usage:\n  kmcp-search [-w] -d <kmcp db> [-t <min-query-cov>] [read1.fq.gz] [read2.fq.gz] [unpaired.fq.gz] [-o read.tsv.gz]\n\n"
      "Flags (identical to `kmcp search`, kmcp/cmd/search.go:1031-1107):\n"
      "  -d, --db-dir string            database directory created by \"kmcp index\"\n"
      "  -1, --read1 / -2, --read2      paired-end files;   --try-se   retry unmatched pairs with read1, then read2\n"
      "  -o, --out-file string          out file, \".gz\" supported (default \"-\")\n"
      "  -t, --min-query-cov float      (default 0.55)      -T, --min-target-cov float (default 0)\n"
      "  -c, --min-kmers int            (default 10)        -m, --min-query-len int    (default 30)\n"
      "  -f, --max-fpr float            (default 0.01)      -u, --kmer-dedup-threshold int (default 256)\n"
      "  -s, --sort-by qcov|tcov|jacc   -S, --do-not-sort   -n, --keep-top-scores int  -K, --keep-unmatched  -H, --no-header-row\n"
      "  -g, --query-whole-file         -G, --use-filename  --query-id string\n"
      "  -N, --name-map file(s)         -D, --default-name-map\n"
      "  -w, --load-whole-db / --low-mem  accepted for compatibility (the index is always resident in HBM)\n"
      "  -j, --threads int  -i, --infile-list file  -q, --quiet  --log file\n"
      "GPU flags: --gpu int (device, default 0)  --gpus int (use devices 0..N-1, index blocks partitioned over them)\n"
      "           --gpu-ids a,b,c (explicit device list)  --gpu-batch int (queries per GPU call, default 131072)\n"
      "           --gpu-passes int (an index larger than the GPU's memory is searched in this many passes per batch, one part\n"
      "                             resident at a time; 0 = as few as fit; default: only when the index does not fit)\n"
      "           --parse-only (read the inputs and print records / bases / checksum per file; no database, no GPU)\n",
      stderr);
}

static double to_f(const std::string& flag, const std::string& v) {
  char* e = nullptr;
  double d = strtod(v.c_str(), &e);
  if (!e || *e || v.empty()) die("invalid argument \"%s\" for \"%s\" flag", v.c_str(), flag.c_str());
  return d;
}
static int to_i(const std::string& flag, const std::string& v) {
  char* e = nullptr;
  long d = strtol(v.c_str(), &e, 10);
  if (!e || *e || v.empty()) die("invalid argument \"%s\" for \"%s\" flag", v.c_str(), flag.c_str());
  return (int)d;
}

static Options parse_args(int argc, char** argv) {
  Options o;
  struct Spec { const char* lng; char sht; int kind; };  // kind 0 bool, 1 value
  static const Spec specs[] = {
      {"db-dir", 'd', 1}, {"out-file", 'o', 1}, {"read1", '1', 1}, {"read2", '2', 1}, {"try-se", 0, 0}, {"load-whole-db", 'w', 0},
      {"low-mem", 0, 0}, {"kmer-dedup-threshold", 'u', 1}, {"query-whole-file", 'g', 0}, {"use-filename", 'G', 0}, {"query-id", 0, 1},
      {"min-kmers", 'c', 1}, {"min-query-len", 'm', 1}, {"min-query-cov", 't', 1}, {"min-target-cov", 'T', 1}, {"max-fpr", 'f', 1},
      {"name-map", 'N', 1}, {"default-name-map", 'D', 0}, {"keep-unmatched", 'K', 0}, {"keep-top-scores", 'n', 1}, {"no-header-row", 'H', 0},
      {"sort-by", 's', 1}, {"do-not-sort", 'S', 0}, {"threads", 'j', 1}, {"quiet", 'q', 0}, {"infile-list", 'i', 1}, {"log", 0, 1},
      {"gpu", 0, 1}, {"gpu-batch", 0, 1}, {"gpus", 0, 1}, {"gpu-ids", 0, 1}, {"gpu-passes", 0, 1}, {"parse-only", 0, 0}, {"help", 'h', 0}, {"version", 'V', 0}};
  auto apply = [&](const std::string& name, const std::string& v) {
    if (name == "db-dir") o.db_dir = v;
    else if (name == "out-file") o.out_file = v;
    else if (name == "read1") o.read1 = v;
    else if (name == "read2") o.read2 = v;
    else if (name == "try-se") o.try_se = true;
    else if (name == "load-whole-db") o.load_whole = true;
    else if (name == "low-mem") o.low_mem = true;
    else if (name == "kmer-dedup-threshold") o.dedup = to_i(name, v);
    else if (name == "query-whole-file") o.whole_file = true;
    else if (name == "use-filename") o.use_filename = true;
    else if (name == "query-id") o.query_id = v;
    else if (name == "min-kmers") o.min_kmers = to_i(name, v);
    else if (name == "min-query-len") o.min_qlen = to_i(name, v);
    else if (name == "min-query-cov") o.min_qcov = to_f(name, v);
    else if (name == "min-target-cov") o.min_tcov = to_f(name, v);
    else if (name == "max-fpr") o.max_fpr = to_f(name, v);
    else if (name == "name-map") {
      size_t b = 0;  // StringSlice: comma separated and repeatable
      while (b <= v.size()) {
        size_t e = v.find(',', b);
        if (e == std::string::npos) e = v.size();
        if (e > b) o.name_maps.push_back(v.substr(b, e - b));
        b = e + 1;
      }
    } else if (name == "default-name-map") o.default_name_map = true;
    else if (name == "keep-unmatched") o.keep_unmatched = true;
    else if (name == "keep-top-scores") o.top_scores = to_i(name, v);
    else if (name == "no-header-row") o.no_header = true;
    else if (name == "sort-by") o.sort_by = v;
    else if (name == "do-not-sort") o.do_not_sort = true;
    else if (name == "threads") o.threads = to_i(name, v);
    else if (name == "quiet") o.quiet = true;
    else if (name == "infile-list") o.infile_list = v;
    else if (name == "log") o.log_file = v;
    else if (name == "gpu") o.device = to_i(name, v);
    else if (name == "gpu-batch") { o.batch = to_i(name, v); o.batch_given = true; }
    else if (name == "gpu-passes") o.gpu_passes = to_i(name, v);
    else if (name == "parse-only") o.parse_only = true;
    else if (name == "gpus") o.gpus = to_i(name, v);
    else if (name == "gpu-ids") {
      size_t b = 0;
      while (b <= v.size()) {
        size_t e = v.find(',', b);
        if (e == std::string::npos) e = v.size();
        if (e > b) o.gpu_ids.push_back(to_i(name, v.substr(b, e - b)));
        b = e + 1;
      }
    }
    else if (name == "help") { usage(); exit(0); }
    else if (name == "version") { printf("kmcp-search v%s\n", VERSION); exit(0); }
  };
  bool only_pos = false;
  int first = 1;
  // `kmcp-search search ...` = `kmcp search ...`: cobra's sub-command word, accepted (only) as the first argument
  if (argc > 1 && strcmp(argv[1], "search") == 0) first = 2;
  for (int i = first; i < argc; i++) {
    std::string a = argv[i];
    if (only_pos || a == "-" || a.empty() || a[0] != '-') { o.files.push_back(a); continue; }
    if (a == "--") { only_pos = true; continue; }
    if (a[1] == '-') {
      std::string name = a.substr(2), val;
      bool has = false;
      size_t eq = name.find('=');
      if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has = true; }
      const Spec* sp = nullptr;
      for (const auto& s : specs) if (name == s.lng) sp = &s;
      if (!sp) die("unknown flag: --%s", name.c_str());
      if (sp->kind == 1 && !has) {
        if (i + 1 >= argc) die("flag needs an argument: --%s", name.c_str());
        val = argv[++i];
      }
      apply(sp->lng, val);
    } else {
      for (size_t p = 1; p < a.size(); p++) {
        const Spec* sp = nullptr;
        for (const auto& s : specs) if (s.sht && a[p] == s.sht) sp = &s;
        if (!sp) die("unknown shorthand flag: '%c' in %s", a[p], a.c_str());
        if (sp->kind == 0) { apply(sp->lng, ""); continue; }
        std::string val = a.substr(p + 1);
        if (!val.empty() && val[0] == '=') val = val.substr(1);
        if (val.empty()) {
          if (i + 1 >= argc) die("flag needs an argument: '%c' in %s", a[p], a.c_str());
          val = argv[++i];
        }
        apply(sp->lng, val);
        break;
      }
    }
  }
  return o;
}

#include "fastx_reader.hpp"

// ------------------------------------------------------------------------------------------------
// pipeline
// ------------------------------------------------------------------------------------------------
// one 64-bit word per (query, column, count) tuple, summed mod 2^64 over a run: bench.py's hits_checksum (kmcp_amd/dist.py) in C++
static inline uint64_t tuple_mix(uint64_t query, uint32_t col, uint32_t count) {
  uint64_t x = query * 0x9E3779B97F4A7C15ULL + (uint64_t)col * 0xC2B2AE3D27D4EB4FULL + (uint64_t)count * 0x165667B19E3779F9ULL;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}

// Host cores this process may use: the affinity mask capped by the cgroup CPU quota (a GPU box shows 256 hardware threads and grants 16)
static unsigned usable_cpus() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)std::max(1, CPU_COUNT(&set));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long long period = 0;
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1ll, atoll(q) / period));
    fclose(f);
  }
  return n;
}

// On a two-socket host the threads of this process — parsers, searchers, formatters, the flusher — pass every batch from one to the next;
// spread over both sockets each hand-over crosses the interconnect (formatting cost 1.5-2x the thread-seconds, profiles/r06_cli_e2e.txt).
// When the affinity mask spans several NUMA nodes and one node has the cores the CPU quota grants anyway, the process keeps to the node it
// was started on (KMCP_SEARCH_NUMA=<node> picks another, KMCP_SEARCH_NUMA=off leaves the mask alone).  Called before any thread exists.
static void keep_to_one_numa_node() {
  const char* env = getenv("KMCP_SEARCH_NUMA");
  if (env && (!strcmp(env, "off") || !strcmp(env, "no"))) return;
  cpu_set_t mask;
  if (sched_getaffinity(0, sizeof mask, &mask) != 0) return;
  std::vector<cpu_set_t> nodes;
  for (int n = 0; n < 64; n++) {
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
    FILE* f = fopen(path, "r");
    if (!f) break;
    char buf[4096];
    cpu_set_t cs;
    CPU_ZERO(&cs);
    if (fgets(buf, sizeof buf, f))
      for (char* tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int got = sscanf(tok, "%d-%d", &a, &b);
        if (got == 1) b = a;
        if (got >= 1)
          for (int c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET(c, &cs);
      }
    fclose(f);
    cpu_set_t both;
    CPU_AND(&both, &cs, &mask);
    nodes.push_back(both);
  }
  int spanned = 0;
  for (const auto& n : nodes) spanned += CPU_COUNT(&n) > 0;
  if (spanned < 2) return;
  int want = -1;
  if (env && *env >= '0' && *env <= '9') want = atoi(env);
  else {
    const int cpu = sched_getcpu();
    for (size_t n = 0; n < nodes.size(); n++)
      if (cpu >= 0 && CPU_ISSET(cpu, &nodes[n])) want = (int)n;
  }
  if (want < 0 || want >= (int)nodes.size() || (unsigned)CPU_COUNT(&nodes[(size_t)want]) < std::min(usable_cpus(), 4u)) return;
  (void)sched_setaffinity(0, sizeof(cpu_set_t), &nodes[(size_t)want]);
}

struct Batch {
  uint64_t seq = 0;  // position in the input: the writer emits batches in this order
  uint64_t n_seq = 1;  // how many of the reader's batches this one holds (batches read before the database was open are joined for a paged index)
  uint64_t first_idx = 0;
  std::vector<char> id_buf;  // query IDs back to back
  std::vector<uint64_t> id_offs{0};
  std::vector<uint8_t> seqs, seqs2;
  std::vector<uint64_t> offs{0}, offs2{0};
  kmcpg_result_pairs res{};  // compact result: (column, mKmers) pairs; the formatter threads expand a query's pairs right before its rows
  bool paired = false;
  // -g queries (whole files) are packed where the reader first touches their bases: 2-bit codes + the runs of other bytes
  // (kmcp_gpu.h kmcpg_pack2 / kmcpg_submit_packed); `seqs` stays empty, `offs` counts bases as ever
  bool packed = false;
  std::vector<uint8_t> codes;
  std::vector<kmcpg_exc_run> exc;  // size = capacity; n_exc of them are in use
  uint64_t n_exc = 0, n_bases = 0;
  void pack_append(const char* s, size_t n) {
    const size_t need = (size_t)((n_bases + n + 3) / 4 + 8);
    if (codes.size() < need) codes.resize(std::max(need, codes.size() + codes.size() / 2 + (1u << 20)));
    if (exc.size() < n_exc + 64) exc.resize(std::max<size_t>(1024, 2 * exc.size()));
    for (;;) {
      const uint64_t before = n_exc;
      const int rc = kmcpg_pack2((const uint8_t*)s, n, n_bases, codes.data(), exc.data(), exc.size(), &n_exc);
      if (rc == 0) break;
      if (rc != KMCPG_ENOMEM) die("%s", kmcpg_last_error());
      exc.resize(std::max<size_t>(2 * exc.size(), (size_t)n_exc + 1024));  // n_exc = how many runs there are in all
      n_exc = before;
    }
    n_bases += n;
  }
  // a whole query that was packed on its own (from base 0 of `src`): its codes are moved behind this batch's — a plain copy when the batch
  // ends on a byte, two shifts per byte otherwise — and its runs shifted to their place
  void append_packed(const uint8_t* src, uint64_t nb, const kmcpg_exc_run* runs, uint64_t n_runs) {
    const size_t need = (size_t)((n_bases + nb + 3) / 4 + 8);
    if (codes.size() < need) codes.resize(std::max(need, codes.size() + codes.size() / 2 + (1u << 20)));
    const size_t nbytes = (size_t)((nb + 3) / 4);
    const unsigned sh = 2u * (unsigned)(n_bases & 3);
    uint8_t* d = codes.data() + (n_bases >> 2);
    if (sh == 0) {
      memcpy(d, src, nbytes);
    } else {
      unsigned carry = d[0] & ((1u << sh) - 1u);
      for (size_t i = 0; i < nbytes; i++) {
        const unsigned v = src[i];
        d[i] = (uint8_t)(carry | (v << sh));
        carry = v >> (8 - sh);
      }
      d[nbytes] = (uint8_t)carry;
    }
    if (exc.size() < n_exc + n_runs) exc.resize(std::max<size_t>((size_t)(n_exc + n_runs), 2 * exc.size()));
    for (uint64_t i = 0; i < n_runs; i++) exc[n_exc + i] = kmcpg_exc_run{runs[i].pos + n_bases, runs[i].len, runs[i].byte};
    n_exc += n_runs;
    n_bases += nb;
  }
  uint64_t bases() const { return packed ? n_bases : (uint64_t)(seqs.size() + seqs2.size()); }
  size_t size() const { return id_offs.size() - 1; }
  std::string_view id(size_t i) const { return std::string_view(id_buf.data() + id_offs[i], (size_t)(id_offs[i + 1] - id_offs[i])); }
  // the queries of `o` (the reader's next batch) behind this one's
  void append(const Batch& o) {
    const uint64_t ib = id_buf.size(), sb = seqs.size(), sb2 = seqs2.size();
    id_buf.insert(id_buf.end(), o.id_buf.begin(), o.id_buf.end());
    for (size_t i = 1; i < o.id_offs.size(); i++) id_offs.push_back(ib + o.id_offs[i]);
    seqs.insert(seqs.end(), o.seqs.begin(), o.seqs.end());
    for (size_t i = 1; i < o.offs.size(); i++) offs.push_back(sb + o.offs[i]);
    if (paired) {
      seqs2.insert(seqs2.end(), o.seqs2.begin(), o.seqs2.end());
      for (size_t i = 1; i < o.offs2.size(); i++) offs2.push_back(sb2 + o.offs2[i]);
    }
    n_seq += o.n_seq;
  }
};

template <typename T>
class Queue {
 public:
  explicit Queue(size_t cap) : cap_(cap) {}
  void push(T v) {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return q_.size() < cap_; });
    q_.push_back(std::move(v));
    cv_.notify_all();
  }
  bool pop(T* v) {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    *v = std::move(q_.front());
    q_.pop_front();
    cv_.notify_all();
    return true;
  }
  bool try_pop(T* v) {  // what is there right now, without waiting
    std::lock_guard<std::mutex> l(m_);
    if (q_.empty()) return false;
    *v = std::move(q_.front());
    q_.pop_front();
    cv_.notify_all();
    return true;
  }
  void close() {
    std::lock_guard<std::mutex> l(m_);
    closed_ = true;
    cv_.notify_all();
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
  size_t cap_;
  bool closed_ = false;
};

// The records of one single-end input file as batches of about `batch_reads` queries, in file order.  Plain four-line FASTQ
// files are cut and parsed by several threads (ParallelFastq: every chunk becomes a batch without another copy); everything
// else — gzip, BGZF, FASTA, wrapped FASTQ, pipes — goes through the single-threaded FastxReader.  Returns the number of records.
// `stop` (optional) is looked at between batches / records: once set the rest of the file is left unread.
template <class Emit>
static uint64_t read_single_end(const std::string& file, size_t batch_reads, size_t max_bases, Emit&& emit, const std::atomic<bool>* stop = nullptr) {
  uint64_t n = 0;
  std::unique_ptr<Batch> b(new Batch());
  auto flush = [&] {
    if (b->size() == 0) return;
    n += b->size();
    emit(std::move(b));
    b.reset(new Batch());
  };
  auto add = [&](const FastxRec& r) {
    b->id_buf.insert(b->id_buf.end(), r.id, r.id + r.id_len);
    b->id_offs.push_back(b->id_buf.size());
    b->seqs.insert(b->seqs.end(), (const uint8_t*)r.seq, (const uint8_t*)r.seq + r.seq_len);
    b->offs.push_back(b->seqs.size());
    if (b->size() >= batch_reads || b->seqs.size() >= max_bases) flush();
  };
  uint64_t resume = 0;
  bool serial = true;
  if (ParallelFastq::eligible(file)) {
    int w = (int)std::min(8u, std::max(2u, usable_cpus() / 2));
    if (const char* e = getenv("KMCP_READER_THREADS")) w = std::max(1, atoi(e));
    ParallelFastq pf(file, batch_reads, w, 2 * max_bases);  // a record is its bases twice (qualities) plus the header
    serial = false;
    while (std::unique_ptr<FastqChunk> c = pf.next()) {
      if (stop && stop->load(std::memory_order_relaxed)) return n;
      if (!c->strict) {  // not four-line FASTQ from here on: the general reader takes over at the chunk's first byte
        resume = c->file_off;
        serial = true;
        break;
      }
      if (c->size() == 0) continue;
      std::unique_ptr<Batch> cb(new Batch());
      cb->id_buf.swap(c->id_buf);
      cb->id_offs.swap(c->id_offs);
      cb->seqs.swap(c->seqs);
      cb->offs.swap(c->offs);
      n += cb->size();
      emit(std::move(cb));
    }
  }
  if (serial) {
    FastxReader r(file, resume);
    FastxRec rec;
    while (!(stop && stop->load(std::memory_order_relaxed)) && r.next(&rec)) add(rec);
    if (!(stop && stop->load(std::memory_order_relaxed))) flush();
  }
  return n;
}

// The records of two mate files as batches of pairs (IDs of read 1), in file order; ends with the shorter file, like the
// reference's loop (search.go:807-826).  Both files go through read_single_end — several parser threads each for plain FASTQ —
// the mates on a thread of their own; read 2's batches are re-cut at read 1's batch boundaries (buffers are taken over
// without a copy where the boundaries agree, which they do for reads of equal length).  Returns the number of pairs.
template <class Emit>
static uint64_t read_paired(const std::string& file1, const std::string& file2, size_t batch_reads, size_t max_bases, Emit&& emit) {
  Queue<std::unique_ptr<Batch>> q2(4);
  // the pairs end with the shorter file (search.go:807-826): whichever reader is still going when the other file is exhausted
  // stops at its next batch instead of parsing the rest of a file nobody will look at
  std::atomic<bool> ended{false}, stop2{false};
  std::thread mate_reader([&] {
    read_single_end(file2, batch_reads, std::max<size_t>(1, max_bases / 2), [&](std::unique_ptr<Batch> b) { q2.push(std::move(b)); }, &stop2);
    q2.close();
  });
  std::unique_ptr<Batch> cur;  // the batch of read 2 being consumed
  size_t ci = 0;               // records of it already handed out
  uint64_t n = 0;
  read_single_end(file1, batch_reads, std::max<size_t>(1, max_bases / 2), [&](std::unique_ptr<Batch> b) {
    if (ended) return;
    const size_t want = b->size();
    size_t have = 0;
    b->paired = true;
    while (have < want) {
      if (!cur || ci == cur->size()) {
        ci = 0;
        cur.reset();
        if (!q2.pop(&cur)) {
          ended = true;
          break;
        }
        continue;
      }
      if (have == 0 && ci == 0 && cur->size() == want) {
        b->seqs2.swap(cur->seqs);
        b->offs2.swap(cur->offs);
        cur.reset();
        have = want;
        break;
      }
      const size_t take = std::min(want - have, cur->size() - ci);
      const uint64_t lo = cur->offs[ci], hi = cur->offs[ci + take], base = b->seqs2.size();
      b->seqs2.insert(b->seqs2.end(), cur->seqs.begin() + (ptrdiff_t)lo, cur->seqs.begin() + (ptrdiff_t)hi);
      for (size_t i = 1; i <= take; i++) b->offs2.push_back(base + (cur->offs[ci + i] - lo));
      ci += take;
      have += take;
    }
    if (have < want) {  // read 2 ended inside this batch
      b->id_buf.resize((size_t)b->id_offs[have]);
      b->id_offs.resize(have + 1);
      b->seqs.resize((size_t)b->offs[have]);
      b->offs.resize(have + 1);
    }
    if (have == 0) return;
    n += have;
    emit(std::move(b));
  }, &ended);
  stop2 = true;            // read 1 ended first (or both did): the mates' thread stops at its next batch
  while (q2.pop(&cur)) {}  // ... and is not left blocked on a full queue
  mate_reader.join();
  return n;
}

// one complete gzip member holding `in` (deflate level 6 as compress/gzip's default in the reference's outStream)
static std::string gzip_member(const std::string& in) {
  z_stream z;
  memset(&z, 0, sizeof z);
  if (deflateInit2(&z, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("zlib: deflateInit2 failed");
  std::string out;
  out.resize(deflateBound(&z, (uLong)in.size()) + 64);
  z.next_in = (Bytef*)in.data();
  z.avail_in = (uInt)in.size();
  z.next_out = (Bytef*)&out[0];
  z.avail_out = (uInt)out.size();
  if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("zlib: deflate failed");
  out.resize(z.total_out);
  deflateEnd(&z);
  return out;
}

class Out {
 public:
  explicit Out(const std::string& path) {
    gz_ = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    f_ = path == "-" ? stdout : fopen(path.c_str(), "wb");
    if (!f_) die("%s: %s", path.c_str(), strerror(errno));
  }
  bool gz() const { return gz_; }
  // text: compressed here when the file is .gz
  void write(const std::string& s) {
    if (s.empty()) return;
    if (gz_) write_raw(gzip_member(s));
    else write_raw(s);
  }
  // bytes that are already in the file's encoding
  void write_raw(const std::string& s) {
    if (!s.empty() && fwrite(s.data(), 1, s.size(), f_) != s.size()) die("write failed: %s", strerror(errno));
  }
  void close() {
    if (f_ != stdout) fclose(f_);
    else fflush(f_);
  }

 private:
  bool gz_ = false;
  FILE* f_ = nullptr;
};

// ---- TSV rows.  Number formatting must equal Go's strconv (FormatFloat 'f',4 / 'e',4 = correctly rounded decimals, which is
// what printf gives); the fast paths below produce the same digits and fall back to snprintf whenever a rounding tie is near.
struct RowFormatter {
  // text buffers this formatter has filled before (the flusher hands them back): the next part is written where this thread's last ones were
  std::mutex free_mu;
  std::vector<std::string> free_bufs;
  void take(std::string& into) {
    std::lock_guard<std::mutex> g(free_mu);
    if (free_bufs.empty()) return;
    into = std::move(free_bufs.back());
    free_bufs.pop_back();
    into.clear();
  }
  void give_back(std::string&& s) {
    if (s.capacity() > (1ull << 30)) return;
    std::lock_guard<std::mutex> g(free_mu);
    if (free_bufs.size() < 64) free_bufs.push_back(std::move(s));
  }
  char tmp[64];
  std::vector<kmcpg_match> scratch;  // the records of the query being formatted (kmcpg_expand_pairs)
  std::unordered_map<uint64_t, std::string> fpr_cache;  // the FPR of a match depends on (qKmers, mKmers) only

  // A row is assembled in a fixed scratch line through a moving pointer (no capacity checks per character) and appended to the
  // batch's text in one go; rows that could not fit (IDs or target names of kilobytes) take the std::string path below.
  static char* w_u64(char* p, uint64_t v) {
    char t[24];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = t[--n];
    return p;
  }
  static char* w_i(char* p, int64_t v) {
    if (v < 0) { *p++ = '-'; return w_u64(p, (uint64_t)(-v)); }
    return w_u64(p, (uint64_t)v);
  }
  static char* w_f4(char* p, double v) {  // "%.4f"
    if (v >= 0 && v < 1e5) {  // v * 10000 < 1e9: its rounding error (< 2e-7) cannot carry the fraction across the 1e-6 guard below
      const double sc = v * 10000.0;
      const double fl = floor(sc);
      const double fr = sc - fl;
      if (fabs(fr - 0.5) > 1e-6) {  // far from a tie: the scaled value rounds like the exact decimal expansion
        const uint64_t q = (uint64_t)fl + (fr > 0.5 ? 1 : 0);
        p = w_u64(p, q / 10000);
        *p++ = '.';
        const unsigned f = (unsigned)(q % 10000);
        *p++ = (char)('0' + f / 1000);
        *p++ = (char)('0' + f / 100 % 10);
        *p++ = (char)('0' + f / 10 % 10);
        *p++ = (char)('0' + f % 10);
        return p;
      }
    }
    return p + snprintf(p, 48, "%.4f", v);
  }
  static void put_u64(std::string& b, uint64_t v) {
    char t[24];
    b.append(t, (size_t)(w_u64(t, v) - t));
  }
  static void put_i(std::string& b, int64_t v) {
    char t[24];
    b.append(t, (size_t)(w_i(t, v) - t));
  }
  void put_f4(std::string& b, double v) {
    char t[64];
    b.append(t, (size_t)(w_f4(t, v) - t));
  }
  // FPR strings of short queries by (n, c) in a table, the rest in a map
  std::vector<std::vector<std::string>> fpr_tab;
  const std::string& fpr(int n, int c, double v) {
    if (n > 0 && n <= 4096 && c >= 0 && c <= n) {
      if (fpr_tab.empty()) fpr_tab.resize(4097);
      std::vector<std::string>& row_of_n = fpr_tab[(size_t)n];
      if (row_of_n.empty()) row_of_n.resize((size_t)n + 1);
      std::string& e = row_of_n[(size_t)c];
      if (e.empty()) e.assign(tmp, (size_t)snprintf(tmp, sizeof tmp, "%.4e", v));
      return e;
    }
    const uint64_t key = ((uint64_t)(uint32_t)n << 32) | (uint32_t)c;
    auto it = fpr_cache.find(key);
    if (it != fpr_cache.end()) return it->second;
    if (fpr_cache.size() > (1u << 20)) fpr_cache.clear();
    return fpr_cache.emplace(key, std::string(tmp, (size_t)snprintf(tmp, sizeof tmp, "%.4e", v))).first->second;
  }
  static constexpr size_t LINE = 8192;
  char line[LINE];
  void row(std::string& b, std::string_view id, int qlen, int qkmers, uint64_t hits, const std::string& target, const kmcpg_match& m, int k,
           uint64_t qidx) {
    const std::string& f = fpr(qkmers, m.mkmers, m.fpr);
    if (id.size() + target.size() + f.size() + 400 > LINE) {  // oversized names: the slow, unbounded path
      b += id; b.push_back('\t'); put_i(b, qlen); b.push_back('\t'); put_i(b, qkmers); b.push_back('\t');
      b += f; b.push_back('\t'); put_u64(b, hits); b.push_back('\t');
      b += target; b.push_back('\t'); put_u64(b, (uint16_t)m.target_idx); b.push_back('\t'); put_u64(b, m.target_idx >> 16); b.push_back('\t');
      put_u64(b, m.gsize); b.push_back('\t'); put_i(b, k); b.push_back('\t'); put_i(b, m.mkmers); b.push_back('\t');
      put_f4(b, m.qcov); b.push_back('\t'); put_f4(b, m.tcov); b.push_back('\t'); put_f4(b, m.jacc); b.push_back('\t');
      put_u64(b, qidx); b.push_back('\n');
      return;
    }
    char* p = line;
    memcpy(p, id.data(), id.size()); p += id.size(); *p++ = '\t';
    p = w_i(p, qlen); *p++ = '\t';
    p = w_i(p, qkmers); *p++ = '\t';
    memcpy(p, f.data(), f.size()); p += f.size(); *p++ = '\t';
    p = w_u64(p, hits); *p++ = '\t';
    memcpy(p, target.data(), target.size()); p += target.size(); *p++ = '\t';
    p = w_u64(p, (uint16_t)m.target_idx); *p++ = '\t';
    p = w_u64(p, m.target_idx >> 16); *p++ = '\t';
    p = w_u64(p, m.gsize); *p++ = '\t';
    p = w_i(p, k); *p++ = '\t';
    p = w_i(p, m.mkmers); *p++ = '\t';
    p = w_f4(p, m.qcov); *p++ = '\t';
    p = w_f4(p, m.tcov); *p++ = '\t';
    p = w_f4(p, m.jacc); *p++ = '\t';
    p = w_u64(p, qidx); *p++ = '\n';
    b.append(line, (size_t)(p - line));
  }
  // All rows of one query.  With many matches (a database full of close relatives: hundreds per read) what is the same in every
  // row — ID, qLen, qKmers in front, hits, kSize, queryIdx — is formatted once, and what depends on the column only (target,
  // chunkIdx, chunks, tLen) once per column and formatter thread; a row then costs one integer, three fixed-point numbers and
  // a few copies.
  std::vector<std::string> col_text;  // "target\tchunkIdx\tchunks\ttLen\t" by column, filled on first use
  void rows(std::string& b, std::string_view id, int qlen, int qkmers, const kmcpg_match* ms, uint64_t cnt, const std::vector<std::string>& target,
            int k, uint64_t qidx) {
    if (cnt < 4 || id.size() > 1024) {
      for (uint64_t j = 0; j < cnt; j++) row(b, id, qlen, qkmers, cnt, target[ms[j].col], ms[j], k, qidx);
      return;
    }
    char pre[1024 + 64], mid[32], ks[24], suf[32];
    char* q = pre;
    memcpy(q, id.data(), id.size()); q += id.size(); *q++ = '\t';
    q = w_i(q, qlen); *q++ = '\t';
    q = w_i(q, qkmers); *q++ = '\t';
    const size_t pre_n = (size_t)(q - pre);
    q = w_u64(mid, cnt); *q++ = '\t';
    const size_t mid_n = (size_t)(q - mid);
    q = w_i(ks, k); *q++ = '\t';
    const size_t ks_n = (size_t)(q - ks);
    q = w_u64(suf, qidx); *q++ = '\n';
    const size_t suf_n = (size_t)(q - suf);
    if (col_text.size() < target.size()) col_text.resize(target.size());
    for (uint64_t j = 0; j < cnt; j++) {
      const kmcpg_match& m = ms[j];
      std::string& ct = col_text[m.col];
      if (ct.empty()) {
        ct = target[m.col];
        ct.push_back('\t'); put_u64(ct, (uint16_t)m.target_idx);
        ct.push_back('\t'); put_u64(ct, m.target_idx >> 16);
        ct.push_back('\t'); put_u64(ct, m.gsize);
        ct.push_back('\t');
      }
      const std::string& f = fpr(qkmers, m.mkmers, m.fpr);
      if (pre_n + f.size() + ct.size() + 400 > LINE) {
        row(b, id, qlen, qkmers, cnt, target[m.col], m, k, qidx);
        continue;
      }
      char* p = line;
      memcpy(p, pre, pre_n); p += pre_n;
      memcpy(p, f.data(), f.size()); p += f.size(); *p++ = '\t';
      memcpy(p, mid, mid_n); p += mid_n;
      memcpy(p, ct.data(), ct.size()); p += ct.size();
      memcpy(p, ks, ks_n); p += ks_n;
      p = w_i(p, m.mkmers); *p++ = '\t';
      p = w_f4(p, m.qcov); *p++ = '\t';
      p = w_f4(p, m.tcov); *p++ = '\t';
      p = w_f4(p, m.jacc); *p++ = '\t';
      memcpy(p, suf, suf_n); p += suf_n;
      b.append(line, (size_t)(p - line));
    }
  }
  void unmatched(std::string& b, std::string_view id, int qlen, int qkmers, int k, uint64_t qidx) {
    b += id; b.push_back('\t'); put_i(b, qlen); b.push_back('\t'); put_i(b, qkmers);
    b += "\t0\t0\t\t-1\t0\t0\t"; put_i(b, k); b += "\t0\t0\t0\t0\t"; put_u64(b, qidx); b.push_back('\n');
  }
};

// Formatter threads that live as long as the run: each keeps its RowFormatter (and with it the cache of FPR strings, which
// a fresh formatter per batch would fill again and again).
class FormatPool {
 public:
  explicit FormatPool(int n) {
    for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); });
  }
  ~FormatPool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
      cv_.notify_all();
    }
    for (auto& t : th_) t.join();
  }
  // fn(part, formatter) for part = 0 .. parts-1, spread over the pool; returns when all are done
  void run(int parts, const std::function<void(int, RowFormatter&)>& fn) {
    std::unique_lock<std::mutex> l(m_);
    fn_ = &fn;
    next_ = 0;
    parts_ = parts;
    left_ = parts;
    cv_.notify_all();
    done_cv_.wait(l, [&] { return left_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop() {
    RowFormatter F;
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      cv_.wait(l, [&] { return stop_ || (fn_ && next_ < parts_); });
      if (stop_) return;
      const int pi = next_++;
      const auto* fn = fn_;
      l.unlock();
      (*fn)(pi, F);
      l.lock();
      if (--left_ == 0) done_cv_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int, RowFormatter&)>* fn_ = nullptr;
  int next_ = 0, parts_ = 0, left_ = 0;
  bool stop_ = false;
};

static std::unordered_map<std::string, std::string> read_kvs(const std::string& file) {  // cliutil.ReadKVs
  std::unordered_map<std::string, std::string> m;
  gzFile g = gzopen(file.c_str(), "rb");
  if (!g) die("%s: %s", file.c_str(), strerror(errno));
  char buf[1 << 16];
  while (gzgets(g, buf, sizeof buf)) {
    size_t n = strlen(buf);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
    if (!n || buf[0] == '#') continue;
    char* tab = strchr(buf, '\t');
    if (!tab) continue;
    *tab = 0;
    m[buf] = tab + 1;
  }
  gzclose(g);
  return m;
}

static std::string trim_ext(const std::string& path) {  // filepathTrimExtension: basename without (.gz/.xz/..)+ext
  size_t s = path.find_last_of('/');
  std::string b = s == std::string::npos ? path : path.substr(s + 1);
  for (const char* z : {".gz", ".xz", ".zst", ".bz2"}) {
    size_t l = strlen(z);
    if (b.size() > l && b.compare(b.size() - l, l, z) == 0) { b.resize(b.size() - l); break; }
  }
  size_t d = b.find_last_of('.');
  if (d != std::string::npos && d > 0) b.resize(d);
  return b;
}

int main(int argc, char** argv) {
  Options o = parse_args(argc, argv);
  g_quiet = o.quiet;
  if (!o.log_file.empty()) {
    g_log = fopen(o.log_file.c_str(), "w");
    if (!g_log) die("%s: %s", o.log_file.c_str(), strerror(errno));
  }
  const bool verbose = !o.quiet;
  const auto t_start = std::chrono::steady_clock::now();
  keep_to_one_numa_node();
  if (o.parse_only) {  // reader check, no database and no GPU: one summary line per input file
    // checksum = sum over records i (0-based, in file order) of fnv1a("id\tseq\n") * (2 i + 1) mod 2^64: order-sensitive, yet
    // every batch can be summed on its own thread
    // (pairs, -1/-2: "id\tseq1\tseq2\n")
    const bool pe = !o.read1.empty() && !o.read2.empty();
    std::vector<std::string> inputs = pe ? std::vector<std::string>{o.read1 + "," + o.read2} : o.files;
    for (const auto& file : inputs) {
      const auto t0 = std::chrono::steady_clock::now();
      Queue<std::unique_ptr<Batch>> q(8);
      std::mutex mu;
      uint64_t n = 0, bases = 0, id_bytes = 0, sum = 0;
      std::vector<std::thread> th;
      for (int t = 0; t < 4; t++)
        th.emplace_back([&] {
          std::unique_ptr<Batch> b;
          uint64_t my_sum = 0, my_bases = 0, my_ids = 0, my_n = 0;
          while (q.pop(&b)) {
            for (size_t i = 0; i < b->size(); i++) {
              uint64_t h = 1469598103934665603ULL;
              auto mix = [&](const char* p, size_t len) {
                for (size_t j = 0; j < len; j++) h = (h ^ (uint8_t)p[j]) * 1099511628211ULL;
              };
              mix(b->id_buf.data() + b->id_offs[i], (size_t)(b->id_offs[i + 1] - b->id_offs[i]));
              mix("\t", 1);
              mix((const char*)b->seqs.data() + b->offs[i], (size_t)(b->offs[i + 1] - b->offs[i]));
              if (b->paired) {
                mix("\t", 1);
                mix((const char*)b->seqs2.data() + b->offs2[i], (size_t)(b->offs2[i + 1] - b->offs2[i]));
              }
              mix("\n", 1);
              my_sum += h * (2 * (b->first_idx + i) + 1);
            }
            my_n += b->size();
            my_bases += b->seqs.size() + b->seqs2.size();
            my_ids += b->id_buf.size();
          }
          std::lock_guard<std::mutex> g(mu);
          sum += my_sum;
          n += my_n;
          bases += my_bases;
          id_bytes += my_ids;
        });
      uint64_t idx = 0;
      auto emit = [&](std::unique_ptr<Batch> b) {
        b->first_idx = idx;
        idx += b->size();
        q.push(std::move(b));
      };
      if (pe) read_paired(o.read1, o.read2, (size_t)o.batch, 64u << 20, emit);
      else read_single_end(file, (size_t)o.batch, 64u << 20, emit);
      q.close();
      for (auto& t : th) t.join();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("%s\trecords=%llu\tbases=%llu\tid_bytes=%llu\tfnv1a=%016llx\n", file.c_str(), (unsigned long long)n, (unsigned long long)bases,
             (unsigned long long)id_bytes, (unsigned long long)sum);
      if (!o.quiet) fprintf(stderr, "%s: %.3f s, %.2f M records/s\n", file.c_str(), dt, n / dt / 1e6);
    }
    return 0;
  }
  if (o.db_dir.empty()) die("flag -d/--db-dir needed");
  if (o.min_kmers < 1) die("value of flag --min-kmers should be positive: %d", o.min_kmers);
  if (o.dedup < 1) die("value of flag --kmer-dedup-threshold should be positive: %d", o.dedup);
  if (!(o.max_fpr > 0)) die("value of flag --max-fpr should be positive: %f", o.max_fpr);
  if (o.min_qlen < 0 || o.top_scores < 0) die("value of flag --min-query-len/--keep-top-scores should not be negative");
  if (o.do_not_sort && o.top_scores > 0) warn("flag -n/--keep-top-scores ignored when -S/--do-not-sort given");
  int sort_by = 0;
  if (o.sort_by == "qcov") sort_by = 0;
  else if (o.sort_by == "tcov") sort_by = 1;
  else if (o.sort_by == "jacc") sort_by = 2;
  else die("invalid value for flag -s/--sort-by: %s. Available: qcov/tsov/jacc", o.sort_by.c_str());
  if (o.min_qcov < 0 || o.min_qcov > 1) die("value of -t/--min-query-cov should be in range [0, 1]");
  if (o.min_tcov < 0 || o.min_tcov > 1) die("value of -T/-target-cov should be in range [0, 1]");
  if (verbose) {
    info("kmcp-search v%s (MI355X build of the kmcp search hot path)", VERSION);
    info("  https://github.com/shenwei356/kmcp");
    info("");
    info("checking input files ...");
  }

  // ---- input files (search.go:219-290)
  bool paired = false;
  std::vector<std::string> files;
  if (o.read1.empty()) {
    if (!o.read2.empty()) { warn("only flag -2/--read2 given, it's treated as single-end"); files.push_back(o.read2); }
  } else if (o.read2.empty()) {
    warn("only flag -1/--read1 given, it's treated as single-end");
    files.push_back(o.read1);
  } else {
    paired = true;
    if (verbose) { info("paired end files given: %s, %s", o.read1.c_str(), o.read2.c_str()); info("other input files via positional arguments are ignored"); }
  }
  if (o.try_se && !paired) { warn("flag --try-se ignored for single-end input(s)"); o.try_se = false; }
  if (!paired) {
    std::vector<std::string> f1 = o.files;
    if (!o.infile_list.empty()) {
      gzFile g = gzopen(o.infile_list.c_str(), "rb");
      if (!g) die("%s: %s", o.infile_list.c_str(), strerror(errno));
      char buf[1 << 14];
      while (gzgets(g, buf, sizeof buf)) {
        size_t n = strlen(buf);
        while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
        if (n) f1.push_back(buf);
      }
      gzclose(g);
    }
    if (f1.empty() && files.empty()) f1.push_back("-");
    for (const auto& f : f1) {
      if ((!o.read1.empty() || !o.read2.empty()) && f == "-") continue;
      files.push_back(f);
    }
    for (const auto& f : files) {
      struct stat st;
      if (f != "-" && stat(f.c_str(), &st) != 0) die("%s: %s", f.c_str(), strerror(errno));
      if (f != "-" && f == o.out_file) die("out file should not be one of the input file");
    }
    if (verbose) {
      if (files.size() == 1 && files[0] == "-") info("  no files given, reading from stdin");
      else info("  %zu input file(s) given", files.size());
    }
  }

  // ---- database: sub-directories holding __db.yml (search.go:299-324)
  if (verbose) info("checking the database: %s", o.db_dir.c_str());
  std::vector<std::string> db_dirs;
  {
    DIR* d = opendir(o.db_dir.c_str());
    if (!d) die("read database error: open %s: %s", o.db_dir.c_str(), strerror(errno));
    std::vector<std::string> subs;
    while (struct dirent* e = readdir(d)) {
      std::string n = e->d_name;
      if (n == "." || n == "..") continue;
      subs.push_back(n);
    }
    closedir(d);
    std::sort(subs.begin(), subs.end());
    for (const auto& n : subs) {
      struct stat st;
      std::string p = o.db_dir + "/" + n;
      if (stat(p.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) continue;
      if (stat((p + "/__db.yml").c_str(), &st) == 0) db_dirs.push_back(p);
    }
  }
  if (db_dirs.empty()) die("invalid kmcp database: %s", o.db_dir.c_str());
  if (db_dirs.size() > 1) die("databases with several repeats (R001, R002, ...) are not supported: `kmcp index` only writes R001");

  std::unordered_map<std::string, std::string> name_map;
  const bool mapping = !o.name_maps.empty();
  if (mapping) {
    if (verbose) info("loading name mapping file ...");
    for (const auto& f : o.name_maps)
      for (auto& kv : read_kvs(f)) name_map[kv.first] = kv.second;
    if (verbose) info("  %zu pairs of name mapping values from %zu file(s) loaded", name_map.size(), o.name_maps.size());
  }
  std::unordered_map<std::string, std::string> default_map;
  if (o.default_name_map) {
    struct stat st;
    std::string f = db_dirs[0] + "/__name_mapping.tsv";
    if (stat(f.c_str(), &st) == 0) default_map = read_kvs(f);
  }

  // ---- the reader starts NOW, before the database is opened: parsing the input needs neither the GPU nor the index, and the HIP
  //      runtime alone takes 0.2 s to come up (tools/ubench_init.cpp) — by the time the index is resident the first batches (up to
  //      q_in's capacity) are waiting.  Batch limits are the defaults until the open has finished; should the index turn out to be
  //      paged (larger than the GPU's memory: a batch then costs passes - 1 uploads), the early batches are joined into large ones
  //      before they are searched (Batch::append below).
  Queue<std::unique_ptr<Batch>> q_in(24), q_out(3);
  std::atomic<size_t> max_bases{(size_t)64 << 20}, batch_reads{(size_t)o.batch};
  std::atomic<int> db_k{0};
  std::atomic<bool> db_ready{false};
  std::mutex ready_mu;
  std::condition_variable ready_cv;
  auto wait_db = [&] {
    std::unique_lock<std::mutex> l(ready_mu);
    ready_cv.wait(l, [&] { return db_ready.load(); });
  };
  double t_reader_blocked = 0, t_reader_total = 0;  // the reader thread: waiting for a free queue slot / its whole life
  if (o.whole_file && o.gpu_passes < 0) {
    // -g needs the database's k before the first file can be joined (k - 1 N's between records): a metadata-only handle reads it from
    // __db.yml and the block headers in a millisecond, without the GPU runtime, so that the files are parsed while the index is loaded
    kmcpg_db* meta = nullptr;
    kmcpg_opts mo{-1, 0, 1, 0};
    if (kmcpg_open(db_dirs[0].c_str(), &mo, &meta) == 0) {
      kmcpg_info mi;
      if (kmcpg_db_info(meta, &mi) == 0) db_k.store(mi.k);
      kmcpg_close(meta);
    }
  }
  std::thread reader([&] {
    if (o.gpu_passes >= 0) wait_db();
    const auto tr0 = std::chrono::steady_clock::now();
    uint64_t id = 0, seq = 0;
    std::unique_ptr<Batch> b(new Batch());
    b->paired = paired;
    auto flush = [&] {
      if (b->size() == 0) return;
      b->seq = seq++;
      const auto tp = std::chrono::steady_clock::now();
      q_in.push(std::move(b));
      t_reader_blocked += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
      b.reset(new Batch());
      b->paired = paired;
      b->first_idx = id;
    };
    std::string id1, s1, id2, s2;
    if (paired) {
      if (verbose) info("reading from paired-end files: %s, %s", o.read1.c_str(), o.read2.c_str());
      flush();
      read_paired(o.read1, o.read2, batch_reads.load(), max_bases.load(), [&](std::unique_ptr<Batch> nb) {
        nb->first_idx = id;
        id += nb->size();
        b = std::move(nb);
        flush();
      });
      if (id == 0) warn("no valid sequences in files: %s, %s", o.read1.c_str(), o.read2.c_str());
    } else {
      std::string nnn;
      if (o.whole_file) {  // the gap between records is k - 1 N's: the database's k is needed first
        if (db_k.load() <= 0) wait_db();  // (normally known already: read from the headers before the GPU was touched, below)
        nnn.assign((size_t)std::max(0, db_k.load() - 1), 'N');
      }
      if (o.whole_file) {  // search.go:885-935
        // One query per file: the records of the file back to back, records 2..m each followed by k - 1 N's (search.go:899-914) — packed to
        // 2-bit codes as they are read (the file's bases are touched once, the batch is a quarter of the text and the library takes it as
        // it is).  The files are parsed by several threads, a file each (a 4-Mbp assembly is ~4 ms of line joining and packing: one reader
        // thread fed 250 genomes/s to a GPU that searches 40 000), and joined into batches in the order of the command line.
        struct FileQuery {
          Batch q;  // the file's one query, packed from base 0
          std::string qid;
          bool empty = true, done = false;
        };
        const size_t nf = files.size();
        std::vector<std::unique_ptr<FileQuery>> slots(nf);
        std::mutex fm;
        std::condition_variable fcv;
        std::atomic<size_t> next_file{0};
        size_t consumed = 0;  // under fm: files the joiner has taken (workers stay at most `ahead` files in front of it)
        const size_t n_workers = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)usable_cpus() / 2, nf}));
        const size_t ahead = 4 * n_workers;
        std::vector<std::thread> workers;
        for (size_t wi = 0; wi < n_workers; wi++)
          workers.emplace_back([&] {
            std::string wid, ws;
            for (;;) {
              const size_t fi = next_file.fetch_add(1);
              if (fi >= nf) return;
              {
                std::unique_lock<std::mutex> l(fm);
                fcv.wait(l, [&] { return fi < consumed + ahead; });
              }
              std::unique_ptr<FileQuery> fq(new FileQuery());
              fq->q.packed = true;
              FastxReader r(files[fi]);
              while (r.next(&wid, &ws)) {
                if (fq->empty) {
                  fq->qid = o.use_filename ? trim_ext(files[fi]) : (!o.query_id.empty() ? o.query_id : wid);
                  fq->empty = false;
                  fq->q.pack_append(ws.data(), ws.size());
                } else {
                  fq->q.pack_append(ws.data(), ws.size());
                  fq->q.pack_append(nnn.data(), nnn.size());
                }
              }
              fq->done = true;
              std::lock_guard<std::mutex> l(fm);
              slots[fi] = std::move(fq);
              fcv.notify_all();
            }
          });
        for (size_t fi = 0; fi < nf; fi++) {
          if (verbose) info("reading sequence file: %s", files[fi].c_str());
          std::unique_ptr<FileQuery> fq;
          {
            std::unique_lock<std::mutex> l(fm);
            fcv.wait(l, [&] { return slots[fi] != nullptr; });
            fq = std::move(slots[fi]);
            consumed = fi + 1;
            fcv.notify_all();
          }
          if (fq->empty) { warn("no valid sequences in file: %s", files[fi].c_str()); continue; }
          b->packed = true;
          b->append_packed(fq->q.codes.data(), fq->q.n_bases, fq->q.exc.data(), fq->q.n_exc);
          b->id_buf.insert(b->id_buf.end(), fq->qid.begin(), fq->qid.end());
          b->id_offs.push_back(b->id_buf.size());
          b->offs.push_back(b->n_bases);
          id++;
          if (b->size() >= batch_reads.load() || b->bases() >= max_bases.load()) flush();
        }
        for (auto& t : workers) t.join();
      }
      for (const auto& file : o.whole_file ? std::vector<std::string>() : files) {
        if (verbose) info("reading sequence file: %s", file.c_str());
        flush();  // batches do not span input files on this path
        const uint64_t got = read_single_end(file, batch_reads.load(), max_bases.load(), [&](std::unique_ptr<Batch> nb) {
          nb->paired = false;
          nb->first_idx = id;
          id += nb->size();
          b = std::move(nb);
          flush();
        });
        if (got == 0) warn("no valid sequences in file: %s", file.c_str());
      }
    }
    flush();
    q_in.close();
    t_reader_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
  });


  if (verbose) info("loading database into GPU memory ...");
  kmcpg_db* db = nullptr;
  int32_t paged_passes = 0;
  if (o.gpu_ids.empty() && o.gpus > 1)
    for (int i = 0; i < o.gpus; i++) o.gpu_ids.push_back(i);
  if (!o.gpu_ids.empty()) {  // one process, several GPUs: blocks partitioned over the devices, hits merged on the host
    if (kmcpg_open_devices(db_dirs[0].c_str(), o.gpu_ids.data(), (int32_t)o.gpu_ids.size(), &db) != 0)
      die("open kmcp db: %s: %s", db_dirs[0].c_str(), kmcpg_last_error());
    if (verbose) info("  %zu GPUs, exchange of the hit lists: %s", o.gpu_ids.size(), kmcpg_exchange_info(db));
  } else {
    kmcpg_opts gopts{o.device, 0, 1, 0};
    int rc = o.gpu_passes >= 0 ? KMCPG_ENOMEM : kmcpg_open(db_dirs[0].c_str(), &gopts, &db);
    if (rc == KMCPG_ENOMEM) {
      // the index is larger than the GPU's memory (or --gpu-passes asks for it): one part of it resident at a time, every batch
      // searched against all parts in turn (the reference's counterpart: mmap / --low-mem, search.go:80)
      if (o.gpu_passes < 0) warn("%s", kmcpg_last_error());
      if (kmcpg_open_paged(db_dirs[0].c_str(), o.device, std::max(0, o.gpu_passes), &db) != 0) die("open kmcp db: %s: %s", db_dirs[0].c_str(), kmcpg_last_error());
      int32_t passes = 0;
      kmcpg_paged_info(db, &passes, nullptr);
      paged_passes = passes;
      if (passes > 1) {
        if (!o.batch_given) o.batch = 4 << 20;  // a batch costs passes - 1 uploads of index parts: large batches keep their share small
        warn("the index is searched in %d passes per batch of %d queries (one part resident in GPU memory at a time); more GPUs (--gpus) avoid this", passes, o.batch);
      }
    } else if (rc != 0) die("open kmcp db: %s: %s", db_dirs[0].c_str(), kmcpg_last_error());
  }
  kmcpg_info dbi;
  kmcpg_db_info(db, &dbi);
  if (dbi.minimizer && !dbi.syncmer)
    warn("this is a minimizer database: the reference publishes no result for minimizer sketches to check against, so this mode is "
         "verified against a restatement of bio/sketches only (DESIGN.md section 2)");
  if (verbose) {
    // Narrow blocks (rows of up to 64 bytes: what `kmcp index -j 32` makes of a small database) cost one memory request per (k-mer,
    // block) whatever their width; blocks that share NumSigs are laid side by side in GPU memory and served by ONE request.  Blocks
    // with a NumSigs of their own cannot be: say so once, with the remedy (profiles/r04_narrow_rows.txt: ~3x).
    std::set<uint64_t> sigs;
    int narrow = 0;
    for (int32_t b = 0; b < dbi.n_blocks; b++) {
      uint64_t ns = 0;
      uint32_t nc = 0, rb = 0, st = 0, cb = 0;
      int32_t loc = 0;
      if (kmcpg_block_info(db, (uint32_t)b, &ns, &nc, &rb, &st, &loc, &cb) == 0 && rb <= 64) {
        narrow++;
        sigs.insert(ns);
      }
    }
    if (narrow > 1 && sigs.size() > 1)
      info("  note: %zu distinct NumSigs over %d narrow blocks (rows <= 64 bytes): every k-mer costs %zu gathers; a database built with fewer, wider "
           "blocks (`kmcp index -b`) or with equal NumSigs (kmcpg_build_db uniform_sigs = 1) is searched ~3x faster",
           sigs.size(), narrow, sigs.size());
  }
  if (o.min_qcov <= dbi.fpr)  // search.go:405-409
    die("query coverage threshold (%f) should not be smaller than FPR of single bloom filter of index database (%f)", o.min_qcov, dbi.fpr);
  if (verbose) {
    info("database loaded: %s", o.db_dir.c_str());
    info("");
    info("-------------------- [main parameters] --------------------");
    info("  minimum    query length: %d", o.min_qlen);
    info("  minimum  matched k-mers: %d", o.min_kmers);
    info("  minimum  query coverage: %f", o.min_qcov);
    info("  minimum target coverage: %f", o.min_tcov);
    info("-------------------- [main parameters] --------------------");
    info("");
    info("searching ...");
  }
  // target names after mapping (util-db-search.go:317-332), resolved once per column
  std::vector<std::string> target(dbi.n_cols);
  for (uint32_t c = 0; c < dbi.n_cols; c++) {
    const char* nm = nullptr;
    kmcpg_col_info(db, c, &nm, nullptr, nullptr, nullptr);
    target[c] = nm;
    if (mapping || o.default_name_map) {
      auto it = name_map.find(target[c]);
      if (it != name_map.end()) target[c] = it->second;
      else if (o.default_name_map) {
        auto it2 = default_map.find(target[c]);
        if (it2 != default_map.end()) target[c] = it2->second;
      }
    }
  }

  kmcpg_params params{};
  params.min_qlen = o.min_qlen;
  params.min_matched = o.min_kmers;
  params.min_qcov = o.min_qcov;
  params.min_tcov = o.min_tcov;
  params.max_fpr = o.max_fpr;
  params.dedup_threshold = o.dedup;
  params.try_se = o.try_se;
  params.sort_by = sort_by;
  params.do_not_sort = o.do_not_sort;
  params.top_n_scores = o.top_scores;
  params.fpr_buf_size = paired ? 499 : 249;

  const auto t_search = std::chrono::steady_clock::now();
  Out out(o.out_file);
  if (!o.no_header) out.write("#query\tqLen\tqKmers\tFPR\thits\ttarget\tchunkIdx\tchunks\ttLen\tkSize\tmKmers\tqCov\ttCov\tjacc\tqueryIdx\n");

  uint64_t total = 0, matched = 0;
  // a batch also closes at 64 Mbases (long queries); paged indexes want the largest batches the host can hold
  // (a batch's device workspace is up to 24 B per base: the library says how many bases fit beside the resident index)
  {
    // (-g: whole genomes as queries, packed 4 bases to a byte on the host — 256 Mbases per batch, 64 assemblies of 4 Mbp: the GPU needs
    // ~2 ms for them while the readers need ~100, and the device workspace of a batch is 24 bytes per base — a gigabase batch made the
    // process allocate, and the driver reclaim after it, 25 GB for nothing: profiles/r06_cli_e2e.txt)
    size_t mb = paged_passes > 1 ? std::min<size_t>((size_t)o.batch * 512, (size_t)2 << 30) : (o.whole_file ? (size_t)256 << 20 : (size_t)64 << 20);
    uint64_t hint = 0;
    if (kmcpg_batch_hint(db, &hint) == 0 && hint > 0) mb = std::max<size_t>((size_t)1 << 20, std::min<size_t>(mb, (size_t)hint));
    max_bases.store(mb);
    batch_reads.store((size_t)o.batch);
    db_k.store(dbi.k);
    db_ready.store(true);
    { std::lock_guard<std::mutex> g(ready_mu); }
    ready_cv.notify_all();
  }

  int fmt_threads = 0;
  double t_fmt_busy = 0;  // summed over the formatter threads: time inside the parts
  double t_fmt_pool = 0, t_fmt_push = 0, t_fmt_wait = 0;  // of the writer loop: rows being formatted / waiting for the flusher / waiting for a searched batch
  double t_gpu = 0, t_fmt = 0, t_read_wait = 0;  // seconds spent inside libkmcpgpu / formatting+writing / waiting for input
  uint64_t sum_matches = 0, sum_check = 0;       // matches of the run and their order-independent checksum (log line below)
  // two searchers: libkmcpgpu serialises their GPU halves and runs the host half (thresholds, FPR, sorting) outside that lock,
  // so one batch is finalized while the next one's kernels run
  // (a paged index searches one batch at a time inside the library and wants the largest batches: one searcher, which joins the
  // batches the reader cut before the database was open — consecutive ones, so the order of the output is untouched)
  const int n_search = paged_passes > 1 ? 1 : 2;
  std::mutex t_mu;
  std::atomic<int> live{n_search};
  std::vector<std::thread> searchers;
  for (int si = 0; si < n_search; si++)
    searchers.emplace_back([&] {
      std::unique_ptr<Batch> b;
      double my_gpu = 0, my_wait = 0;
      uint64_t my_sum = 0, my_matches = 0;
      // Each searcher keeps `depth` batches in flight through kmcpg_submit / kmcpg_wait_pairs (round 6; one synchronous
      // kmcpg_search_batch_pairs call per batch before): a submit returns once the batch is staged, so the upload and the kernels of
      // the next batch queue up behind this one's instead of waiting for this thread to come back from the host half.
      const size_t depth = paged_passes > 1 ? 1 : 2;
      std::deque<std::pair<kmcpg_ticket*, std::unique_ptr<Batch>>> fl;
      auto search_sync = [&](Batch& bb) {  // the one-call form: it also halves a batch whose workspace does not fit (kmcp_gpu.h kmcpg_batch_hint)
        if (bb.packed) {  // (a rare path: the text again, the one-call form reads text)
          bb.seqs.resize((size_t)bb.n_bases + 16);
          if (kmcpg_unpack2(bb.codes.data(), bb.n_bases, bb.exc.data(), bb.n_exc, bb.seqs.data()) != 0) die("%s", kmcpg_last_error());
        }
        if (kmcpg_search_batch_pairs(db, bb.seqs.data(), bb.offs.data(), bb.paired ? bb.seqs2.data() : nullptr, bb.paired ? bb.offs2.data() : nullptr,
                                     (uint32_t)bb.size(), &params, &bb.res) != 0)
          die("%s", kmcpg_last_error());
      };
      auto publish = [&](std::unique_ptr<Batch> bb) {
        if (verbose) {  // order-independent checksum of the (query, column, mKmers) tuples: the same on 1, 2, 4, 8 GPUs
          const kmcpg_result_pairs& r = bb->res;
          for (uint32_t i = 0; i < r.n_reads; i++)
            for (uint64_t j = r.match_offs[i]; j < r.match_offs[i + 1]; j++)
              my_sum += tuple_mix(bb->first_idx + i, r.pairs[j].col, r.pairs[j].count);
          if (r.n_reads) my_matches += r.match_offs[r.n_reads];
        }
        q_out.push(std::move(bb));
      };
      auto finish_oldest = [&] {
        kmcpg_ticket* t = fl.front().first;
        std::unique_ptr<Batch> bb = std::move(fl.front().second);
        fl.pop_front();
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = kmcpg_wait_pairs(t, &bb->res);
        if (rc == KMCPG_ENOMEM) search_sync(*bb);  // (kmcpg_wait consumed the ticket; the batch's buffers are still ours)
        else if (rc != 0) die("%s", kmcpg_last_error());
        my_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        publish(std::move(bb));
      };
      for (;;) {
        const auto tw = std::chrono::steady_clock::now();
        // (with batches of its own in flight a searcher does not sleep on an empty input queue: it brings its oldest batch home first)
        if (!fl.empty() ? !q_in.try_pop(&b) : !q_in.pop(&b)) {
          if (fl.empty()) break;
          finish_oldest();
          continue;
        }
        if (paged_passes > 1 && !b->packed) {
          std::unique_ptr<Batch> nb;
          while (b->size() < batch_reads.load() && b->seqs.size() + b->seqs2.size() < max_bases.load() && q_in.try_pop(&nb)) b->append(*nb);
        }
        const auto t0 = std::chrono::steady_clock::now();
        my_wait += std::chrono::duration<double>(t0 - tw).count();
        kmcpg_ticket* t = nullptr;
        int rc;
        while ((rc = b->packed ? kmcpg_submit_packed(db, b->codes.data(), b->offs.data(), b->exc.data(), b->n_exc, (uint32_t)b->size(), &params, &t)
                               : kmcpg_submit(db, b->seqs.data(), b->offs.data(), b->paired ? b->seqs2.data() : nullptr, b->paired ? b->offs2.data() : nullptr,
                                              (uint32_t)b->size(), &params, &t)) == KMCPG_EBUSY && !fl.empty())
          finish_oldest();  // every lane of the handle is taken: one of this thread's own comes back first
        if (rc == KMCPG_EBUSY || rc == KMCPG_ENOMEM) {  // the lanes are all the other searcher's / a batch that must be halved: the one-call form
          search_sync(*b);
          my_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          publish(std::move(b));
          continue;
        }
        if (rc != 0) die("%s", kmcpg_last_error());
        my_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fl.emplace_back(t, std::move(b));
        if (fl.size() >= depth) finish_oldest();
      }
      {
        std::lock_guard<std::mutex> g(t_mu);
        sum_matches += my_matches;
        sum_check += my_sum;
        t_gpu += my_gpu;
        t_read_wait += my_wait;
      }
      if (live.fetch_sub(1) == 1) q_out.close();
    });

  // writer: rows exactly as search.go:517-575 / :458-512.  A batch is formatted by several threads (contiguous ranges of
  // queries, concatenated in order); with -o *.gz each range becomes its own gzip member, compressed in the same thread
  // (a multi-member .gz is what pgzip/gzip readers, `kmcp profile` included, accept).
  {
    // -j formatter threads; by default three quarters of the cores this process may use (the rest: two searchers, the reader's
    // parsers, the library's workers, the flusher) — on the 16-core grant of a GPU box 12 threads were the knee, profiles/r06_cli_e2e.txt
    const int nfmt = o.threads > 0 ? std::max(1, std::min(o.threads, 64)) : (int)std::max(4u, std::min(32u, usable_cpus() * 3 / 4));
    fmt_threads = nfmt;
    FormatPool pool(nfmt);
    // the formatted text of a batch goes to the file on a thread of its own, while the next batch is being formatted
    // One batch's text: the parts in order, each with the formatter that wrote it.  Text buffers go round — a match-heavy batch is hundreds
    // of megabytes of rows, fresh strings would be page-faulted in (and grown by doubling) for every batch — and they go back to the
    // THREAD that wrote them: a buffer another core filled last costs a cache-line transfer per line written (measured: formatting
    // took 2-4x the thread-seconds of the same loop on thread-owned buffers, profiles/r06_cli_e2e.txt).
    struct Text {
      std::vector<std::string> part;
      std::vector<RowFormatter*> owner;
    };
    Queue<std::unique_ptr<Text>> q_flush(4);
    std::thread flusher([&] {
      std::unique_ptr<Text> t;
      while (q_flush.pop(&t)) {
        for (size_t i = 0; i < t->part.size(); i++) {
          out.write_raw(t->part[i]);
          if (RowFormatter* F = t->owner[i]) F->give_back(std::move(t->part[i]));
        }
      }
    });
    std::unique_ptr<Batch> b, got;
    std::map<uint64_t, std::unique_ptr<Batch>> pending;  // batches that finished ahead of their turn
    uint64_t next_seq = 0;
    for (;;) {
      auto it = pending.find(next_seq);
      if (it != pending.end()) {
        b = std::move(it->second);
        pending.erase(it);
      } else {
        const auto tp0 = std::chrono::steady_clock::now();
        const bool more = q_out.pop(&got);
        t_fmt_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count();
        if (!more) break;
        if (got->seq != next_seq) {
          pending.emplace(got->seq, std::move(got));
          continue;
        }
        b = std::move(got);
      }
      next_seq += b->n_seq;
      const auto tf0 = std::chrono::steady_clock::now();
      const kmcpg_result_pairs& r = b->res;
      const uint32_t n = r.n_reads;
      // parts of about equal work: a query costs one unit, a row one more (reads of a family database carry hundreds of rows)
      const uint64_t rows = n ? r.match_offs[n] : 0;
      // (up to four parts per thread, taken in turn: a thread that is descheduled for a while holds up a small part, not an eighth of the batch)
      const int parts = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)nfmt * 4, std::max<uint64_t>((n + 2047) / 2048, rows / 16384)));
      std::vector<uint32_t> cut((size_t)parts + 1, n);
      cut[0] = 0;
      for (int pi = 1; pi < parts; pi++) {
        const uint64_t want = (rows + n) * (uint64_t)pi / (uint64_t)parts;
        uint32_t a = cut[(size_t)pi - 1], z = n;  // first query i with match_offs[i] + i >= want
        while (a < z) {
          const uint32_t mid = a + (z - a) / 2;
          if (r.match_offs[mid] + mid < want) a = mid + 1; else z = mid;
        }
        cut[(size_t)pi] = a;
      }
      std::unique_ptr<Text> chunk_p(new Text());
      chunk_p->part.resize((size_t)parts);
      chunk_p->owner.assign((size_t)parts, nullptr);
      std::vector<std::string>& chunk = chunk_p->part;
      std::vector<RowFormatter*>& chunk_owner = chunk_p->owner;
      std::vector<uint64_t> part_matched((size_t)parts, 0);
      std::vector<double> part_busy((size_t)parts, 0);
      const std::function<void(int, RowFormatter&)> work = [&](int pi, RowFormatter& F) {
        const auto tb0 = std::chrono::steady_clock::now();
        std::string& buf = chunk[(size_t)pi];
        F.take(buf);  // one of this thread's own buffers, if one has come back from the flusher
        chunk_owner[(size_t)pi] = &F;
        const uint32_t lo = cut[(size_t)pi], hi = cut[(size_t)pi + 1];
        buf.reserve((size_t)(r.match_offs[hi] - r.match_offs[lo]) * 112 + (size_t)(hi - lo) * (o.keep_unmatched ? 64 : 8) + 256);
        for (uint32_t i = lo; i < hi; i++) {
          const uint64_t qidx = b->first_idx + i;
          const uint64_t m0 = r.match_offs[i], m1 = r.match_offs[i + 1];
          if (m0 == m1) {
            if (o.keep_unmatched) F.unmatched(buf, b->id(i), r.qlen[i], r.qkmers[i], r.ksize[i], qidx);
            continue;
          }
          part_matched[(size_t)pi]++;
          // the query's Match records (float64 qCov / tCov / jacc, the FPR column, tLen ...) from its pairs, into a scratch array that
          // stays in this thread's cache: the batch's records never exist as a whole (1.5 GB per 131 072 reads of a family database)
          if (F.scratch.size() < m1 - m0) F.scratch.resize((size_t)(m1 - m0));
          if (kmcpg_expand_pairs(db, r.qkmers[i], r.pairs + m0, m1 - m0, F.scratch.data()) != 0) die("%s", kmcpg_last_error());
          F.rows(buf, b->id(i), r.qlen[i], r.qkmers[i], F.scratch.data(), m1 - m0, target, r.ksize[i], qidx);
        }
        if (out.gz()) buf = gzip_member(buf);
        part_busy[(size_t)pi] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
      };
      const auto tq0 = std::chrono::steady_clock::now();
      pool.run(parts, work);
      const auto tq1 = std::chrono::steady_clock::now();
      for (int pi = 0; pi < parts; pi++) {
        matched += part_matched[(size_t)pi];
        t_fmt_busy += part_busy[(size_t)pi];
      }
      q_flush.push(std::move(chunk_p));
      t_fmt_pool += std::chrono::duration<double>(tq1 - tq0).count();
      t_fmt_push += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq1).count();
      total += n;
      kmcpg_result_pairs_free(&b->res);
      // the batch's vectors go back to the reader (fastx_reader.hpp ChunkPool)
      ChunkPool::get().give(b->id_buf, b->id_offs, b->seqs, b->offs);
      if (b->paired) {
        std::vector<char> no_ids;
        std::vector<uint64_t> no_offs{0};
        ChunkPool::get().give(no_ids, no_offs, b->seqs2, b->offs2);
      }
      t_fmt += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
      if (verbose && !o.quiet) {
        double min = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_search).count() / 60.0;
        fprintf(stderr, "processed queries: %llu, speed: %.3f million queries per minute\r", (unsigned long long)total, total / 1e6 / min);
      }
    }
    q_flush.close();
    flusher.join();
  }
  reader.join();
  for (auto& t : searchers) t.join();

  if (verbose) {
    fprintf(stderr, "\n");
    double min = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_search).count() / 60.0;
    info("");
    info("processed queries: %llu, speed: %.3f million queries per minute", (unsigned long long)total, total / 1e6 / min);
    info("%.4f%% (%llu/%llu) queries matched", total ? (double)matched / (double)total * 100 : NAN, (unsigned long long)matched, (unsigned long long)total);
    info("done searching (pipeline: %.3f s in the GPU library, %.3f s formatting/writing, %.3f s waiting for the reader; reader: %.3f s parsing, %.3f s "
         "blocked; %.3f s before the search started)",
         t_gpu, t_fmt, t_read_wait, t_reader_total - t_reader_blocked, t_reader_blocked, std::chrono::duration<double>(t_search - t_start).count());
    info("writer loop: %.3f s formatting rows on %d threads (%.3f thread-seconds inside the parts), %.3f s waiting for the flusher, %.3f s waiting for searched batches",
         t_fmt_pool, fmt_threads, t_fmt_busy, t_fmt_push, t_fmt_wait);
    info("matches: %llu, checksum %016llx (order-independent over (queryIdx, column, mKmers): the same on any number of GPUs)", (unsigned long long)sum_matches,
         (unsigned long long)sum_check);
    if (o.out_file != "-") info("search results saved to: %s", o.out_file.c_str());
  }
  // trailer read by `kmcp profile` (profile.go:1945-1951)
  char tr[256];
  int n = snprintf(tr, sizeof tr, "# input queries: %llu\n# matched queries: %llu\n", (unsigned long long)total, (unsigned long long)matched);
  std::string trailer(tr, (size_t)n);
  if (total) n = snprintf(tr, sizeof tr, "# matched percentage: %.4f%%\n", (double)matched / (double)total * 100);
  else n = snprintf(tr, sizeof tr, "# matched percentage: NaN%%\n");
  trailer.append(tr, (size_t)n);
  out.write(trailer);
  out.close();
  // Everything the user asked for is on disk.  Giving back pinned staging buffers, streams and a resident index one by one takes
  // ~0.1 s and the runtime's own exit handlers as long again (profiles/r06_cli_e2e.txt) — a short-lived process leaves that to the
  // kernel driver, which reclaims a dead process's GPU memory anyway (the Go reference exits the same way: search.go:1027).
  // KMCP_SEARCH_FULL_TEARDOWN=1 closes the handle and returns through the runtime's handlers (leak checks, sanitizers).
  // (The handle itself IS closed: device memory a process leaves behind is reclaimed by the driver while the NEXT process is starting —
  // three back-to-back runs that each left 25 GB took 0.69, 0.90, 1.74 s.)
  const bool full_teardown = getenv("KMCP_SEARCH_FULL_TEARDOWN") != nullptr;
  if (kmcpg_close(db) != 0) die("%s", kmcpg_last_error());
  if (verbose) {
    info("");
    info("elapsed time: %.3fs", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    info("");
  }
  if (g_log) fclose(g_log);
  fflush(stdout);
  fflush(stderr);
  if (!full_teardown) _exit(0);
  return 0;
}
