// kmcp-merge: the fan-in of search results of the same reads against several databases (or several nodes' database
// partitions) — host-side mirror of `kmcp merge` (kmcp/cmd/merge.go:40-420, parser :459-573).
//
//   kmcp-merge [-o out.tsv[.gz]] [-s qcov|tcov|jacc] [-f 15] [-n 5] [-H] [-i list.txt] a.tsv[.gz] b.tsv[.gz] ...
//
// Semantics restated from the reference:
//   * every input is a `kmcp search` TSV whose rows are grouped by queryIdx (field -f) in ascending order; the inputs are
//     merged k-way on queryIdx; the rows of one query from all inputs are concatenated, sorted by the chosen score column
//     (qCov/tCov/jacc = fields f-3/f-2/f-1, parsed from the text) in descending order, and their `hits` field (-n) is
//     rewritten to the new row count (merge.go:190-262);
//   * the query IDs (field 1) of one queryIdx must agree across inputs (merge.go:244-247);
//   * "# input queries:" must agree across inputs that carry it (merge.go:283-341); the trailer is re-emitted with the merged
//     counts (merge.go:386-388).  `matched` counts queryIdx groups — including rows kept by `search -K` — and is 1 for inputs
//     without any row, exactly as the reference's loop does (merge.go:200-262);
//   * one input: copied through unchanged (merge.go:117-139).
// Where the reference is nondeterministic (its parallel unstable quicksort + heap order among equal keys) this tool is
// deterministic: equal scores keep (input file order, row order).
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <set>
#include <string>
#include <vector>

namespace {

[[noreturn]] void die(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "[ERRO] ");
  vfprintf(stderr, fmt, ap);
  fputc('\n', stderr);
  va_end(ap);
  exit(255);  // checkError: os.Exit(-1)
}

// gz-or-plain line reader (zlib reads plain files transparently)
struct LineReader {
  gzFile f = nullptr;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  explicit LineReader(const std::string& path) : buf(1 << 20) {
    f = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
    if (!f) die("%s: %s", path.c_str(), strerror(errno));
    gzbuffer(f, 1 << 18);
  }
  ~LineReader() {
    if (f) gzclose(f);
  }
  bool fill() {
    if (eof) return false;
    if (pos > 0) {
      memmove(buf.data(), buf.data() + pos, end - pos);
      end -= pos;
      pos = 0;
    }
    if (end == buf.size()) buf.resize(buf.size() * 2);
    int n = gzread(f, buf.data() + end, (unsigned)(buf.size() - end));
    if (n < 0) die("read error");
    if (n == 0) {
      eof = true;
      return false;
    }
    end += (size_t)n;
    return true;
  }
  bool next(std::string& line) {
    for (;;) {
      char* nl = (char*)memchr(buf.data() + pos, '\n', end - pos);
      if (nl) {
        size_t len = (size_t)(nl - (buf.data() + pos));
        line.assign(buf.data() + pos, len);
        pos += len + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();  // bufio.ScanLines drops a trailing \r
        return true;
      }
      if (!fill()) {
        if (pos < end) {
          line.assign(buf.data() + pos, end - pos);
          pos = end;
          if (!line.empty() && line.back() == '\r') line.pop_back();
          return true;
        }
        return false;
      }
    }
  }
};

// An error found by a parser thread: reported by the merge when it gets there (in the order the reference, which parses as it
// merges, would meet it), not whenever the thread happens to read that far ahead.
struct ParseError {
  std::string msg;
};
[[noreturn]] void fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw ParseError{buf};
}

// One result row, located inside the text of its batch.  The reference splits a line into numFields fields, the last one holding
// the rest of the line (util.go:257-277), rewrites the hits field and joins the fields with tabs again — i.e. it prints the line
// with one field replaced; only the fields it looks at are located here (no string per field, no string per row).
struct Row {
  uint64_t qidx;
  double score;
  uint32_t lo, len;           // the line (without its end-of-line) inside Batch::text
  uint32_t id_len;            // field 1: the query ID
  uint32_t hits_lo, hits_hi;  // the hits field (-n), relative to the line
};

// A block of an input as read (whole lines), and its rows
struct Batch {
  std::string text;
  std::vector<Row> rows;
};

// merge.go:459-573.  The input is read in blocks of whole lines and parsed in place on a thread of its own; the merge takes
// the batches in order.  (Runs of rows with one queryIdx — the reference's parser hands those out — are formed by the
// consumer, across batch boundaries where they fall inside a run.)
struct ResultParser {
  gzFile f = nullptr;
  std::string path;
  int num_fields, score_field, hits_field;
  long long input_queries = 0;  // "# input queries: N" (0 = absent, e.g. an old kmcp version)
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::shared_ptr<Batch>> q;
  bool closed = false;
  // batches go round (a fresh 4 MB text and 2 MB of rows per block would be zero-filled and page-faulted in every time)
  std::mutex pool_mu;
  std::vector<Batch*> pool;
  std::shared_ptr<Batch> new_batch() {
    Batch* b = nullptr;
    {
      std::lock_guard<std::mutex> g(pool_mu);
      if (!pool.empty()) {
        b = pool.back();
        pool.pop_back();
      }
    }
    if (!b) b = new Batch();
    b->rows.clear();
    return std::shared_ptr<Batch>(b, [this](Batch* x) {
      std::lock_guard<std::mutex> g(pool_mu);
      if (pool.size() < 16) pool.push_back(x);
      else delete x;
    });
  }

  ResultParser(const std::string& path_, int nf, int sf, int hf) : path(path_), num_fields(nf), score_field(sf), hits_field(hf) {
    f = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");  // zlib reads plain files transparently
    if (!f) die("%s: %s", path.c_str(), strerror(errno));
    gzbuffer(f, 1 << 18);
    th = std::thread([this] { run(); });
  }
  ~ResultParser() {
    if (th.joinable()) th.join();
    if (f) gzclose(f);
    cur.reset();
    q.clear();
    for (Batch* x : pool) delete x;
  }

  void comment(const char* p, size_t n) {
    // ^# ([\w ]+): (.+)
    if (n > 2 && p[1] == ' ') {
      size_t i = 2;
      while (i < n && (isalnum((unsigned char)p[i]) || p[i] == '_' || p[i] == ' ')) i++;
      if (i > 2 && i + 2 < n && p[i] == ':' && p[i + 1] == ' ') {
        // greedy [\w ]+ then ": ": the regexp backtracks to the LAST position where ": " follows a [\w ]+ prefix; a key made
        // of word characters and blanks cannot contain ':', so the first ':' is the only candidate
        if (i - 2 == 13 && memcmp(p + 2, "input queries", 13) == 0) {
          const std::string val(p + i + 2, n - i - 2);
          char* e = nullptr;
          errno = 0;
          long long v = strtoll(val.c_str(), &e, 10);
          if (errno || e == val.c_str() || *e) fail("invalid value of input queries");
          input_queries = v;
        }
      }
    }
  }

  // one line [p, p+n) of the batch (no end-of-line characters); appends a Row unless it is empty or a comment
  void parse_line(Batch& b, const char* base, size_t lo, size_t n) {
    const char* p = base + lo;
    if (n == 0) return;
    if (p[0] == '#') {
      comment(p, n);
      return;
    }
    Row r{};
    r.lo = (uint32_t)lo;
    r.len = (uint32_t)n;
    r.id_len = (uint32_t)n;
    size_t s = 0, score_lo = 0, score_hi = 0;
    int nf = 1;
    // the first num_fields - 1 tabs delimit the fields; the last field is the rest
    auto tab_at = [&](size_t i) {
      if (nf == 1) r.id_len = (uint32_t)i;
      if (nf == hits_field) { r.hits_lo = (uint32_t)s; r.hits_hi = (uint32_t)i; }
      if (nf == score_field) { score_lo = s; score_hi = i; }
      s = i + 1;
      nf++;
    };
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i tabs = _mm_set1_epi8('\t');
    for (; i + 16 <= n && nf < num_fields; i += 16) {
      unsigned mask = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + i)), tabs));
      while (mask && nf < num_fields) {
        tab_at(i + (size_t)__builtin_ctz(mask));
        mask &= mask - 1;
      }
    }
#endif
    for (; i < n && nf < num_fields; i++)
      if (p[i] == '\t') tab_at(i);
    if (nf < num_fields) fail("number of fields (%d) < query index field (%d)", nf, num_fields);
    if (hits_field == num_fields) { r.hits_lo = (uint32_t)s; r.hits_hi = (uint32_t)n; }
    // queryIdx: the last field, digits only
    {
      const size_t m2 = n - s;
      uint64_t v = 0;
      bool ok = m2 > 0 && m2 <= 19;  // 19 digits cannot overflow
      for (size_t i = 0; ok && i < m2; i++) {
        const unsigned d = (unsigned)(p[s + i] - '0');
        ok = d <= 9;
        v = v * 10 + d;
      }
      if (!ok) {  // not plain digits (or 20 of them): the checked path
        char tmp[32];
        if (m2 == 0 || m2 >= sizeof tmp || !isdigit((unsigned char)p[s])) fail("invalid query index at field %d: %s", num_fields, std::string(p + s, m2).c_str());
        memcpy(tmp, p + s, m2);
        tmp[m2] = 0;
        char* e = nullptr;
        errno = 0;
        v = strtoull(tmp, &e, 10);
        if (errno || *e) fail("invalid query index at field %d: %s", num_fields, tmp);
      }
      r.qidx = v;
    }
    {
      // the score.  `kmcp search` prints it as digits '.' digits: with at most 15 significant digits and at most 22 decimals the
      // integer and the power of ten are exact doubles and their quotient is the correctly rounded value — what strtod returns;
      // anything else goes to strtod.
      const size_t m2 = score_hi - score_lo;
      const char* q = p + score_lo;
      uint64_t mant = 0;
      int digits = 0, decimals = -1;
      bool fast = m2 > 0;
      for (size_t i = 0; fast && i < m2; i++) {
        const char c = q[i];
        if (c == '.' && decimals < 0) decimals = 0;
        else if (c >= '0' && c <= '9') {
          mant = mant * 10 + (uint64_t)(c - '0');
          digits++;
          if (decimals >= 0) decimals++;
          fast = digits <= 15;
        } else fast = false;
      }
      static const double P10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
      if (fast && digits > 0 && decimals <= 22 && q[m2 - 1] != '.' && q[0] != '.') {
        r.score = decimals > 0 ? (double)mant / P10[decimals] : (double)mant;
      } else {
        const std::string sc(q, m2);
        char* e = nullptr;
        r.score = strtod(sc.c_str(), &e);
        if (m2 == 0 || e == sc.c_str() || *e) fail("failed to parse score: %s", sc.c_str());
      }
    }
    b.rows.push_back(r);
  }

  void run() {
    size_t BLOCK = 4u << 20;
    if (const char* e = getenv("KMCP_MERGE_BLOCK")) BLOCK = (size_t)std::max(16L, atol(e));  // tests: lines and runs across block boundaries
    std::string carry;  // the incomplete last line of the previous block
    bool eof = false;
    while (!eof) {
      std::shared_ptr<Batch> b = new_batch();
      b->text.resize(carry.size() + BLOCK);
      memcpy(&b->text[0], carry.data(), carry.size());
      size_t have = carry.size();
      carry.clear();
      while (have < b->text.size()) {
        const int n = gzread(f, &b->text[have], (unsigned)(b->text.size() - have));
        if (n < 0) die("read error");
        if (n == 0) {
          eof = true;
          break;
        }
        have += (size_t)n;
      }
      size_t end = have;
      if (!eof) {  // keep whole lines; a line longer than the block grows the next one
        const void* nl = have ? memrchr(b->text.data(), '\n', have) : nullptr;
        end = nl ? (size_t)((const char*)nl - b->text.data()) + 1 : 0;
        carry.assign(b->text.data() + end, have - end);
      }
      b->text.resize(end);
      b->rows.reserve(end / 80 + 16);
      const char* base = b->text.data();
      size_t lo = 0;
      try {
        while (lo < end) {
          const char* nl = (const char*)memchr(base + lo, '\n', end - lo);
          size_t hi = nl ? (size_t)(nl - base) : end;
          size_t n = hi - lo;
          if (n && base[lo + n - 1] == '\r') n--;  // bufio.ScanLines drops a trailing \r
          parse_line(*b, base, lo, n);
          lo = hi + 1;
        }
      } catch (const ParseError& e) {
        error = e.msg;  // the rows before the bad line are still merged; the merge stops when it needs more of this input
        eof = true;
      }
      if (!b->rows.empty()) push(std::move(b));
    }
    std::lock_guard<std::mutex> l(m);
    closed = true;
    cv.notify_all();
  }
  std::string error;  // set before `closed`
  void push(std::shared_ptr<Batch> b) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return q.size() < 6; });
    q.push_back(std::move(b));
    cv.notify_all();
  }

  // ---- consumer side ----
  std::shared_ptr<Batch> cur;
  size_t ri = 0;
  // makes cur/ri point at the next unread row; false at the end of the input
  bool ready() {
    while (!cur || ri == cur->rows.size()) {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return !q.empty() || closed; });
      if (q.empty()) {
        if (!error.empty()) die("%s", error.c_str());
        cur.reset();
        return false;
      }
      cur = std::move(q.front());
      q.pop_front();
      ri = 0;
      cv.notify_all();
    }
    return true;
  }
  // valid once ready() has returned false (the thread has seen the whole input, trailer included)
  long long queries() {
    if (th.joinable()) th.join();
    return input_queries;
  }
};

// a row of the run being merged: where its text is
struct RowRef {
  const char* line;
  const Row* r;
};

// The output: plain, or gzip written as a sequence of members (1 MB of text each) that a few threads compress side by side —
// what the reference gets from pgzip; gzip readers, `kmcp profile` included, read the members as one stream.
struct Out {
  FILE* fp = nullptr;
  bool gz = false;
  int level = 6;
  std::string buf;
  struct Job {
    std::string in, out;
    bool done = false;
  };
  std::deque<std::shared_ptr<Job>> order_, todo_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_;
  bool stop_ = false;

  Out(const std::string& path, int level_) {
    gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    level = level_ < 0 ? 6 : std::min(level_, 9);
    fp = path == "-" ? stdout : fopen(path.c_str(), "wb");
    if (!fp) die("%s: %s", path.c_str(), strerror(errno));
    buf.reserve(1 << 20);
    if (gz) {
      const unsigned n = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
      for (unsigned i = 0; i < n; i++) workers_.emplace_back([this] { work(); });
    }
  }
  ~Out() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
      cv_.notify_all();
    }
    for (auto& t : workers_) t.join();
  }
  static void gzip_member(const std::string& in, int level, std::string* out) {
    z_stream z{};
    if (deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("zlib: deflateInit2 failed");
    out->resize(deflateBound(&z, (uLong)in.size()) + 64);
    z.next_in = (Bytef*)in.data();
    z.avail_in = (uInt)in.size();
    z.next_out = (Bytef*)&(*out)[0];
    z.avail_out = (uInt)out->size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("zlib: deflate failed");
    out->resize(z.total_out);
    deflateEnd(&z);
  }
  void work() {
    for (;;) {
      std::shared_ptr<Job> j;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || !todo_.empty(); });
        if (todo_.empty()) return;
        j = std::move(todo_.front());
        todo_.pop_front();
      }
      gzip_member(j->in, level, &j->out);
      std::lock_guard<std::mutex> l(m_);
      j->done = true;
      cv_.notify_all();
    }
  }
  void put(const char* p, size_t n) {
    if (n && fwrite(p, 1, n, fp) != n) die("write error");
  }
  // writes the members that are ready, in order; with `all` waits for every one of them
  void drain(bool all) {
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      if (order_.empty()) return;
      if (!order_.front()->done) {
        if (!all && order_.size() < 4 * workers_.size()) return;
        cv_.wait(l, [&] { return order_.front()->done; });
      }
      std::shared_ptr<Job> j = std::move(order_.front());
      order_.pop_front();
      l.unlock();
      put(j->out.data(), j->out.size());
      l.lock();
    }
  }
  void flush() {
    if (buf.empty()) return;
    if (gz) {
      std::shared_ptr<Job> j(new Job());
      members_++;
      j->in.swap(buf);
      buf.reserve(1 << 20);
      {
        std::lock_guard<std::mutex> l(m_);
        order_.push_back(j);
        todo_.push_back(j);
        cv_.notify_all();
      }
      drain(false);
    } else {
      put(buf.data(), buf.size());
      buf.clear();
    }
  }
  void write(const char* p, size_t n) {
    buf.append(p, n);
    if (buf.size() > (1u << 20) - 4096) flush();
  }
  void write(const std::string& s) { write(s.data(), s.size()); }
  // a result line with its hits field [lo, hi) replaced, and a newline
  void write_row(const char* line, size_t lo, size_t hi, size_t len, const char* hits, size_t hn) {
    const size_t at = buf.size(), n = lo + hn + (len - hi) + 1;
    buf.resize(at + n);
    char* d = &buf[at];
    memcpy(d, line, lo);
    memcpy(d + lo, hits, hn);
    memcpy(d + lo + hn, line + hi, len - hi);
    d[n - 1] = '\n';
    if (buf.size() > (1u << 20) - 4096) flush();
  }
  void write(const char* s) { write(s, strlen(s)); }
  void close() {
    flush();
    if (gz) {
      drain(true);
      if (members_ == 0) {  // an empty .gz is still a gzip file
        std::string e;
        gzip_member(std::string(), level, &e);
        put(e.data(), e.size());
      }
    }
    if (fp != stdout) {
      if (fclose(fp) != 0) die("write error");
    } else fflush(fp);
    fp = nullptr;
  }
  size_t members_ = 0;
};

void usage() {
  fputs(
      "Merge search results from multiple databases\n\n"
      "Usage:\n  kmcp-merge [-o read.tsv.gz] [<search results> ...]\n\n"
      "Flags:\n"
      "  -f, --field-queryIdx int   Field of queryIdx. (default 15)\n"
      "  -n, --field-hits int       Field of hits. (default 5)\n"
      "  -H, --no-header-row        Do not print header row.\n"
      "  -o, --out-file string      Out file, supports a \".gz\" suffix (\"-\" for stdout). (default \"-\")\n"
      "  -s, --sort-by string       Sort hits by \"qcov\", \"tcov\" or \"jacc\" (Jaccard Index). (default \"qcov\")\n"
      "  -i, --infile-list string   File of input files list (one file per line).\n"
      "      --compression-level int  gzip level of the output (default 6)\n",
      stderr);
}

}  // namespace

int main(int argc, char** argv) {
  std::string out_file = "-", sort_by = "qcov", infile_list;
  int f_qidx = 15, f_hits = 5, level = -1;
  bool no_header = false;
  std::vector<std::string> files;
  auto need = [&](int& i) -> const char* {
    if (i + 1 >= argc) die("flag needs an argument: %s", argv[i]);
    return argv[++i];
  };
  auto pos_int = [&](const char* flag, const char* v) {
    char* e = nullptr;
    long x = strtol(v, &e, 10);
    if (*e || e == v) die("invalid argument \"%s\" for \"%s\" flag", v, flag);
    if (x <= 0) die("value of flag --%s should be positive: %ld", flag, x);
    return (int)x;
  };
  // `kmcp-merge merge ...` = `kmcp merge ...`: cobra's sub-command word, accepted (only) as the first argument
  for (int i = (argc > 1 && strcmp(argv[1], "merge") == 0) ? 2 : 1; i < argc; i++) {
    std::string a = argv[i];
    if (a == "-o" || a == "--out-file") out_file = need(i);
    else if (a == "-s" || a == "--sort-by") sort_by = need(i);
    else if (a == "-f" || a == "--field-queryIdx") f_qidx = pos_int("field-queryIdx", need(i));
    else if (a == "-n" || a == "--field-hits") f_hits = pos_int("field-hits", need(i));
    else if (a == "-H" || a == "--no-header-row") no_header = true;
    else if (a == "-i" || a == "--infile-list") infile_list = need(i);
    else if (a == "--compression-level") level = atoi(need(i));
    else if (a == "-j" || a == "--threads" || a == "--log") need(i);  // accepted for command-line compatibility
    else if (a == "-q" || a == "--quiet") {
    } else if (a == "-h" || a == "--help") {
      usage();
      return 0;
    } else if (a.size() > 1 && a[0] == '-') die("unknown flag: %s", a.c_str());
    else files.push_back(a);
  }
  if (sort_by != "qcov" && sort_by != "tcov" && sort_by != "jacc")
    die("invalid value for flag -s/--sort-by: %s. Available: qcov/tsov/jacc", sort_by.c_str());
  int score_field = f_qidx - (sort_by == "qcov" ? 3 : sort_by == "tcov" ? 2 : 1);
  if (score_field < 1 || f_hits > f_qidx) die("fields out of range: queryIdx %d, hits %d", f_qidx, f_hits);
  if (!infile_list.empty()) {
    std::ifstream fh(infile_list);
    if (!fh) die("%s: %s", infile_list.c_str(), strerror(errno));
    std::string l;
    while (std::getline(fh, l)) {
      while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
      if (!l.empty()) files.push_back(l);
    }
  }
  if (files.empty()) files.push_back("-");

  if (files.size() < 2) {  // merge.go:117-139: copy through
    fprintf(stderr, "[WARN] only one file given, we just copy and write the original data\n");
    LineReader in(files[0]);
    Out out(out_file, level);
    for (;;) {
      if (in.pos < in.end) {
        out.write(std::string(in.buf.data() + in.pos, in.end - in.pos));
        in.pos = in.end;
      }
      if (!in.fill()) break;
    }
    out.close();
    return 0;
  }
  std::set<std::string> seen;
  for (auto& f : files) {
    if (!seen.insert(f).second) die("duplicated file: %s", f.c_str());
    if (f == out_file) die("input and output file should not be the same: %s", f.c_str());
  }

  Out out(out_file, level);
  if (!no_header) out.write("#query\tqLen\tqKmers\tFPR\thits\ttarget\tchunkIdx\tchunks\ttLen\tkSize\tmKmers\tqCov\ttCov\tjacc\tqueryIdx\n");

  std::vector<std::unique_ptr<ResultParser>> parsers;
  for (auto& f : files) parsers.emplace_back(new ResultParser(f, f_qidx, score_field, f_hits));
  using Key = std::pair<uint64_t, size_t>;  // (queryIdx, input index): equal queryIdx pop in input order
  std::priority_queue<Key, std::vector<Key>, std::greater<Key>> heap;
  for (size_t i = 0; i < files.size(); i++)
    if (parsers[i]->ready()) heap.push({parsers[i]->cur->rows[parsers[i]->ri].qidx, i});

  long long matched = 0;
  std::vector<RowRef> rows;                   // the rows of the queryIdx being merged, from all inputs
  std::vector<std::shared_ptr<Batch>> held;   // batches those rows point into that their parser has moved on from
  std::vector<uint32_t> order;
  auto emit = [&]() {
    order.resize(rows.size());
    for (size_t i = 0; i < rows.size(); i++) order[i] = (uint32_t)i;
    if (rows.size() > 16) {
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rows[a].r->score > rows[b].r->score; });
    } else {  // the usual handful of rows: a stable insertion sort
      for (size_t i = 1; i < rows.size(); i++) {
        const uint32_t v = order[i];
        const double sv = rows[v].r->score;
        size_t j = i;
        for (; j > 0 && rows[order[j - 1]].r->score < sv; j--) order[j] = order[j - 1];
        order[j] = v;
      }
    }
    char hits[24];
    const int hn = snprintf(hits, sizeof hits, "%zu", rows.size());
    for (uint32_t oi : order) {
      const RowRef& r = rows[oi];
      out.write_row(r.line, r.r->hits_lo, r.r->hits_hi, r.r->len, hits, (size_t)hn);
    }
    rows.clear();
    held.clear();
  };
  bool first = true;
  uint64_t pre_idx = 0;
  const char* pre_id = nullptr;  // the query ID of the run being merged: inside one of its rows
  uint32_t pre_id_len = 0;
  while (!heap.empty()) {
    auto [qidx, i] = heap.top();
    heap.pop();
    ResultParser& P = *parsers[i];
    // the run of rows with this queryIdx in input i (it may continue in the input's next batches)
    const Row& r0 = P.cur->rows[P.ri];
    const char* id0 = P.cur->text.data() + r0.lo;
    if (first) {
      first = false;
      pre_idx = qidx;
      pre_id = id0;
      pre_id_len = r0.id_len;
    } else if (qidx != pre_idx) {
      matched++;
      emit();
      pre_idx = qidx;
      pre_id = id0;
      pre_id_len = r0.id_len;
    } else if (pre_id_len != r0.id_len || memcmp(pre_id, id0, r0.id_len) != 0) {
      die("[queryIdx: %" PRIu64 "] unmatched sequence Ids detected: idx '%s' != '%s'. please make sure the search results coming from same query files",
          qidx, std::string(id0, r0.id_len).c_str(), std::string(pre_id, pre_id_len).c_str());
    }
    bool more = true;
    while (more && P.cur->rows[P.ri].qidx == qidx) {
      const char* base = P.cur->text.data();
      const std::vector<Row>& br = P.cur->rows;
      while (P.ri < br.size() && br[P.ri].qidx == qidx) {
        rows.push_back(RowRef{base + br[P.ri].lo, &br[P.ri]});
        P.ri++;
      }
      if (P.ri == br.size()) held.push_back(P.cur);  // the parser moves on; the rows collected from this batch are not out yet
      more = P.ready();
    }
    if (more) heap.push({P.cur->rows[P.ri].qidx, i});
  }
  matched++;  // the last group — also when there was none (merge.go:243)
  emit();

  // merge.go:283-341 (every parser has reached its end, so every trailer has been seen)
  long long first_n = parsers[0]->queries(), total = first_n;
  for (size_t i = 1; i < parsers.size(); i++) {
    long long n = parsers[i]->queries();
    if (first_n == 0) {
      total = n;
      continue;
    }
    if (n == 0) continue;
    if (n != first_n)
      die("different numbers of queries in %s (%lld) and %s (%lld), please make sure they come from the same input query", files[0].c_str(), first_n,
          files[i - 1].c_str(), n);  // the reference indexes files[i] of a slice that starts at the second file
    total = n;
  }
  char tail[256];
  snprintf(tail, sizeof tail, "# input queries: %lld\n# matched queries: %lld\n", total, matched);
  out.write(tail);
  if (total == 0) snprintf(tail, sizeof tail, "# matched percentage: %s%%\n", matched ? "+Inf" : "NaN");
  else snprintf(tail, sizeof tail, "# matched percentage: %.4f%%\n", (double)matched / (double)total * 100);
  out.write(tail);
  out.close();
  return 0;
}
