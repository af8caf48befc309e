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

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <queue>
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

struct Row {
  std::vector<std::string> f;  // numFields fields, the last one holds the rest of the line (util.go:257-277)
  double score;
};

struct Group {
  uint64_t qidx = 0;
  std::string qid;
  std::vector<Row> rows;
};

// merge.go:459-573
struct ResultParser {
  LineReader in;
  int num_fields, score_field;
  bool have_pending = false, done = false;
  Row pending;
  uint64_t pending_idx = 0;
  long long input_queries = 0;  // "# input queries: N" (0 = absent, e.g. an old kmcp version)
  std::string line;
  ResultParser(const std::string& path, int nf, int sf) : in(path), num_fields(nf), score_field(sf) {}

  bool parse_row(Row& r, uint64_t& idx) {
    while (in.next(line)) {
      if (line.empty()) continue;
      if (line[0] == '#') {
        // ^# ([\w ]+): (.+)
        if (line.size() > 2 && line[1] == ' ') {
          size_t i = 2;
          while (i < line.size() && (isalnum((unsigned char)line[i]) || line[i] == '_' || line[i] == ' ')) i++;
          if (i > 2 && i + 2 < line.size() && line[i] == ':' && line[i + 1] == ' ') {
            // greedy [\w ]+ then ": ": the regexp backtracks to the LAST position where ": " follows a [\w ]+ prefix; a key made
            // of word characters and blanks cannot contain ':', so the first ':' is the only candidate
            std::string key = line.substr(2, i - 2), val = line.substr(i + 2);
            if (key == "input queries") {
              char* e = nullptr;
              errno = 0;
              long long v = strtoll(val.c_str(), &e, 10);
              if (errno || e == val.c_str() || *e) die("invalid value of input queries");
              input_queries = v;
            }
          }
        }
        continue;
      }
      r.f.clear();
      size_t s = 0;
      for (int i = 0; i < num_fields - 1; i++) {
        size_t t = line.find('\t', s);
        if (t == std::string::npos) break;
        r.f.emplace_back(line, s, t - s);
        s = t + 1;
      }
      r.f.emplace_back(line, s);
      if ((int)r.f.size() < num_fields) die("number of fields (%d) < query index field (%d)", (int)r.f.size(), num_fields);
      const std::string& qi = r.f[num_fields - 1];
      char* e = nullptr;
      errno = 0;
      idx = strtoull(qi.c_str(), &e, 10);
      if (qi.empty() || !isdigit((unsigned char)qi[0]) || errno || *e) die("invalid query index at field %d: %s", num_fields, qi.c_str());
      const std::string& sc = r.f[score_field - 1];
      errno = 0;
      r.score = strtod(sc.c_str(), &e);
      if (sc.empty() || e == sc.c_str() || *e) die("failed to parse score: %s", sc.c_str());
      return true;
    }
    return false;
  }

  // next run of rows sharing one queryIdx
  bool next(Group& g) {
    if (done) return false;
    g.rows.clear();
    if (!have_pending) {
      if (!parse_row(pending, pending_idx)) {
        done = true;
        return false;
      }
      have_pending = true;
    }
    g.qidx = pending_idx;
    g.qid = pending.f[0];
    g.rows.push_back(std::move(pending));
    have_pending = false;
    Row r;
    uint64_t idx;
    while (parse_row(r, idx)) {
      if (idx != g.qidx) {
        pending = std::move(r);
        pending_idx = idx;
        have_pending = true;
        return true;
      }
      g.rows.push_back(std::move(r));
      r = Row();
    }
    done = true;
    return true;
  }
};

struct Out {
  gzFile gz = nullptr;
  FILE* fp = nullptr;
  std::string buf;
  Out(const std::string& path, int level) {
    bool is_gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    if (is_gz) {
      char mode[8];
      snprintf(mode, sizeof mode, "wb%d", level < 0 ? 6 : std::min(level, 9));
      gz = gzopen(path.c_str(), mode);
      if (!gz) die("%s: %s", path.c_str(), strerror(errno));
      gzbuffer(gz, 1 << 18);
    } else {
      fp = path == "-" ? stdout : fopen(path.c_str(), "wb");
      if (!fp) die("%s: %s", path.c_str(), strerror(errno));
    }
    buf.reserve(1 << 20);
  }
  void flush() {
    if (buf.empty()) return;
    if (gz) {
      if (gzwrite(gz, buf.data(), (unsigned)buf.size()) != (int)buf.size()) die("write error");
    } else if (fwrite(buf.data(), 1, buf.size(), fp) != buf.size()) {
      die("write error");
    }
    buf.clear();
  }
  void write(const std::string& s) {
    buf += s;
    if (buf.size() > (1u << 20) - 4096) flush();
  }
  void write(const char* s) { write(std::string(s)); }
  void close() {
    flush();
    if (gz) gzclose(gz);
    else if (fp != stdout) fclose(fp);
    else fflush(fp);
    gz = nullptr;
    fp = nullptr;
  }
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
  for (auto& f : files) parsers.emplace_back(new ResultParser(f, f_qidx, score_field));
  std::vector<Group> head(files.size());
  using Key = std::pair<uint64_t, size_t>;  // (queryIdx, input index): equal queryIdx pop in input order
  std::priority_queue<Key, std::vector<Key>, std::greater<Key>> heap;
  for (size_t i = 0; i < files.size(); i++)
    if (parsers[i]->next(head[i])) heap.push({head[i].qidx, i});

  long long matched = 0;
  std::vector<Row> rows;
  std::vector<size_t> order;
  auto emit = [&]() {
    order.resize(rows.size());
    for (size_t i = 0; i < rows.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rows[a].score > rows[b].score; });
    std::string hits = std::to_string(rows.size());
    for (size_t oi : order) {
      Row& r = rows[oi];
      r.f[f_hits - 1] = hits;
      std::string line;
      for (size_t j = 0; j < r.f.size(); j++) {
        if (j) line.push_back('\t');
        line += r.f[j];
      }
      line.push_back('\n');
      out.write(line);
    }
    rows.clear();
  };
  bool first = true;
  uint64_t pre_idx = 0;
  std::string pre_id;
  while (!heap.empty()) {
    auto [qidx, i] = heap.top();
    heap.pop();
    Group& g = head[i];
    if (first) {
      first = false;
      pre_idx = qidx;
      pre_id = g.qid;
    } else if (qidx != pre_idx) {
      matched++;
      emit();
      pre_idx = qidx;
      pre_id = g.qid;
    } else if (g.qid != pre_id) {
      die("[queryIdx: %" PRIu64 "] unmatched sequence Ids detected: idx '%s' != '%s'. please make sure the search results coming from same query files",
          qidx, g.qid.c_str(), pre_id.c_str());
    }
    for (auto& r : g.rows) rows.push_back(std::move(r));
    if (parsers[i]->next(g)) heap.push({g.qidx, i});
  }
  matched++;  // the last group — also when there was none (merge.go:243)
  emit();

  // merge.go:283-341 (every parser has reached its end, so every trailer has been seen)
  long long first_n = parsers[0]->input_queries, total = first_n;
  for (size_t i = 1; i < parsers.size(); i++) {
    long long n = parsers[i]->input_queries;
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
