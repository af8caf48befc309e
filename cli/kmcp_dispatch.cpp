// kmcp: the reference's command spelling in front of this build's two commands, so that scripts written for the reference
// (`kmcp search -d db reads.fq.gz -o out.tsv.gz`, `kmcp merge -o all.tsv.gz a.tsv.gz b.tsv.gz`) run unchanged.
//
//   kmcp [persistent flags] search ...  -> kmcp-search (the MI355X search path; flags of kmcp/cmd/search.go:1031-1107)
//   kmcp [persistent flags] merge  ...  -> kmcp-merge  (kmcp/cmd/merge.go)
//   kmcp <any other command> ...        -> the reference binary named by $KMCP_REFERENCE_BIN, or the next `kmcp` on PATH that
//                                          is not this file (compute / index / profile / utils are out of scope of this build,
//                                          SURVEY.md §2); without one: an error that says so, exit status 255 like checkError.
//
// cobra accepts the root command's persistent flags (-j/--threads, -q/--quiet, -i/--infile-list, --log; root.go:62-82) before
// the sub-command: they are handed on behind it.
#include <limits.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

static std::string self_dir() {
  char buf[PATH_MAX];
  ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
  if (n <= 0) return ".";
  buf[n] = 0;
  std::string p(buf);
  size_t s = p.rfind('/');
  return s == std::string::npos ? "." : p.substr(0, s);
}

static bool same_file(const std::string& a, const std::string& b) {
  struct stat sa, sb;
  return stat(a.c_str(), &sa) == 0 && stat(b.c_str(), &sb) == 0 && sa.st_dev == sb.st_dev && sa.st_ino == sb.st_ino;
}

static std::string find_reference() {
  if (const char* e = getenv("KMCP_REFERENCE_BIN")) return e;
  const char* path = getenv("PATH");
  if (!path) return "";
  const std::string me = self_dir() + "/kmcp";
  std::string p(path);
  size_t b = 0;
  while (b <= p.size()) {
    size_t e = p.find(':', b);
    if (e == std::string::npos) e = p.size();
    const std::string cand = (e > b ? p.substr(b, e - b) : std::string(".")) + "/kmcp";
    if (access(cand.c_str(), X_OK) == 0 && !same_file(cand, me)) return cand;
    b = e + 1;
  }
  return "";
}

[[noreturn]] static void run(const std::string& bin, const std::vector<std::string>& args) {
  std::vector<char*> av;
  av.push_back(const_cast<char*>(bin.c_str()));
  for (const auto& a : args) av.push_back(const_cast<char*>(a.c_str()));
  av.push_back(nullptr);
  execv(bin.c_str(), av.data());
  fprintf(stderr, "[ERRO] kmcp: cannot run %s: %s\n", bin.c_str(), strerror(errno));
  exit(255);
}

int main(int argc, char** argv) {
  std::vector<std::string> before, after;
  std::string cmd;
  int i = 1;
  for (; i < argc; i++) {
    const std::string a = argv[i];
    if (a.empty() || a[0] != '-' || a == "-") {
      cmd = a;
      i++;
      break;
    }
    before.push_back(a);
    // persistent flags that take a value in a separate word
    if (a == "-j" || a == "--threads" || a == "-i" || a == "--infile-list" || a == "--log") {
      if (i + 1 < argc) before.push_back(argv[++i]);
    }
  }
  for (; i < argc; i++) after.push_back(argv[i]);
  const std::string dir = self_dir();
  if (cmd == "search" || cmd == "merge") {
    std::vector<std::string> args = after;
    args.insert(args.end(), before.begin(), before.end());
    run(dir + (cmd == "search" ? "/kmcp-search" : "/kmcp-merge"), args);
  }
  if (cmd.empty()) {
    bool help = false, version = false;
    for (const auto& b : before) {
      help |= b == "-h" || b == "--help";
      version |= b == "-V" || b == "--version";
    }
    if (version) run(dir + "/kmcp-search", {"--version"});
    fputs("kmcp (MI355X build of the `kmcp search` hot path)\n\nUsage:\n  kmcp search [flags]   search sequences against a database on the GPU (kmcp-search)\n"
          "  kmcp merge  [flags]   merge search results from several databases (kmcp-merge)\n\n"
          "Every other kmcp command (compute, index, profile, utils, ...) is handed to the reference binary:\n"
          "$KMCP_REFERENCE_BIN, or the next `kmcp` on PATH.\n", help ? stdout : stderr);
    return help ? 0 : 255;
  }
  const std::string ref = find_reference();
  if (ref.empty()) {
    fprintf(stderr, "[ERRO] kmcp %s is not part of this build (only `search` and `merge` are); no reference kmcp binary found "
                    "(set KMCP_REFERENCE_BIN or put it on PATH)\n", cmd.c_str());
    return 255;
  }
  std::vector<std::string> args = before;
  args.push_back(cmd);
  args.insert(args.end(), after.begin(), after.end());
  run(ref, args);
}
