// dbformat.cpp — see dbformat.hpp.
#include "dbformat.hpp"

#include <stdio.h>
#include <string.h>

#include <fstream>

namespace kmcpg {

namespace {

struct BeReader {
  FILE* f;
  uint64_t left;  // bytes of the file not consumed yet: every count read from the header is checked against it
  bool ok = true;
  bool read(void* p, size_t n) {
    if (!ok) return false;
    if (n > left || fread(p, 1, n, f) != n) ok = false;
    else left -= n;
    return ok;
  }
  uint32_t u32() {
    uint8_t b[4] = {0, 0, 0, 0};
    read(b, 4);
    return (uint32_t(b[0]) << 24) | (uint32_t(b[1]) << 16) | (uint32_t(b[2]) << 8) | b[3];
  }
  uint64_t u64() {
    uint64_t hi = u32();
    return (hi << 32) | u32();
  }
  // a count of items of at least `item_bytes` each that still have to fit in the file
  bool fits(uint64_t count, uint64_t item_bytes) {
    if (ok && count > left / item_bytes) ok = false;
    return ok;
  }
};

std::string strip(const std::string& s) {
  size_t b = 0, e = s.size();
  while (b < e && (s[b] == ' ' || s[b] == '\t')) b++;
  while (e > b && (s[e - 1] == ' ' || s[e - 1] == '\t' || s[e - 1] == '\r' || s[e - 1] == '\n')) e--;
  std::string t = s.substr(b, e - b);
  if (t.size() >= 2 && ((t.front() == '"' && t.back() == '"') || (t.front() == '\'' && t.back() == '\''))) t = t.substr(1, t.size() - 2);
  return t;
}

bool as_bool(const std::string& v) { return v == "true" || v == "True" || v == "TRUE" || v == "yes" || v == "on"; }

}  // namespace

std::string read_uniki_header(const std::string& path, UnikiHeader* h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "kmcp index file missing: " + path;
  fseeko(f, 0, SEEK_END);
  const uint64_t file_size = (uint64_t)ftello(f);
  fseeko(f, 0, SEEK_SET);
  BeReader r{f, file_size};
  uint8_t magic[8], meta[4];
  if (!r.read(magic, 8) || memcmp(magic, ".kmcpidx", 8) != 0) {
    fclose(f);
    return "kmcp: invalid index format: " + path;  // ErrInvalidIndexFileFormat
  }
  r.read(meta, 4);
  if (!r.ok || meta[0] != 4) {
    fclose(f);
    return "kmcp: version mismatch: " + path;  // ErrVersionMismatch
  }
  h->version = meta[0];
  h->k = meta[1];
  h->canonical = (meta[2] & 1) != 0;
  h->compact = (meta[2] & 2) != 0;
  h->num_hashes = meta[3];
  h->num_sigs = r.u64();
  const uint32_t n = r.u32();
  // every name group costs >= 4 bytes here, >= 4 in the genome-size and index tables and 8 in the size table
  if (r.fits(n, 20)) h->names.assign(n, "");
  std::vector<char> buf;
  for (uint32_t i = 0; i < n && r.ok; i++) {
    const uint32_t len = r.u32();
    if (!r.fits(len, 1)) break;
    buf.resize(len);
    if (len) r.read(buf.data(), len);
    size_t e = 0;
    while (e < len && buf[e] != '\n') e++;
    h->names[i].assign(buf.data(), e);
  }
  const uint32_t ng = r.u32();
  if (r.fits(ng, 4)) h->gsizes.assign(n, 0);
  for (uint32_t i = 0; i < ng && r.ok; i++) {
    const uint32_t m = r.u32();
    if (!r.fits(m, 8)) break;
    for (uint32_t j = 0; j < m && r.ok; j++) {
      const uint64_t v = r.u64();
      if (j == 0 && i < n) h->gsizes[i] = v;
    }
  }
  const uint32_t ni = r.u32();
  if (r.fits(ni, 4)) h->indices.assign(n, 0);
  for (uint32_t i = 0; i < ni && r.ok; i++) {
    const uint32_t m = r.u32();
    if (!r.fits(m, 4)) break;
    for (uint32_t j = 0; j < m && r.ok; j++) {
      const uint32_t v = r.u32();
      if (j == 0 && i < n) h->indices[i] = v;
    }
  }
  if (r.fits(n, 8)) h->sizes.assign(n, 0);
  for (uint32_t i = 0; i < n && r.ok; i++) h->sizes[i] = r.u64();
  if (!r.ok) {
    fclose(f);
    return "kmcp: truncated index file: " + path;
  }
  h->row_bytes = (n + 7) / 8;
  h->offset0 = file_size - r.left;
  h->file_size = file_size;
  fclose(f);
  // ErrTruncateIndexFile; written without the product so that a corrupt NumSigs cannot overflow it
  if (h->row_bytes && h->num_sigs > r.left / h->row_bytes) return "kmcp: truncated index file: " + path;
  return "";
}

std::string read_db_yml(const std::string& path, DbYml* y) {
  std::ifstream in(path);
  if (!in) return "fail to open kmcp database info file: " + path;
  std::string line, section;
  while (std::getline(in, line)) {
    std::string s = strip(line);
    if (s.empty() || s[0] == '#') continue;
    if (s[0] == '-') {
      std::string v = strip(s.substr(1));
      if (section == "files") y->files.push_back(v);
      else if (section == "ks") y->ks.push_back(atoi(v.c_str()));
      continue;
    }
    size_t c = s.find(':');
    if (c == std::string::npos) continue;
    std::string key = strip(s.substr(0, c)), val = strip(s.substr(c + 1));
    section = key;
    if (key == "ks" && !val.empty() && val[0] == '[') {
      for (size_t i = 1; i < val.size();) {
        if (isdigit((unsigned char)val[i])) {
          y->ks.push_back(atoi(val.c_str() + i));
          while (i < val.size() && isdigit((unsigned char)val[i])) i++;
        } else i++;
      }
    } else if (key == "files" && !val.empty() && val[0] == '[') {
      std::string cur;
      for (size_t i = 1; i < val.size(); i++) {
        if (val[i] == ',' || val[i] == ']') {
          std::string t = strip(cur);
          if (!t.empty()) y->files.push_back(t);
          cur.clear();
        } else cur += val[i];
      }
    } else if (key == "version") y->version = atoi(val.c_str());
    else if (key == "unikiVersion") y->uniki_version = atoi(val.c_str());
    else if (key == "alias") y->alias = val;
    else if (key == "k") y->k = atoi(val.c_str());
    else if (key == "hashed") y->hashed = as_bool(val);
    else if (key == "canonical") y->canonical = as_bool(val);
    else if (key == "scaled") y->scaled = as_bool(val);
    else if (key == "scale") y->scale = (uint32_t)strtoul(val.c_str(), nullptr, 10);
    else if (key == "minimizer") y->minimizer = as_bool(val);
    else if (key == "minimizer-w") y->minimizer_w = (uint32_t)strtoul(val.c_str(), nullptr, 10);
    else if (key == "syncmer") y->syncmer = as_bool(val);
    else if (key == "syncmer-s") y->syncmer_s = (uint32_t)strtoul(val.c_str(), nullptr, 10);
    else if (key == "hashes") y->num_hashes = atoi(val.c_str());
    else if (key == "fpr") y->fpr = strtod(val.c_str(), nullptr);
  }
  if (y->version != 4) return "kmcp/index: version mismatch";  // ErrVersionMismatch (util-db-info.go:40,119)
  if (y->ks.empty()) y->ks.push_back(y->k);                    // util-db-info.go:124-126
  if (y->files.empty()) return "no index files";               // util-db-search.go:654-656
  return "";
}

}  // namespace kmcpg
