// dbformat.hpp — readers for the on-disk formats the search path consumes (kept byte-for-byte):
//   `.uniki` block header   kmcp/cmd/index/serialization.go:383-593 (Reader.readHeader), Header :66-82
//   `__db.yml`              kmcp/cmd/util-db-info.go:46-79 (UnikIndexDBInfo), :98-129
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace kmcpg {

struct UnikiHeader {
  int version = 0;
  int k = 0;
  bool canonical = false;
  bool compact = false;
  int num_hashes = 0;
  uint64_t num_sigs = 0;
  std::vector<std::string> names;   // Names[i][0]: single-set databases have one name per column (index.go:622-626)
  std::vector<uint64_t> gsizes;     // GSizes[i][0]
  std::vector<uint32_t> indices;    // Indices[i][0] = chunkIdx | chunks<<16 (index.go:1096)
  std::vector<uint64_t> sizes;      // Sizes[i]: #k-mers of column i
  uint32_t row_bytes = 0;           // NumRowBytes = (len(Names)+7)/8 (serialization.go:379)
  uint64_t offset0 = 0;             // byte offset of row 0 (util-db-search.go:1207)
  uint64_t file_size = 0;
};

struct DbYml {
  int version = -1;
  int uniki_version = -1;
  std::string alias;
  int k = 0;
  std::vector<int> ks;
  bool hashed = false, canonical = false, scaled = false, minimizer = false, syncmer = false;
  uint32_t scale = 0, minimizer_w = 0, syncmer_s = 0;
  int num_hashes = 0;
  double fpr = 0;
  std::vector<std::string> files;
};

// Both return "" on success, else the error text (wording follows the reference's errors).
std::string read_uniki_header(const std::string& path, UnikiHeader* h);
std::string read_db_yml(const std::string& path, DbYml* y);

}  // namespace kmcpg
