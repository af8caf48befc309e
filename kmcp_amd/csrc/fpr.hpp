// fpr.hpp — false-positive rate of a query (Theorem 2 of doi:10.1038/nbt.3442) as kmcp computes it:
// kmcp/cmd/util-fpr.go:32-50 (QueryFPR), :54-71 (BinomialCoeff in big.Float), :140-191 (cached form).
// One pass per n yields FPR(n, k) for every k: the reference's running value r is shared by all k.
#pragma once
#include <stdint.h>

#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace kmcpg {

// Go's math.Pow (pure-Go algorithm on amd64) so the FPR column agrees with the Go binary.
double go_pow(double x, double y);

class QueryFpr {
 public:
  explicit QueryFpr(double p) : p_(p), id_(next_id()) {}
  // unique per object for the life of the process (callers that cache a row per thread key it on this, never on an address: a
  // later database may be allocated where an earlier one lived)
  uint64_t id() const { return id_; }
  // queryFPR(n, k): 1 - sum_{i<=k} C(n,i) p^i (1-p)^(n-i), clamped at 0
  double get(int n, int k);
  // Rows are cached for every n (a row ends at its first dead entry, see fpr.cpp: at most ~1 030 + 1 values whatever n is; at
  // most kMaxRows rows are kept, then the cache starts over).  ensure_row builds row n if needed and returns a reference-counted
  // handle: value(*row, n, k) reads FPR(n, k) without a lock, and the row lives as long as anybody holds it, whatever the cache does.
  static constexpr size_t kMaxRows = 65536;
  typedef std::shared_ptr<const std::vector<double>> Row;
  Row ensure_row(int n);
  static double value(const std::vector<double>& row, int n, int k) {
    if (k > n) k = n;
    if (k < 0) return 1;
    return (size_t)k < row.size() ? row[(size_t)k] : 0.0;
  }

 private:
  const Row& row(int n);
  static uint64_t next_id();
  double p_;
  uint64_t id_;
  std::mutex mu_;
  std::unordered_map<int, Row> rows_;
};
typedef QueryFpr::Row FprRow;

}  // namespace kmcpg
