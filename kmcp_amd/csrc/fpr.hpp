// fpr.hpp — false-positive rate of a query (Theorem 2 of doi:10.1038/nbt.3442) as kmcp computes it:
// kmcp/cmd/util-fpr.go:32-50 (QueryFPR), :54-71 (BinomialCoeff in big.Float), :140-191 (cached form).
// One pass per n yields FPR(n, k) for every k: the reference's running value r is shared by all k.
#pragma once
#include <stdint.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace kmcpg {

// Go's math.Pow (pure-Go algorithm on amd64) so the FPR column agrees with the Go binary.
double go_pow(double x, double y);

class QueryFpr {
 public:
  explicit QueryFpr(double p) : p_(p) {}
  // queryFPR(n, k): 1 - sum_{i<=k} C(n,i) p^i (1-p)^(n-i), clamped at 0
  double get(int n, int k);
  // rows are cached for n <= kCachedMaxN; ensure_row builds row n (FPR(n, 0..n)) if needed and returns it — the pointer
  // stays valid for the life of the object, so readers need no lock afterwards
  static constexpr int kCachedMaxN = 4096;
  const std::vector<double>* ensure_row(int n);

 private:
  const std::vector<double>& row(int n);
  double p_;
  std::mutex mu_;
  std::unordered_map<int, std::vector<double>> rows_;
};

}  // namespace kmcpg
