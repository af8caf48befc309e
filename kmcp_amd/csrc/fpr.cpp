// fpr.cpp — see fpr.hpp.  Compiled with -ffp-contract=off (Go on amd64 does not fuse multiply-add).
#include "fpr.hpp"

#include <math.h>

#include <algorithm>
#include <atomic>

#include <limits>

namespace kmcpg {

double go_pow(double x, double y) {
  if (y == 0 || x == 1) return 1;
  if (y == 1) return x;
  if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<double>::quiet_NaN();
  if (x == 0) return y < 0 ? std::numeric_limits<double>::infinity() : 0.0;
  if (std::isinf(y)) {
    if (x == -1) return 1;
    return ((fabs(x) < 1) == (y > 0)) ? 0.0 : std::numeric_limits<double>::infinity();
  }
  if (std::isinf(x)) return y < 0 ? 0.0 : std::numeric_limits<double>::infinity();
  if (y == 0.5) return sqrt(x);
  if (y == -0.5) return 1 / sqrt(x);
  double yi;
  double yf = modf(fabs(y), &yi);
  if (yf != 0 && x < 0) return std::numeric_limits<double>::quiet_NaN();
  if (yi >= 9.223372036854775808e18) {
    if (x == -1) return 1;
    return ((fabs(x) < 1) == (y > 0)) ? 0.0 : std::numeric_limits<double>::infinity();
  }
  double a1 = 1.0;  // answer = a1 * 2^ae
  int ae = 0;
  if (yf != 0) {
    if (yf > 0.5) {
      yf -= 1;
      yi += 1;
    }
    a1 = exp(yf * log(x));
  }
  int xe;
  double x1 = frexp(x, &xe);
  for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
    if (xe < -(1 << 12) || (1 << 12) < xe) {  // overflow/underflow is certain
      ae += xe;
      break;
    }
    if (i & 1) {
      a1 *= x1;
      ae += xe;
    }
    x1 *= x1;
    xe <<= 1;
    if (x1 < .5) {
      x1 += x1;
      xe--;
    }
  }
  if (y < 0) {
    a1 = 1 / a1;
    ae = -ae;
  }
  return ldexp(a1, ae);
}

// FPR(n, i) for i = 0..upto.  The reference's running value r only ever decreases, and once it is clamped to 0 — the next term
// pushed it below zero, or BinomialCoeff left the float64 range (util-fpr.go:40-47), which happens within ~1 030 terms whatever
// n is — every later value is 0.  The row therefore ENDS at the first dead entry (value 0): FPR(n, k) for k beyond the row is 0.
// A row has at most min(n, ~1 030) + 1 entries, so rows of long queries (thousands of k-mers) are as cheap to keep as those of
// short reads.
static std::vector<double> fpr_row(double p_, int n, int upto) {
  std::vector<double> out;
  out.reserve((size_t)std::min(upto, 1100) + 1);
  // C(n,i) for i <= n/2 exactly as BinomialCoeff's product loop rounds it: a 53-bit mantissa with an
  // unbounded exponent (big.Float), kept here as a normalised (mantissa, exponent) pair, one step per term.
  const int half = n / 2;
  double m = 0.5;
  int e = 1;
  std::vector<double> binom;  // C(n, i) for the i <= n/2 reached so far (the terms beyond n/2 read them back: C(n,i) = C(n,n-i))
  binom.push_back(1.0);
  const double q = 1 - p_;
  double r = 1;
  for (int i = 0; i <= upto; i++) {
    const int bi = i > n - i ? n - i : i;
    while ((int)binom.size() <= bi && (int)binom.size() <= half) {
      const int j = (int)binom.size() - 1;
      int de;
      m = frexp(m * (double)(n - j), &de);
      e += de;
      m = frexp(m / (double)(j + 1), &de);
      e += de;
      binom.push_back(ldexp(m, e));
    }
    const double coeff = binom[(size_t)bi];
    bool dead = false;
    if (coeff > std::numeric_limits<double>::max()) {
      dead = true;
    } else {
      double t = coeff * go_pow(p_, (double)i);
      t = t * go_pow(q, (double)(n - i));
      r -= t;
      if (r < 0) dead = true;
    }
    if (dead) {
      out.push_back(0.0);
      break;
    }
    out.push_back(r);
  }
  return out;
}

uint64_t QueryFpr::next_id() {
  static std::atomic<uint64_t> counter{0};
  return ++counter;
}

const QueryFpr::Row& QueryFpr::row(int n) {
  auto it = rows_.find(n);
  if (it != rows_.end()) return it->second;
  // a host that feeds queries of ever new lengths: start over (rows handed out stay alive through their holders' references)
  if (rows_.size() >= kMaxRows) rows_.clear();
  return rows_.emplace(n, std::make_shared<const std::vector<double>>(fpr_row(p_, n, n))).first->second;
}

QueryFpr::Row QueryFpr::ensure_row(int n) {
  std::lock_guard<std::mutex> g(mu_);
  return row(n);
}

double QueryFpr::get(int n, int k) {
  if (n <= 0) return 1;
  if (k > n) k = n;
  if (k < 0) return 1;
  std::lock_guard<std::mutex> g(mu_);
  const std::vector<double>& r = *row(n);
  return (size_t)k < r.size() ? r[(size_t)k] : 0.0;
}

}  // namespace kmcpg
