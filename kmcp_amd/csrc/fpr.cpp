// fpr.cpp — see fpr.hpp.  Compiled with -ffp-contract=off (Go on amd64 does not fuse multiply-add).
#include "fpr.hpp"

#include <math.h>

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

// FPR(n, i) for i = 0..upto
static std::vector<double> fpr_row(double p_, int n, int upto) {
  std::vector<double> out((size_t)upto + 1, 0.0);
  // C(n,i) for i <= n/2 exactly as BinomialCoeff's product loop rounds it: a 53-bit mantissa with an
  // unbounded exponent (big.Float), kept here as a normalised (mantissa, exponent) pair.
  const int half = upto < n / 2 ? upto : n / 2;
  std::vector<double> binom((size_t)half + 1);
  double m = 0.5;
  int e = 1;
  binom[0] = 1.0;
  for (int i = 0; i < half; i++) {
    int de;
    m = frexp(m * (double)(n - i), &de);
    e += de;
    m = frexp(m / (double)(i + 1), &de);
    e += de;
    binom[(size_t)i + 1] = ldexp(m, e);
  }
  const double q = 1 - p_;
  double r = 1;
  bool dead = false;
  for (int i = 0; i <= upto; i++) {
    if (!dead) {
      const double coeff = binom[(size_t)(i > n - i ? n - i : i)];
      if (coeff > std::numeric_limits<double>::max()) {
        dead = true;
        r = 0;
      } else {
        double t = coeff * go_pow(p_, (double)i);
        t = t * go_pow(q, (double)(n - i));
        r -= t;
        if (r < 0) {
          dead = true;
          r = 0;
        }
      }
    }
    out[(size_t)i] = r;
  }
  return out;
}

const std::vector<double>& QueryFpr::row(int n) {
  auto it = rows_.find(n);
  if (it != rows_.end()) return it->second;
  return rows_.emplace(n, fpr_row(p_, n, n)).first->second;
}

const std::vector<double>* QueryFpr::ensure_row(int n) {
  std::lock_guard<std::mutex> g(mu_);
  return &row(n);  // unordered_map never moves its mapped values
}

double QueryFpr::get(int n, int k) {
  if (n <= 0) return 1;
  if (k > n) k = n;
  if (k < 0) return 1;
  if (n > kCachedMaxN) return fpr_row(p_, n, k)[(size_t)k];  // long queries: do not keep O(n) rows around
  std::lock_guard<std::mutex> g(mu_);
  return row(n)[(size_t)k];
}

}  // namespace kmcpg
