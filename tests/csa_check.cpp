// csa_check.cpp — host instantiation of kmcp_amd/csrc/csa.hpp (the bit-sliced counters of k2_cobs) against scalar counts:
// the 8-row and 4-row groups of the short-query kernels, and the deferred carries of the 16 / 24-plane kernels driven the way
// the kernel drives them (blocks of up to four 8-row groups, groups past the end of a chunk handing in zero carries), up to the
// largest count the planes hold; and the integer threshold of a query against the reference's float64 test, count by count.
// Built and run by tests/test_csa_cpu.py.
#include <stdio.h>
#include <string.h>

#include <random>
#include <vector>

#include "../kmcp_amd/csrc/csa.hpp"

using namespace kmcpg;

static unsigned long long bad = 0, checked = 0;

template <int NPL>
static void compare(const uint32_t (&pl)[NPL], const std::vector<uint64_t>& want, const char* what) {
  for (int q = 0; q < 32; q++) {
    uint64_t c = 0;
    for (int p = 0; p < NPL; p++) c |= (uint64_t)((pl[p] >> q) & 1u) << p;
    checked++;
    if (c != (want[q] & ((1ull << NPL) - 1))) {
      if (bad < 5) printf("%s, %d planes, column %d: %llu, want %llu\n", what, NPL, q, (unsigned long long)c, (unsigned long long)want[q]);
      bad++;
    }
  }
}

static uint32_t draw(std::mt19937_64& g, int density) {  // density/8 of the bits set
  uint32_t v = 0;
  for (int q = 0; q < 32; q++) v |= (uint32_t)((int)(g() % 8) < density) << q;
  return v;
}

template <int NPL>
static void short_query(std::mt19937_64& g, int rows, int density) {  // csa8 / csa4 in any order, as the 8- and 4-row forms use them
  uint32_t pl[NPL] = {};
  std::vector<uint64_t> want(32, 0);
  int done = 0;
  while (done < rows) {
    uint32_t x[8];
    const int nr = (g() & 1) ? 8 : 4;
    for (int i = 0; i < nr; i++) {
      x[i] = done + i < rows ? draw(g, density) : 0;  // rows past the end of a read are the all-zero row
      for (int q = 0; q < 32; q++) want[q] += (x[i] >> q) & 1u;
    }
    if (nr == 8) csa8<NPL>(pl, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
    else csa4<NPL>(pl, x[0], x[1], x[2], x[3]);
    done += nr;
    if (g() % 16 == 0) compare<NPL>(pl, want, "csa8/csa4");
  }
  compare<NPL>(pl, want, "csa8/csa4");
}

template <int NPL>
static void long_query(std::mt19937_64& g, long rows, int density, bool ragged) {
  uint32_t pl[NPL] = {};
  std::vector<uint64_t> want(32, 0);
  long done = 0;
  while (done < rows) {
    // one block of the kernel's loop: up to four groups; a chunk that ends early leaves the later groups out (en = 0)
    const int groups = ragged ? 1 + (int)(g() % 4) : 4;
    uint32_t e8 = 0, s16 = 0;
    for (int gi = 0; gi < 4; gi++) {
      uint32_t en = 0;
      if (gi < groups) {
        uint32_t x[8];
        for (int i = 0; i < 8; i++) {
          x[i] = done < rows ? draw(g, density) : 0;
          done++;
          for (int q = 0; q < 32; q++) want[q] += (x[i] >> q) & 1u;
        }
        en = csa8_low<NPL>(pl, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
      }
      carry_step<NPL>(pl, gi, en, e8, s16);
    }
    compare<NPL>(pl, want, ragged ? "deferred carries, ragged blocks" : "deferred carries");  // canonical after every block: where the pruning test reads the planes
  }
}

int main() {
  std::mt19937_64 g(20260927);
  for (int rep = 0; rep < 200; rep++) {
    short_query<8>(g, 1 + (int)(g() % 255), 1 + (int)(g() % 8));
    short_query<10>(g, 1 + (int)(g() % 1023), 1 + (int)(g() % 8));
  }
  short_query<8>(g, 255, 8);    // every column reaches 255
  short_query<10>(g, 1023, 8);  // ... 1023
  for (int rep = 0; rep < 40; rep++) {
    long_query<16>(g, 1 + (long)(g() % 9000), 1 + (int)(g() % 8), rep & 1);
    long_query<24>(g, 1 + (long)(g() % 9000), 1 + (int)(g() % 8), rep & 1);
  }
  long_query<16>(g, 65535, 8, false);  // the largest count 16 planes hold, in every column
  long_query<16>(g, 65535, 8, true);
  long_query<24>(g, 300000, 8, true);
  // the integer threshold against the reference's own test, count by count: count >= minMatched && float64(count) > float64(n) * t
  {
    const double ts[] = {0.0, 1e-12, 0.1, 0.3, 0.31, 0.4, 0.5, 0.55, 0.7, 0.8, 0.9, 0.99, 0.9995, 1.0};
    for (int rep = 0; rep < 400000; rep++) {
      const int n = rep < 70000 ? rep : (int)(g() % 16777215) + 1;
      const double t = rep % 3 == 0 ? ts[g() % (sizeof ts / sizeof ts[0])] : (double)(g() >> 11) / 9007199254740992.0;
      const int m = (rep % 5 == 0) ? (int)(g() % 200) + 1 : 10;
      const uint32_t c = count_threshold(n, t, m);
      auto passes = [&](uint64_t x) { return x >= (uint64_t)m && (double)x > (double)n * t; };
      checked++;
      if (!passes(c) || (c > 0 && passes((uint64_t)c - 1))) {
        if (bad < 5) printf("threshold n=%d t=%.17g -c %d: %u\n", n, t, m, c);
        bad++;
      }
    }
  }
  printf("%llu counts checked, %llu wrong\n", checked, bad);
  return bad ? 1 : 0;
}
