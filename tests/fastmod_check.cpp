// fastmod_check.cpp — host instantiation of kmcp_amd/csrc/fastmod.hpp against the % operator: random operands, multiples of
// the divisor and their neighbours, the top of the 64-bit range, every small divisor, powers of two and their neighbours.
// Built and run by tests/test_fastmod_cpu.py.
#include <stdio.h>

#include <random>

#include "../kmcp_amd/csrc/fastmod.hpp"

int main() {
  std::mt19937_64 g(20260927);
  unsigned long long bad = 0, n = 0;
  auto chk = [&](uint64_t a, uint64_t d) {
    const uint64_t r = kmcpg::fastmod_u64(a, d, kmcpg::fastmod_magic(d));
    n++;
    if (r != a % d) {
      if (bad < 5) printf("a=%llu d=%llu: %llu, want %llu\n", (unsigned long long)a, (unsigned long long)d, (unsigned long long)r, (unsigned long long)(a % d));
      bad++;
    }
  };
  for (int i = 0; i < 2000000; i++) {
    uint64_t d = g() >> (g() % 63);  // every magnitude of NumSigs, up to 2^63 - 1
    if (d >> 63) d >>= 1;
    if (d < 1) d = 1;
    uint64_t a = g();
    if (i % 3 == 0) a >>= g() % 64;
    const uint64_t q = a / d;
    chk(a, d);
    chk(q * d, d);
    chk(q * d + d - 1, d);
    if (q * d) chk(q * d - 1, d);
    chk(~0ull, d);
    chk(~0ull - g() % d, d);
  }
  for (uint64_t d = 1; d < 3000; d++)
    for (uint64_t a = 0; a < 1500; a++) {
      chk(a, d);
      chk(~0ull - a, d);
    }
  for (int s = 0; s < 63; s++) {
    const uint64_t d = 1ull << s;
    for (int i = 0; i < 1000; i++) {
      chk(g(), d);
      chk(g(), d + 1);
      if (d > 1) chk(g(), d - 1);
    }
  }
  printf("%llu checks, %llu wrong\n", n, bad);
  return bad != 0;
}
