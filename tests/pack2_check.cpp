// pack2_check.cpp — the 2-bit packer of the upload path (kmcp_amd/csrc/pack2.hpp) against a byte-by-byte definition:
// unpack(pack(x)) spells every base A/C/G/T (lower case and U folded onto them) and restores every other byte verbatim, and the
// folded spelling has the same ntHash seeds as the original (nthash.hpp seed_of, complement by b & 7).  g++ -O2 -mno-avx2 is NOT
// used: the AVX2 path is selected at run time, the scalar path is exercised through pack2_scalar directly.
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "../kmcp_amd/csrc/nthash.hpp"
#include "../kmcp_amd/csrc/pack2.hpp"

using namespace kmcpg;

static uint8_t fold(uint8_t c) {
  switch (c) {
    case 'a': case 'A': return 'A';
    case 'c': case 'C': return 'C';
    case 'g': case 'G': return 'G';
    case 't': case 'T': case 'u': case 'U': return 'T';
    default: return c;
  }
}

int main() {
  std::mt19937_64 rng(12345);
  // seeds: the folded spelling hashes like the original
  for (int c = 0; c < 256; c++) {
    const uint8_t f = fold((uint8_t)c);
    if (seed_of(c) != seed_of(f) || seed_of(c & 7) != seed_of(f & 7)) {
      printf("FAIL seed of byte %d vs %d\n", c, f);
      return 1;
    }
    if (pack2_is_base((uint8_t)c) != (f != c || c == 'A' || c == 'C' || c == 'G' || c == 'T')) {
      printf("FAIL is_base %d\n", c);
      return 1;
    }
  }
  const char alphabet[] = "ACGTACGTACGTacgtuUNnRYKM-*\n\0\xff";
  size_t cases = 0, runs_total = 0;
  for (int it = 0; it < 400; it++) {
    size_t n = it < 40 ? (size_t)it : (size_t)(rng() % 100000);
    if (it == 399) n = 9u << 20;  // several threads
    std::vector<uint8_t> x(n);
    const int mode = it % 4;  // 0: clean ACGT, 1: a few foreign bytes, 2: long N runs, 3: anything
    for (size_t j = 0; j < n; j++) {
      if (mode == 0) x[j] = "ACGT"[rng() & 3];
      else if (mode == 1) x[j] = (rng() % 997 == 0) ? (uint8_t)alphabet[rng() % (sizeof alphabet - 1)] : (uint8_t)"ACGTacgt"[rng() & 7];
      else if (mode == 2) x[j] = ((j / 777) % 5 == 0) ? 'N' : (uint8_t)"ACGT"[rng() & 3];
      else x[j] = (uint8_t)alphabet[rng() % (sizeof alphabet - 1)];
    }
    std::vector<uint8_t> want(n);
    for (size_t j = 0; j < n; j++) want[j] = fold(x[j]);
    for (int path = 0; path < 3; path++) {
      std::vector<uint8_t> d((n + 3) / 4 + 8, 0xEE), out(n);
      std::vector<PackRun> exc;
      bool ok = true;
      if (path == 0) pack2_scalar(x.data(), 0, n, d.data(), 0, exc);
      else if (path == 1) pack2_range(x.data(), 0, n, d.data(), 0, exc);
      else ok = pack2_parallel(x.data(), n, d.data(), exc, n + 1, 4);
      if (!ok) {
        printf("FAIL parallel refused case %d\n", it);
        return 1;
      }
      unpack2_host(d.data(), n, exc, out.data());
      if (out != want) {
        size_t j = 0;
        while (out[j] == want[j]) j++;
        printf("FAIL case %d path %d at %zu of %zu: got %d want %d\n", it, path, j, n, out[j], want[j]);
        return 1;
      }
      for (size_t e = 0; e + 1 < exc.size(); e++)
        if (exc[e].pos + exc[e].len > exc[e + 1].pos) {
          printf("FAIL runs out of order / overlapping, case %d path %d\n", it, path);
          return 1;
        }
      runs_total += exc.size();
      cases++;
    }
    // the give-up rule
    if (mode == 3 && n > 1000) {
      std::vector<uint8_t> d((n + 3) / 4 + 8);
      std::vector<PackRun> exc;
      if (pack2_parallel(x.data(), n, d.data(), exc, 16, 4)) {
        printf("FAIL garbage input accepted, case %d\n", it);
        return 1;
      }
    }
  }
  {  // ... and it gives up EARLY: 48 MB of protein-like text, every thread stops after the slice in which its budget ran out
    const size_t n = 48u << 20;
    std::vector<uint8_t> x(n);
    for (size_t j = 0; j < n; j++) x[j] = (uint8_t)"ACDEFGHIKLMNPQRSTVWY"[rng() % 20];
    std::vector<uint8_t> d((n + 3) / 4 + 8);
    std::vector<PackRun> exc;
    size_t scanned = 0;
    if (pack2_parallel(x.data(), n, d.data(), exc, n / 256, 4, &scanned)) {
      printf("FAIL protein text accepted\n");
      return 1;
    }
    if (scanned > (8u << 20)) {  // 4 threads x (one 1-MiB slice to use up 196 608 runs, + the slice in flight when the flag went up)
      printf("FAIL gave up only after %zu of %zu bytes\n", scanned, n);
      return 1;
    }
  }
  printf("ok %zu cases %zu runs\n", cases, runs_total);
  return 0;
}
