// pack2.hpp — the host side of the 2-bit packed upload (host.cpp stage): a batch's bases as 2-bit codes, 4 per byte, plus the runs
// of bytes that are not A/C/G/T/U in either case.  Why this is exact: the k-mer kernels see a base only through the ntHash seed
// tables (nthash.hpp seed_of: 'a' == 'A', ..., 'U' == 'u' == 'T'; complement = tab[b & 7], the same entry for all spellings of a
// base), so the canonical spelling the device unpacks — A, C, G, T — hashes exactly like the byte the caller wrote; every other
// byte is restored verbatim (its seed is 0 but its complement entry depends on its low three bits).
// code = (ascii >> 1) & 3: A 0, C 1, T/U 2, G 3.  Base j sits in bits 2*(j % 4) of byte j / 4.
// Host-only and self-contained: tests/pack2_check.cpp compiles it with g++ (tests/test_pack2_cpu.py).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kmcpg {

struct PackRun {  // = ExcRun of common.hpp (kept separate so that this header needs nothing else)
  uint64_t pos;
  uint32_t len;
  uint32_t byte;
};

inline bool pack2_is_base(uint8_t c) {
  const uint8_t u = c & 0xDF;
  return u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'U';
}

inline void pack2_note(std::vector<PackRun>& exc, uint64_t pos, uint8_t b) {
  if (!exc.empty()) {
    PackRun& l = exc.back();
    if (l.byte == b && l.pos + l.len == pos && l.len < 0xffffffffu) {
      l.len++;
      return;
    }
  }
  exc.push_back(PackRun{pos, 1u, b});
}

// bases [i0, i1) of s -> d (d indexed from the start of s: byte j / 4); i0 is a multiple of 4.  Exceptions get position pos0 + j.
inline void pack2_scalar(const uint8_t* s, size_t i0, size_t i1, uint8_t* d, uint64_t pos0, std::vector<PackRun>& exc) {
  size_t j = i0;
  for (; j + 4 <= i1; j += 4) {
    uint32_t v = 0;
    for (int t = 0; t < 4; t++) {
      const uint8_t c = s[j + t];
      if (pack2_is_base(c)) v |= (uint32_t)((c >> 1) & 3) << (2 * t);
      else pack2_note(exc, pos0 + j + t, c);
    }
    d[j >> 2] = (uint8_t)v;
  }
  if (j < i1) {
    uint32_t v = 0;
    for (int t = 0; j + t < i1; t++) {
      const uint8_t c = s[j + t];
      if (pack2_is_base(c)) v |= (uint32_t)((c >> 1) & 3) << (2 * t);
      else pack2_note(exc, pos0 + j + t, c);
    }
    d[j >> 2] = (uint8_t)v;
  }
}

#if defined(__x86_64__)
// 32 bases per step: compare against the five letters in upper case, codes from one shift, four codes to a byte with two
// multiply-adds (c0 + 4 c1, then + 16 (c2 + 4 c3)); the rare step with a foreign byte notes it and packs 0 in its place
__attribute__((target("avx2"))) inline size_t pack2_avx2(const uint8_t* s, size_t i0, size_t i1, uint8_t* d, uint64_t pos0, std::vector<PackRun>& exc) {
  const __m256i up = _mm256_set1_epi8((char)0xDF), cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'),
                cT = _mm256_set1_epi8('T'), cU = _mm256_set1_epi8('U'), m3 = _mm256_set1_epi8(3);
  const __m256i mul1 = _mm256_set1_epi16(0x0401), mul2 = _mm256_set1_epi32(0x00100001);
  const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  size_t j = i0;
  for (; j + 32 <= i1; j += 32) {
    const __m256i v = _mm256_loadu_si256((const __m256i*)(s + j));
    const __m256i u = _mm256_and_si256(v, up);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT))),
                                       _mm256_cmpeq_epi8(u, cU));
    const __m256i code = _mm256_and_si256(_mm256_and_si256(_mm256_srli_epi16(v, 1), m3), ok);
    const __m256i w16 = _mm256_maddubs_epi16(code, mul1);
    const __m256i w32 = _mm256_madd_epi16(w16, mul2);
    const __m256i by = _mm256_shuffle_epi8(w32, pick);
    const uint32_t lo = (uint32_t)_mm256_extract_epi32(by, 0), hi = (uint32_t)_mm256_extract_epi32(by, 4);
    const uint64_t out = (uint64_t)lo | ((uint64_t)hi << 32);
    memcpy(d + (j >> 2), &out, 8);
    uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(ok);
    while (bad) {
      const int t = __builtin_ctz(bad);
      bad &= bad - 1;
      pack2_note(exc, pos0 + j + t, s[j + t]);
    }
  }
  return j;
}
#endif

inline void pack2_range(const uint8_t* s, size_t i0, size_t i1, uint8_t* d, uint64_t pos0, std::vector<PackRun>& exc) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) i0 = pack2_avx2(s, i0, i1, d, pos0, exc);
#endif
  pack2_scalar(s, i0, i1, d, pos0, exc);
}

// n bases of s -> d[(n + 3) / 4] on up to `threads` threads; false (and nothing usable in d / exc) when more than max_runs
// exception runs turn up — input that is not nucleotide text is better sent as it is
// (*scanned, if given, receives the number of input bytes actually looked at: the host test of the give-up rule)
inline bool pack2_parallel(const uint8_t* s, size_t n, uint8_t* d, std::vector<PackRun>& exc, size_t max_runs, unsigned threads, size_t* scanned = nullptr) {
  exc.clear();
  const size_t kMin = 4u << 20;
  const size_t kSlice = 1u << 20;  // a multiple of 32 bases: the budget is looked at between slices
  const size_t T = std::max<size_t>(1, std::min<size_t>(threads, n / kMin));
  // Input that is not nucleotide text (protein, binary) makes a run of almost every byte: nobody goes on once ONE share has used the
  // whole budget — at most max_runs + one slice of runs per thread are ever held, and the rest of the batch is neither read nor written.
  std::atomic<bool> abort{false};
  std::atomic<size_t> seen{0};
  auto range = [&](size_t a, size_t b, std::vector<PackRun>& out) {
    for (size_t lo = a; lo < b && !abort.load(std::memory_order_relaxed); lo += kSlice) {
      pack2_range(s, lo, std::min(b, lo + kSlice), d, 0, out);
      seen.fetch_add(std::min(b, lo + kSlice) - lo, std::memory_order_relaxed);
      if (out.size() > max_runs) abort.store(true, std::memory_order_relaxed);
    }
  };
  if (T == 1) {
    range(0, n, exc);
    if (scanned) *scanned = seen.load();
    return !abort.load() && exc.size() <= max_runs;
  }
  std::vector<std::vector<PackRun>> part(T);
  auto lo_of = [&](size_t t) { return t >= T ? n : (n / T * t) & ~(size_t)31; };  // multiples of 32 bases = whole packed bytes
  auto work = [&](size_t t) { range(lo_of(t), lo_of(t + 1), part[t]); };
  std::vector<std::thread> th;
  for (size_t t = 1; t < T; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  if (scanned) *scanned = seen.load();
  if (abort.load()) return false;
  size_t total = 0;
  for (auto& v : part) total += v.size();
  if (total > max_runs) return false;
  exc.reserve(total);
  for (auto& v : part) exc.insert(exc.end(), v.begin(), v.end());
  return true;
}

// Appends n bases of s at base position `pos` of the codes array d (any alignment: a reader packs record after record into one batch).
// The byte that holds base `pos` may already carry up to three codes of the previous record: it is completed, not overwritten;
// bytes from the first whole one on are written.  Exceptions get their absolute positions.  d must have room for (pos + n + 3) / 4 + 8 bytes
// (the vector path stores eight bytes at a time).
inline void pack2_append(const uint8_t* s, size_t n, uint8_t* d, uint64_t pos, std::vector<PackRun>& exc) {
  size_t j = 0;
  if (pos & 3) {  // finish the byte the previous record left open
    uint32_t v = d[pos >> 2] & ((1u << (2 * (pos & 3))) - 1u);
    for (; j < n && ((pos + j) & 3); j++) {
      const uint8_t c = s[j];
      if (pack2_is_base(c)) v |= (uint32_t)((c >> 1) & 3) << (2 * ((pos + j) & 3));
      else pack2_note(exc, pos + j, c);
    }
    d[pos >> 2] = (uint8_t)v;
  }
  if (j < n) pack2_range(s + j, 0, n - j, d + ((pos + j) >> 2), pos + j, exc);  // from a whole byte on: source index 0 <-> that byte
}

// what the device does (support.hip k_unpack2 + k_apply_exc), for the host test
inline void unpack2_host(const uint8_t* d, size_t n, const std::vector<PackRun>& exc, uint8_t* out) {
  static const char lut[4] = {'A', 'C', 'T', 'G'};
  for (size_t j = 0; j < n; j++) out[j] = (uint8_t)lut[(d[j >> 2] >> (2 * (j & 3))) & 3];
  for (const PackRun& e : exc)
    for (uint32_t t = 0; t < e.len; t++) out[e.pos + t] = (uint8_t)e.byte;
}

}  // namespace kmcpg
