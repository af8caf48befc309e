// csa.hpp — the bit-sliced counters of k2_cobs: every lane keeps the match counts of its 128 columns (4 dwords) as NPL planes
// (plane p holds bit p of 32 columns' counts), and adds rows to them with carry-save adders (the reference counts a k-mer's
// row into per-column counters one byte at a time: util-db-search.go:6811-6972, Count8 over transposed bytes).
// Host-compilable: tests/csa_check.cpp runs the same functions against scalar counts (tests/test_csa_cpu.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KMCPG_CSA_HD __host__ __device__ __forceinline__
#else
#define KMCPG_CSA_HD inline
#endif

namespace kmcpg {

// carry-save adder: h = majority, l = parity of three words.  CSA3 spells them as one v_bitop3_b32 each (gfx950) for the long-query
// kernels, which run near their issue limits; the short-query kernels (8 / 10 planes) wait for HBM and keep the form and the
// instruction schedule they were tuned with (same-box A/B, tools/ab/r04_call14.sh: the GTDB-scale launch is 1.7 % slower with CSA3 and
// the regrouped loads).
#define CSA(h, l, a_, b_, c_)              \
  {                                        \
    uint32_t u_ = (a_) ^ (b_);             \
    h = ((a_) & (b_)) | (u_ & (c_));       \
    l = u_ ^ (c_);                         \
  }
#if defined(__HIP_DEVICE_COMPILE__)
#define CSA3(h, l, a_, b_, c_)                                  \
  {                                                             \
    const uint32_t a__ = (a_), b__ = (b_), c__ = (c_);          \
    h = __builtin_amdgcn_bitop3_b32(a__, b__, c__, 0xE8);       \
    l = __builtin_amdgcn_bitop3_b32(a__, b__, c__, 0x96);       \
  }
#else
#define CSA3(h, l, a_, b_, c_)                                  \
  {                                                             \
    const uint32_t a__ = (a_), b__ = (b_), c__ = (c_);          \
    h = (a__ & b__) | ((a__ ^ b__) & c__);                      \
    l = a__ ^ b__ ^ c__;                                        \
  }
#endif

// a carry of plane FROM's weight rippling through the planes above
template <int NPL, int FROM>
KMCPG_CSA_HD void ripple(uint32_t (&pl)[NPL], uint32_t e) {
#pragma unroll
  for (int p = FROM; p < NPL; p++) {
    uint32_t t = pl[p] & e;
    pl[p] ^= e;
    e = t;
  }
}

// 8 rows into the planes of weight 1, 2 and 4; returns the carry of weight 8 (the caller reduces the carries of several groups
// before anything ripples: carry_step)
template <int NPL>
KMCPG_CSA_HD uint32_t csa8_low(uint32_t (&pl)[NPL], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5, uint32_t x6,
                               uint32_t x7) {
  uint32_t ta, tb, fa, fb, e;
  CSA3(ta, pl[0], pl[0], x0, x1);
  CSA3(tb, pl[0], pl[0], x2, x3);
  CSA3(fa, pl[1], pl[1], ta, tb);
  CSA3(ta, pl[0], pl[0], x4, x5);
  CSA3(tb, pl[0], pl[0], x6, x7);
  CSA3(fb, pl[1], pl[1], ta, tb);
  CSA3(e, pl[2], pl[2], fa, fb);
  return e;
}

// Deferred carries (16 / 24 planes): group gi = 0..3 of a block of 32 rows hands in its weight-8 carry `en`.  Groups 0 and 2 park
// theirs in e8; group 1 adds e8 + en into plane 3 and parks the weight-16 carry in s16; group 3 does the same, adds the two
// weight-16 carries into plane 4 and lets ONE weight-32 carry ripple from plane 5 on.  After group 3 the planes are the plain
// binary counts again (groups past the end of a chunk hand in en = 0).
template <int NPL>
KMCPG_CSA_HD void carry_step(uint32_t (&pl)[NPL], int gi, uint32_t en, uint32_t& e8, uint32_t& s16) {
  static_assert(NPL >= 6, "needs planes 3, 4 and a ripple from 5");
  if (gi == 0 || gi == 2) {
    e8 = en;
  } else if (gi == 1) {
    CSA3(s16, pl[3], pl[3], e8, en);
  } else {
    uint32_t sb, t;
    CSA3(sb, pl[3], pl[3], e8, en);
    CSA3(t, pl[4], pl[4], s16, sb);
    ripple<NPL, 5>(pl, t);
  }
}

// 8 rows, carry rippled at once (8 / 10 planes).  B3: the adders as v_bitop3 pairs (experiment knob of the narrow-row forms)
#define CSA_SEL(h, l, a_, b_, c_) \
  if constexpr (B3) CSA3(h, l, a_, b_, c_) else CSA(h, l, a_, b_, c_)
template <int NPL, bool B3 = false>
KMCPG_CSA_HD void csa8(uint32_t (&pl)[NPL], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5, uint32_t x6, uint32_t x7) {
  uint32_t ta, tb, fa, fb, e;
  CSA_SEL(ta, pl[0], pl[0], x0, x1);
  CSA_SEL(tb, pl[0], pl[0], x2, x3);
  CSA_SEL(fa, pl[1], pl[1], ta, tb);
  CSA_SEL(ta, pl[0], pl[0], x4, x5);
  CSA_SEL(tb, pl[0], pl[0], x6, x7);
  CSA_SEL(fb, pl[1], pl[1], ta, tb);
  CSA_SEL(e, pl[2], pl[2], fa, fb);
#pragma unroll
  for (int p = 3; p < NPL; p++) {
    uint32_t t = pl[p] & e;
    pl[p] ^= e;
    e = t;
  }
}

// the same for 4 rows (the short groups of the zone where sectors die, see k2_cobs)
template <int NPL, bool B3 = false>
KMCPG_CSA_HD void csa4(uint32_t (&pl)[NPL], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
  uint32_t ta, tb, e;
  CSA_SEL(ta, pl[0], pl[0], x0, x1);
  CSA_SEL(tb, pl[0], pl[0], x2, x3);
  CSA_SEL(e, pl[1], pl[1], ta, tb);
#pragma unroll
  for (int p = 2; p < NPL; p++) {
    uint32_t t = pl[p] & e;
    pl[p] ^= e;
    e = t;
  }
}

// The integer threshold a column's count has to reach: the reference keeps a column when count >= minMatched and
// float64(count) > float64(NumKmers) * queryCov (util-db-search.go:7468-7470) — the smallest such integer, from one float64 product
// (no fused multiply-add: the product is rounded exactly as Go rounds it).  n * t >= 0.
KMCPG_CSA_HD uint32_t count_threshold(int n, double min_qcov, int min_matched) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double thr = __dmul_rn((double)n, min_qcov);
#else
  const double thr = (double)n * min_qcov;
#endif
  uint32_t cmin = (uint32_t)thr + 1u;  // smallest integer c with (double)c > thr
  if (cmin < (uint32_t)min_matched) cmin = (uint32_t)min_matched;
  return cmin;
}

}  // namespace kmcpg
