// fastmod.hpp — exact a % d for the row address `loc = h % NumSigs` (util-db-search.go:6611, :6811; the reference uses
// bmkessler/fastdiv there).  One division-free form for host and device:
//
//   mh    = floor(ceil(2^128 / d) / 2^64)            (per block, computed once on the host: magic_hi below)
//   q'    = floor(a * mh / 2^64)                     (one 64x64 -> high-64 multiply)
//   r'    = a - q' * d;  r = r' >= d ? r' - d : r'
//
// Why one correction is enough, for 2 <= d < 2^63 and any 64-bit a: with m = ceil(2^128/d) = 2^128/d + t, 0 <= t < 1, and
// mh = m/2^64 - e, 0 <= e < 1:  2^64/d - 1 < mh < 2^64/d + 2^-64, so  a/d - 1 < a*mh/2^64 < a/d + 2^-64.  The fractional part
// of a/d is at most 1 - 1/d <= 1 - 2^-63, hence floor(a*mh/2^64) <= floor(a/d); and a*mh/2^64 > a/d - 1 >= floor(a/d) - 1.
// So q' is floor(a/d) or one less, r' is r or r + d (< 2d < 2^64).  d == 1 gives 0.  (Lemire's 128-bit fastmod, used until
// round 3, needs six 64-bit multiplies per k-mer and block; this one needs two — the row-index phase was 18 % of the COBS
// kernel's time on 128-byte rows, where every lane group computes its own address.)  tests/test_fastmod_cpu.py checks the
// host instantiation against `%` on random and adversarial operands.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KMCPG_HD __host__ __device__ __forceinline__
#else
#define KMCPG_HD inline
#endif

namespace kmcpg {

// high 64 bits of ceil(2^128 / d); 0 for d == 1 (never used then)
inline uint64_t fastmod_magic(uint64_t d) {
  const unsigned __int128 m = (~(unsigned __int128)0) / d + 1;  // = ceil(2^128/d) for d >= 2 (exact quotient when d is a power of two)
  return (uint64_t)(m >> 64);
}

KMCPG_HD uint64_t mulhi_u64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

KMCPG_HD uint64_t fastmod_u64(uint64_t a, uint64_t d, uint64_t mh) {
  const uint64_t r = a - mulhi_u64(a, mh) * d;
  return d == 1 ? 0 : (r >= d ? r - d : r);
}

}  // namespace kmcpg
