// device_utils.hpp — small device helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmcpg {

__device__ __forceinline__ uint64_t rol1(uint64_t v) { return (v << 1) | (v >> 63); }

// ntHash v1 seed table entry for byte b (rows 0..7 are N,T,N,G,A,A,N,C so that the complement of
// base x is tab[x & 7]); will-rowe/nthash v0.4.0 seedTab.
__device__ __forceinline__ uint64_t seed_of(int b) {
  const uint64_t A = 0x3c8bfbb395c60474ULL, C = 0x3193c18562a02b4cULL, G = 0x20323ed082572324ULL,
                 T = 0x295549f54be24456ULL;
  switch (b) {
    case 1: return T;
    case 3: return G;
    case 4: case 5: return A;
    case 7: return C;
    case 'A': case 'a': return A;
    case 'C': case 'c': return C;
    case 'G': case 'g': return G;
    case 'T': case 't': case 'U': case 'u': return T;
    default: return 0;
  }
}

// exact a % d for any 64-bit a, d (Lemire fastmod with a 128-bit magic): replaces fastdiv.Uint64.Mod
// (util-db-search.go:6611,6811).
__device__ __forceinline__ uint64_t fastmod_u64(uint64_t a, uint64_t d, uint64_t mh, uint64_t ml) {
  uint64_t lo = ml * a;
  uint64_t hi = __umul64hi(ml, a) + mh * a;
  uint64_t p_hi = __umul64hi(lo, d);
  uint64_t q_lo = hi * d;
  uint64_t q_hi = __umul64hi(hi, d);
  uint64_t sum = q_lo + p_hi;
  return q_hi + (sum < q_lo ? 1ULL : 0ULL);
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace kmcpg
