// device_utils.hpp — small device helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastmod.hpp"

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

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace kmcpg
