// nthash.hpp — ntHash v1 as kmcp uses it (will-rowe/nthash v0.4.0 behind bio/sketches; canonical hash = min(forward, reverse),
// util-db-search.go:1037-1107 generateKmers), the parts that are plain arithmetic: the seed table, the rotations, and the ROLLING form
// of the recurrence that k1_seg_roll walks along a lane's run of positions.  Compiles for the host as well: tests/nthash_check.cpp
// runs the start-up + roll against the CPU restatement of the reference's k-mer hashes (tests/test_nthash_cpu.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KMCPG_NT_HD __host__ __device__ __forceinline__
#else
#define KMCPG_NT_HD inline
#endif

namespace kmcpg {

KMCPG_NT_HD uint64_t rol1(uint64_t v) { return (v << 1) | (v >> 63); }
KMCPG_NT_HD uint64_t rolv(uint64_t x, int n) {
  n &= 63;
  return (x << n) | (x >> ((64 - n) & 63));
}
KMCPG_NT_HD uint64_t rorv(uint64_t x, int n) {
  n &= 63;
  return (x >> n) | (x << ((64 - n) & 63));
}

// ntHash v1 seed table entry for byte b (rows 0..7 are N,T,N,G,A,A,N,C so that the complement of
// base x is tab[x & 7]); will-rowe/nthash v0.4.0 seedTab.
KMCPG_NT_HD uint64_t seed_of(int b) {
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

// Rolling form, position i -> i + 1 of a k-mer window (F = seed of a base, R = seed of its complement = tab[b & 7]):
//     fh(i+1) = rol1(fh(i) ^ rol(F[i], k-1)) ^ F[i+k]        rh(i+1) = ror1(rh(i) ^ R[i] ^ rol(R[i+k], k))
// with two rotated copies of the seed table instead of two variable 64-bit rotates per k-mer:
//     tab_out[b] = rol(F[b], k-1)  (the base that drops out of fh),   tab_in[j] = rol(tab[j], k), j < 8  (the base that enters rh)
KMCPG_NT_HD uint64_t nt_tab_out(uint64_t seed, int k) { return rolv(seed, k - 1); }  // seed = seed_of(b)
KMCPG_NT_HD uint64_t nt_tab_in(uint64_t seed, int k) { return rolv(seed, k); }       // seed = seed_of(j), j < 8

// one base of the start-up at offset j of the first window: fh = XOR_j rol(F[j], k-1-j), rh = XOR_j rol(R[j], j)
KMCPG_NT_HD void nt_start_step(uint64_t& fh, uint64_t& rh, uint8_t b, int j, const uint64_t* tab) {
  fh = rol1(fh) ^ tab[b];
  rh ^= rolv(tab[b & 7], j);
}

// bo leaves the window, bi enters it
KMCPG_NT_HD void nt_roll_step(uint64_t& fh, uint64_t& rh, uint8_t bo, uint8_t bi, const uint64_t* tab, const uint64_t* tab_out, const uint64_t* tab_in) {
  fh = rol1(fh ^ tab_out[bo]) ^ tab[bi];
  const uint64_t x = rh ^ tab[bo & 7] ^ tab_in[bi & 7];
  rh = (x >> 1) | (x << 63);
}

// ---- the 2-bit form (k1_seg_roll2): for A, C, G, T in either case everything the recurrence needs is a function of two codes
//      code = (ascii >> 1) & 3:  A 0, C 1, T 2, G 3      fh' = rol1(fh) ^ F2[out][in]      rh' = ror1(rh ^ R2[out][in])
constexpr uint32_t NT2_LETTERS = (uint32_t)'A' | ((uint32_t)'C' << 8) | ((uint32_t)'T' << 16) | ((uint32_t)'G' << 24);  // code -> letter
KMCPG_NT_HD uint8_t nt2_letter(int c) { return (uint8_t)((NT2_LETTERS >> (8 * (c & 3))) & 0xFFu); }
KMCPG_NT_HD uint64_t nt2_f2(int o, int i, int k) { return rol1(nt_tab_out(seed_of(nt2_letter(o)), k)) ^ seed_of(nt2_letter(i)); }
KMCPG_NT_HD uint64_t nt2_r2(int o, int i, int k) { return seed_of(nt2_letter(o) & 7) ^ nt_tab_in(seed_of(nt2_letter(i) & 7), k); }
// four bytes -> their codes, each in its byte; -> one byte c0 + 4 c1 + 16 c2 + 64 c3 (the products' other terms stay below bit 24
// or leave the word); -> the four letters those codes stand for (what the bytes must equal, case aside, to be taken by this form)
KMCPG_NT_HD uint32_t nt2_codes4(uint32_t w) { return (w >> 1) & 0x03030303u; }
KMCPG_NT_HD uint32_t nt2_fold4(uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_udot4(c, 0x40100401u, 0u, false);  // v_dot4_u32_u8: c0 + 4 c1 + 16 c2 + 64 c3 at full rate (the 32-bit multiply is quarter rate)
#else
  return (c * 0x01041040u) >> 24;
#endif
}
KMCPG_NT_HD uint32_t nt2_canon4(uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(0u, NT2_LETTERS, c);  // v_perm_b32: selector bytes 0..3 pick the bytes of the second operand
#else
  uint32_t r = 0;
  for (int b = 0; b < 4; b++) r |= (uint32_t)nt2_letter((int)((c >> (8 * b)) & 3u)) << (8 * b);
  return r;
#endif
}
KMCPG_NT_HD bool nt2_valid4(uint32_t w, uint32_t c) { return (w & 0xDFDFDFDFu) == nt2_canon4(c); }
// 8 two-bit fields (16 bits) -> the low halves of 8 nibbles; (spread(out) << 2) | spread(in) = the table index of 8 rolls, a nibble each
KMCPG_NT_HD uint32_t nt2_spread(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  return x;
}
// ({hi, lo} >> s)[31:0], s < 32 (v_alignbit_b32)
KMCPG_NT_HD uint32_t nt2_funnel(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, s);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u));
#endif
}
KMCPG_NT_HD uint64_t nt2_rol1(uint64_t v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  return ((uint64_t)nt2_funnel(hi, lo, 31u) << 32) | nt2_funnel(lo, hi, 31u);
}
KMCPG_NT_HD uint64_t nt2_ror1(uint64_t v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  return ((uint64_t)nt2_funnel(lo, hi, 1u) << 32) | nt2_funnel(hi, lo, 1u);
}

}  // namespace kmcpg
