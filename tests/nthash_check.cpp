// nthash_check.cpp — host instantiation of kmcp_amd/csrc/nthash.hpp behind one C function: the canonical ntHash of every k-mer of a
// sequence the way a lane of k1_seg_roll produces them (start-up over the first window, then one rolling step per position, with
// the kernel's three look-up tables), zero hashes dropped and the FracMinHash bound applied.  Built as a shared object and driven by
// tests/test_nthash_cpu.py against the oracle.
#include <stddef.h>

#include "../kmcp_amd/csrc/nthash.hpp"

extern "C" long nt_roll_all(const unsigned char* seq, long len, int k, int scaled, unsigned long long max_hash, unsigned long long* out) {
  using namespace kmcpg;
  if (len < k) return 0;
  uint64_t tab[256], tab_out[256], tab_in[8];
  for (int b = 0; b < 256; b++) {  // (the kernel: one entry per thread)
    const uint64_t sd = seed_of(b);
    tab[b] = sd;
    tab_out[b] = nt_tab_out(sd, k);
    if (b < 8) tab_in[b] = nt_tab_in(sd, k);
  }
  uint64_t fh = 0, rh = 0;
  for (int j = 0; j < k; j++) nt_start_step(fh, rh, seq[j], j, tab);
  long n = 0;
  for (long t = 0;; t++) {
    const uint64_t h = fh < rh ? fh : rh;
    if (h != 0 && (!scaled || h <= max_hash)) out[n++] = h;
    if (t + k >= len) break;
    nt_roll_step(fh, rh, seq[t], seq[t + k], tab, tab_out, tab_in);
  }
  return n;
}

// The 2-bit form of k1_seg_roll2, as one lane with a run as long as the sequence: bases folded to codes four at a time exactly as the
// kernel's staging does (codes, validity, fold), then the start-up over the first window and one roll per position with the pair tables
// F2 / R2, the table index of every roll taken from the nibble words the kernel builds (spread + funnel shift by k).  Returns -1 when a
// byte is not A / C / G / T in either case (the kernel hands such a segment to the byte kernel).
extern "C" long nt_roll2_all(const unsigned char* seq, long len, int k, int scaled, unsigned long long max_hash, unsigned long long* out) {
  using namespace kmcpg;
  if (len < k) return 0;
  const long nwords = len / 16 + 12;
  uint32_t* W = new uint32_t[(size_t)nwords]();
  bool bad = false;
  for (long g = 0; g * 16 < len; g++) {
    uint32_t word = 0;
    if (g * 16 + 16 <= len) {
      for (int d = 0; d < 4; d++) {
        uint32_t in = 0;
        for (int b = 0; b < 4; b++) in |= (uint32_t)seq[g * 16 + d * 4 + b] << (8 * b);
        const uint32_t c = nt2_codes4(in);
        bad |= !nt2_valid4(in, c);
        word |= nt2_fold4(c) << (8 * d);
      }
    } else {
      for (int j = 0; g * 16 + j < len; j++) {
        const uint32_t ch = seq[g * 16 + j], c = (ch >> 1) & 3u;
        bad |= (ch & 0xDFu) != (uint32_t)nt2_letter((int)c);
        word |= c << (2 * j);
      }
    }
    W[g] = word;
  }
  if (bad) {
    delete[] W;
    return -1;
  }
  uint64_t S[4], RC[4], F2[16], R2[16];
  for (int i = 0; i < 16; i++) {
    F2[i] = nt2_f2(i >> 2, i & 3, k);
    R2[i] = nt2_r2(i >> 2, i & 3, k);
  }
  for (int i = 0; i < 4; i++) {
    S[i] = seed_of(nt2_letter(i));
    RC[i] = seed_of(nt2_letter(i) & 7);
  }
  uint64_t fh = 0, rh = 0;
  for (int j = 0; j < k; j++) {
    const uint32_t c = (W[j >> 4] >> (2 * (j & 15))) & 3u;
    fh = nt2_rol1(fh) ^ S[c];
    rh ^= rolv(RC[c], j);
  }
  const long npos = len - k + 1;
  const int kw = k >> 4, ksh = 2 * (k & 15);
  long n = 0, t = 0;
  for (long g = 0; t < npos; g++) {
    const uint32_t outw = W[g], inw = nt2_funnel(W[g + kw + 1], W[g + kw], (uint32_t)ksh);
    const uint32_t m[2] = {(nt2_spread(outw & 0xFFFFu) << 2) | nt2_spread(inw & 0xFFFFu), (nt2_spread(outw >> 16) << 2) | nt2_spread(inw >> 16)};
    for (int j = 0; j < 16 && t < npos; j++, t++) {
      const uint64_t h = fh < rh ? fh : rh;
      if (h != 0 && (!scaled || h <= max_hash)) out[n++] = h;
      const uint32_t idx = (m[j >> 3] >> (4 * (j & 7))) & 15u;
      fh = nt2_rol1(fh) ^ F2[idx];
      rh = nt2_ror1(rh ^ R2[idx]);
    }
  }
  delete[] W;
  return n;
}
