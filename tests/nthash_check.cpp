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
