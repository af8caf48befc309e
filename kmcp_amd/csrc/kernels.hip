// kernels.hip — hand-written CDNA4 (gfx950) kernels of the kmcp search hot path.
//
//   K1  k1_kmers / k1_kmers_wg   ntHash canonical k-mer hashes of batched reads, FracMinHash filter, Closed-Syncmer and
//                     Minimizer selection (one wave per short read; one 1024-thread workgroup with LDS tiles per long
//                     read): replaces bio/sketches NextHash / NextSyncmer / NextMinimizer behind generateKmers
//                     (kmcp/cmd/util-db-search.go:1037-1107)
//   K1d k_dedup       per-read sort + unique when #k-mers > -u (util-db-search.go:874-908); whole genomes go to
//                     sort_huge.hip
//   K2  k2_cobs       the COBS query: row = hash % NumSigs, gather rows, AND the h rows, per-column match counts in
//                     bit-sliced counters, integer threshold, hit emission (util-db-search.go:6611-7742); SPLIT form +
//                     k_threshold_long for long queries
//   k_repack / k_gather_rows / k_synth_fill / k_plant* / k_build_scatter: load-time layout, parity helpers, synthetic
//                     index, `kmcp index` scatter (index.go:1107-1309)
//
// The path is bitwise/integer and HBM-bound; there is no MFMA here by design.  wave = 64 lanes.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"

namespace kmcpg {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------
// K1: k-mer generation.  ntHash of the k-mer at position i in closed form:
//     fh(i) = XOR_j rol(F[i+j], k-1-j),   rh(i) = XOR_j rol(R[i+j], j)      (F = seed of the base, R = of its complement)
// Every term is a rotation of a per-position value by an amount that depends on i+j only up to a common rotation, so with
// the prefix XORs  P(n) = XOR_{m<n} ror(F[m], m)  and  Q(n) = XOR_{m<n} rol(R[m], m)
//     fh(i) = rol(P(i+k) ^ P(i), k-1+i),   rh(i) = ror(Q(i+k) ^ Q(i), i)
// i.e. one XOR scan over the bases gives the hashes of every k (and of the s-mers of a syncmer) for two look-ups each,
// instead of k table look-ups per k-mer.  The scans run on DPP within a wave (row_shr 1/2/4/8, row_bcast 15/31).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rolv(uint64_t x, int n) {
  n &= 63;
  return (x << n) | (x >> ((64 - n) & 63));
}
__device__ __forceinline__ uint64_t rorv(uint64_t x, int n) {
  n &= 63;
  return (x >> n) | (x << ((64 - n) & 63));
}

// inclusive XOR scan over the 64 lanes of a wave (all lanes must be active)
__device__ __forceinline__ uint32_t wave_xor_scan32(uint32_t v) {
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8: scan within rows of 16
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint64_t wave_xor_scan(uint64_t v) {
  const uint32_t lo = wave_xor_scan32((uint32_t)v), hi = wave_xor_scan32((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_last(uint64_t v) {  // lane 63's value, uniform
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
}

// One wave per read: a tile is 64 consecutive bases, lane = base index & 63 (also its rotation amount).
struct WTile {
  uint64_t ip, iq;  // inclusive prefixes P(e+1), Q(e+1) at this lane's base e
  uint64_t xp, xq;  // this base's own terms
};
constexpr int K1_SCAN_MAX_K = 65;  // a k-mer may reach into the next tile only

__device__ __forceinline__ WTile wave_tile(const uint8_t* __restrict__ s, int len, int e0, const uint64_t* tab, uint64_t& cp, uint64_t& cq, int lane) {
  const int e = e0 + lane;
  uint64_t F = 0, R = 0;
  if (e < len) {
    const uint8_t b = s[e];
    F = tab[b];
    R = tab[b & 7];
  }
  WTile t;
  t.xp = rorv(F, lane);
  t.xq = rolv(R, lane);
  t.ip = wave_xor_scan(t.xp) ^ cp;
  t.iq = wave_xor_scan(t.xq) ^ cq;
  cp = wave_last(t.ip);
  cq = wave_last(t.iq);
  return t;
}

// canonical hash of the kk-mer starting at this lane's base of tile c (n = the following tile); every lane must call it
__device__ __forceinline__ uint64_t wave_hash(const WTile& c, const WTile& n, int kk, int lane) {
  const int src = lane + kk - 1;  // the k-mer's last base
  const uint64_t ec = __shfl(c.ip, src & 63), en = __shfl(n.ip, src & 63);
  const uint64_t qc = __shfl(c.iq, src & 63), qn = __shfl(n.iq, src & 63);
  const uint64_t dp = (src >= 64 ? en : ec) ^ c.ip ^ c.xp;
  const uint64_t dq = (src >= 64 ? qn : qc) ^ c.iq ^ c.xq;
  const uint64_t f = rolv(dp, kk - 1 + lane), r = rorv(dq, lane);
  return f < r ? f : r;
}

__device__ __forceinline__ int hash_mate_scan(const uint8_t* __restrict__ s, int len, int k, const uint64_t* tab, bool scaled, uint64_t max_hash,
                                              uint64_t* __restrict__ out, int cnt, int lane) {
  const int npos = len - k + 1;
  if (npos <= 0) return cnt;  // ErrShortSeq => no k-mers (util-db-search.go:1060-1062)
  uint64_t cp = 0, cq = 0;
  WTile cur = wave_tile(s, len, 0, tab, cp, cq, lane);
  for (int base = 0; base < npos; base += 64) {
    const WTile nxt = wave_tile(s, len, base + 64, tab, cp, cq, lane);
    const uint64_t h = wave_hash(cur, nxt, k, lane);
    const bool keep = base + lane < npos && h != 0 && (!scaled || h <= max_hash);  // :1097-1103
    const uint64_t m = __ballot(keep);
    if (keep) out[cnt + __popcll(m & ((1ULL << lane) - 1ULL))] = h;
    cnt += __popcll(m);
    cur = nxt;
  }
  return cnt;
}

// all canonical k1-mer (and, if out2, k2-mer) hashes of s, uncompacted (input of the window sketches)
__device__ __forceinline__ void hash_positions_scan(const uint8_t* __restrict__ s, int len, int k1, uint64_t* __restrict__ out1, int k2,
                                                    uint64_t* __restrict__ out2, const uint64_t* tab, int lane) {
  const int n1 = len - k1 + 1, n2 = out2 ? len - k2 + 1 : 0;
  const int nmax = n1 > n2 ? n1 : n2;
  uint64_t cp = 0, cq = 0;
  WTile cur = wave_tile(s, len, 0, tab, cp, cq, lane);
  for (int base = 0; base < nmax; base += 64) {
    const WTile nxt = wave_tile(s, len, base + 64, tab, cp, cq, lane);
    const uint64_t h1 = wave_hash(cur, nxt, k1, lane);
    if (base + lane < n1) out1[base + lane] = h1;
    if (out2) {
      const uint64_t h2 = wave_hash(cur, nxt, k2, lane);
      if (base + lane < n2) out2[base + lane] = h2;
    }
    cur = nxt;
  }
}

// Fallback for k > 65 (the closed form evaluated per k-mer); kept hashes are compacted in order with a wave ballot.
__device__ __forceinline__ int hash_mate(const uint8_t* __restrict__ s, int len, int k, const uint64_t* tab, bool scaled,
                                         uint64_t max_hash, uint64_t* __restrict__ out, int cnt, int lane) {
  if (k <= K1_SCAN_MAX_K) return hash_mate_scan(s, len, k, tab, scaled, max_hash, out, cnt, lane);
  const int npos = len - k + 1;
  if (npos <= 0) return cnt;  // ErrShortSeq => no k-mers (util-db-search.go:1060-1062)
  for (int base = 0; base < npos; base += 64) {
    const int i = base + lane;
    const bool v = i < npos;
    uint64_t h = 0;
    if (v) {
      uint64_t f = 0, r = 0;
      for (int j = 0; j < k; j++) {
        f = rol1(f) ^ tab[s[i + j]];
        r = rol1(r) ^ tab[s[i + k - 1 - j] & 7];
      }
      h = f < r ? f : r;
    }
    const bool keep = v && h != 0 && (!scaled || h <= max_hash);  // :1097-1103
    const uint64_t m = __ballot(keep);
    if (keep) out[cnt + __popcll(m & ((1ULL << lane) - 1ULL))] = h;
    cnt += __popcll(m);
  }
  return cnt;
}

// all canonical kk-mer hashes of s, uncompacted (input of the window sketches)
__device__ __forceinline__ void hash_positions(const uint8_t* __restrict__ s, int len, int kk, const uint64_t* tab, uint64_t* __restrict__ out,
                                               int lane) {
  const int npos = len - kk + 1;
  for (int i = lane; i < npos; i += 64) {
    uint64_t f = 0, r = 0;
    for (int j = 0; j < kk; j++) {
      f = rol1(f) ^ tab[s[i + j]];
      r = rol1(r) ^ tab[s[i + kk - 1 - j] & 7];
    }
    out[i] = f < r ? f : r;
  }
}

__device__ __forceinline__ int argmin_left(const uint64_t* __restrict__ h, int b, int n) {
  int m = b;
  uint64_t mv = h[b];
  for (int i = b + 1; i < b + n; i++) {
    const uint64_t v = h[i];
    if (v < mv) {  // strict: the leftmost of equal values wins
      mv = v;
      m = i;
    }
  }
  return m;
}

// Closed Syncmer as bio/sketches emits it (NextSyncmer, call site util-db-search.go:1053,1068; semantics pinned by
// demo-searching/README.md:61-68): window of 2k-s-1 bases = 2(k-s) s-mers, m = leftmost minimal canonical s-mer;
// emit the k-mer starting at m if m-w0 < k-s, else the k-mer ending at m+s.  One emission per window.
__device__ __forceinline__ int syncmer_mate(const uint8_t* __restrict__ s, int len, int k, int sm, const uint64_t* tab, bool scaled,
                                            uint64_t max_hash, uint64_t* hk, uint64_t* hs, uint64_t* __restrict__ out, int cnt, int lane) {
  const int L = 2 * k - sm - 1;
  if (sm < 1 || sm > k || len < L || len < k) return cnt;  // ErrShortSeq
  if (k <= K1_SCAN_MAX_K) {
    hash_positions_scan(s, len, k, hk, sm, hs, tab, lane);
  } else {
    hash_positions(s, len, k, tab, hk, lane);
    hash_positions(s, len, sm, tab, hs, lane);
  }
  __threadfence_block();
  const int wsz = 2 * (k - sm);
  const int nw = wsz > 0 ? len - L + 1 : len - k + 1;  // s == k: every k-mer is its own window
  for (int base = 0; base < nw; base += 64) {
    const int w0 = base + lane;
    const bool v = w0 < nw;
    uint64_t h = 0;
    if (v) {
      int pos = w0;
      if (wsz > 0) {
        const int m = argmin_left(hs, w0, wsz);
        pos = (m - w0 < k - sm) ? m : m + sm - k;
      }
      h = hk[pos];
    }
    const bool keep = v && h != 0 && (!scaled || h <= max_hash);
    const uint64_t mk = __ballot(keep);
    if (keep) out[cnt + __popcll(mk & ((1ULL << lane) - 1ULL))] = h;
    cnt += __popcll(mk);
  }
  return cnt;
}

// Minimizer sketch (NextMinimizer, call site util-db-search.go:1055,1081): leftmost minimum of every window of w
// k-mers, emitted when its position changes.  (Parity unpinned: the reference holds no golden for this mode.)
__device__ __forceinline__ int minimizer_mate(const uint8_t* __restrict__ s, int len, int k, int w, const uint64_t* tab, bool scaled,
                                              uint64_t max_hash, uint64_t* hk, uint64_t* __restrict__ out, int cnt, int lane) {
  if (w < 1 || len < k + w - 1) return cnt;  // ErrShortSeq
  if (k <= K1_SCAN_MAX_K) hash_positions_scan(s, len, k, hk, 0, nullptr, tab, lane);
  else hash_positions(s, len, k, tab, hk, lane);
  __threadfence_block();
  const int nw = len - k + 1 - w + 1;
  for (int base = 0; base < nw; base += 64) {
    const int w0 = base + lane;
    const bool v = w0 < nw;
    int m = -1, pm = -2;
    if (v) {
      m = argmin_left(hk, w0, w);
      pm = w0 > 0 ? argmin_left(hk, w0 - 1, w) : -2;
    }
    const uint64_t h = (v && m != pm) ? hk[m] : 0;
    const bool keep = v && m != pm && h != 0 && (!scaled || h <= max_hash);
    const uint64_t mk = __ballot(keep);
    if (keep) out[cnt + __popcll(mk & ((1ULL << lane) - 1ULL))] = h;
    cnt += __popcll(mk);
  }
  return cnt;
}

__device__ __forceinline__ int sketch_mate(const K1Args& a, const uint8_t* s, int len, const uint64_t* tab, uint64_t* tmp_k, uint64_t* tmp_s,
                                           uint64_t* out, int cnt, int lane) {
  if (a.mode == 2) return syncmer_mate(s, len, a.k, (int)a.w_or_s, tab, a.scaled != 0, a.max_hash, tmp_k, tmp_s, out, cnt, lane);
  if (a.mode == 1) return minimizer_mate(s, len, a.k, (int)a.w_or_s, tab, a.scaled != 0, a.max_hash, tmp_k, out, cnt, lane);
  return hash_mate(s, len, a.k, tab, a.scaled != 0, a.max_hash, out, cnt, lane);
}

__global__ void __launch_bounds__(256) k1_kmers(const K1Args a) {
  __shared__ uint64_t tab[256];
  tab[threadIdx.x] = seed_of(threadIdx.x);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < a.n_reads; r += nwaves) {
    const uint64_t o1 = a.offs[r];
    const int len1 = (int)(a.offs[r + 1] - o1);
    uint64_t o2 = 0;
    int len2 = 0;
    const bool pe = a.offs2 != nullptr;
    if (pe) {
      o2 = a.offs2[r];
      len2 = (int)(a.offs2[r + 1] - o2);
    }
    uint64_t* out = a.hashes + o1 + o2;
    // skip short query: handleQuery :778-786
    const bool skip = len1 < a.min_qlen && !(pe && len2 >= a.min_qlen);
    int cnt = 0, cnt1 = 0;
    if (!skip) {
      uint64_t* tk = a.scratch ? a.scratch + o1 + o2 : nullptr;   // k-mer hashes of the mate being sketched
      uint64_t* ts = a.scratch2 ? a.scratch2 + o1 + o2 : nullptr;  // its s-mer hashes (syncmer mode)
      cnt = sketch_mate(a, a.seqs + o1, len1, tab, tk, ts, out, 0, lane);
      cnt1 = cnt;
      if (pe) {
        __threadfence_block();
        cnt = sketch_mate(a, a.seqs2 + o2, len2, tab, tk, ts, out, cnt, lane);
      }
    }
    if (lane == 0) {
      a.nk_raw[r] = cnt;
      a.nk1[r] = cnt1;
      a.qlen[r] = len1 + len2;
    }
  }
}

// ---- long queries (HiFi reads, -g whole genomes): one 1024-thread workgroup per read -------------------------
constexpr int K1WG = 1024;

// ordered compaction of one tile of K1WG candidates into out[cnt...]; returns the new (uniform) count
__device__ __forceinline__ int wg_compact(bool keep, uint64_t h, uint64_t* __restrict__ out, int cnt, int* s_wave, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const uint64_t m = __ballot(keep);
  if (lane == 0) s_wave[w] = __popcll(m);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int i = 0; i < K1WG / 64; i++) {
    const int c = s_wave[i];
    if (i < w) before += c;
    total += c;
  }
  if (keep) out[cnt + before + __popcll(m & ((1ULL << lane) - 1ULL))] = h;
  __syncthreads();
  return cnt + total;
}

__device__ __forceinline__ uint64_t hash_at(const uint8_t* __restrict__ s, int i, int kk, const uint64_t* tab) {
  uint64_t f = 0, r = 0;
  for (int j = 0; j < kk; j++) {
    f = rol1(f) ^ tab[s[i + j]];
    r = rol1(r) ^ tab[s[i + kk - 1 - j] & 7];
  }
  return f < r ? f : r;
}

__device__ __forceinline__ int wg_sketch_mate(const K1Args& a, const uint8_t* __restrict__ s, int len, const uint64_t* tab, uint64_t* hk, uint64_t* hs,
                                              uint64_t* __restrict__ out, int cnt, int* s_wave, int tid) {
  const int k = a.k;
  const bool scaled = a.scaled != 0;
  if (a.mode == 0) {
    const int npos = len - k + 1;
    if (npos <= 0) return cnt;
    for (int base = 0; base < npos; base += K1WG) {
      const int i = base + tid;
      const bool v = i < npos;
      const uint64_t h = v ? hash_at(s, i, k, tab) : 0;
      cnt = wg_compact(v && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
    }
    return cnt;
  }
  if (a.mode == 2) {  // closed syncmer, see syncmer_mate
    const int sm = (int)a.w_or_s, L = 2 * k - sm - 1;
    if (sm < 1 || sm > k || len < L || len < k) return cnt;
    for (int i = tid; i < len - k + 1; i += K1WG) hk[i] = hash_at(s, i, k, tab);
    for (int i = tid; i < len - sm + 1; i += K1WG) hs[i] = hash_at(s, i, sm, tab);
    __threadfence_block();
    __syncthreads();
    const int wsz = 2 * (k - sm);
    const int nw = wsz > 0 ? len - L + 1 : len - k + 1;
    for (int base = 0; base < nw; base += K1WG) {
      const int w0 = base + tid;
      const bool v = w0 < nw;
      uint64_t h = 0;
      if (v) {
        int pos = w0;
        if (wsz > 0) {
          const int m = argmin_left(hs, w0, wsz);
          pos = (m - w0 < k - sm) ? m : m + sm - k;
        }
        h = hk[pos];
      }
      cnt = wg_compact(v && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
    }
    __syncthreads();
    return cnt;
  }
  // minimizer, see minimizer_mate
  const int w = (int)a.w_or_s;
  if (w < 1 || len < k + w - 1) return cnt;
  for (int i = tid; i < len - k + 1; i += K1WG) hk[i] = hash_at(s, i, k, tab);
  __threadfence_block();
  __syncthreads();
  const int nw = len - k + 1 - w + 1;
  for (int base = 0; base < nw; base += K1WG) {
    const int w0 = base + tid;
    const bool v = w0 < nw;
    int m = -1, pm = -2;
    if (v) {
      m = argmin_left(hk, w0, w);
      pm = w0 > 0 ? argmin_left(hk, w0 - 1, w) : -2;
    }
    const uint64_t h = (v && m != pm) ? hk[m] : 0;
    cnt = wg_compact(v && m != pm && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
  }
  __syncthreads();
  return cnt;
}

// LDS-tiled form of wg_sketch_mate: the prefix XORs P, Q (see the K1 header) of one tile — 1024 positions + halo — are
// built in LDS by wave scans + a scan of the 64-base group totals, after which any k-mer or s-mer hash of the tile costs
// four LDS reads; the window scans never go to global memory.  Usable while the halo fits (L = 2k-s-1 <= 512 for syncmers,
// w < 512 for minimizers); otherwise the scratch-buffer version above is used.
constexpr int K1H = 512;
constexpr int K1CAP = 2 * K1WG;  // bases per tile: 1024 positions + a halo of at most 1024
struct K1Lds {
  uint64_t ip[K1CAP + 1];                   // ip[n] = P(n) = XOR_{m<n} ror(F[m], m) over the tile's bases, ip[0] = 0
  uint64_t iq[K1CAP + 1];                   // iq[n] = Q(n)
  uint64_t tp[K1CAP / 64], tq[K1CAP / 64];  // totals of the 64-base groups, then their exclusive prefixes
  uint64_t hw[K1WG + K1H];                  // the hashes the windows scan: s-mers (syncmer) or k-mers (minimizer)
};

// builds L.ip / L.iq over the nb (<= K1CAP) bases at s; all K1WG threads call it; ends with a barrier
__device__ __forceinline__ void wg_prefix(const uint8_t* __restrict__ s, int nb, const uint64_t* tab, K1Lds& L, int tid) {
  const int lane = tid & 63;
  const int rounds = nb > K1WG ? 2 : 1;
  uint64_t ip[2] = {0, 0}, iq[2] = {0, 0};
#pragma unroll
  for (int r = 0; r < 2; r++) {
    if (r < rounds) {
      const int e = r * K1WG + tid;
      uint64_t F = 0, R = 0;
      if (e < nb) {
        const uint8_t b = s[e];
        F = tab[b];
        R = tab[b & 7];
      }
      ip[r] = wave_xor_scan(rorv(F, lane));  // K1WG % 64 == 0: e & 63 == lane
      iq[r] = wave_xor_scan(rolv(R, lane));
      if (lane == 63) {
        L.tp[e >> 6] = ip[r];
        L.tq[e >> 6] = iq[r];
      }
    }
  }
  __syncthreads();
  if (tid < 64) {  // one wave scans the group totals
    const int ng = rounds * (K1WG / 64);
    const uint64_t a = tid < ng ? L.tp[tid] : 0, b = tid < ng ? L.tq[tid] : 0;
    const uint64_t sa = wave_xor_scan(a), sb = wave_xor_scan(b);
    if (tid < ng) {
      L.tp[tid] = sa ^ a;
      L.tq[tid] = sb ^ b;
    }
    if (tid == 0) L.ip[0] = L.iq[0] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; r++) {
    if (r < rounds) {
      const int e = r * K1WG + tid;
      L.ip[e + 1] = ip[r] ^ L.tp[e >> 6];
      L.iq[e + 1] = iq[r] ^ L.tq[e >> 6];
    }
  }
  __syncthreads();
}

// canonical hash of the kk-mer at tile position i (i + kk <= nb of the last wg_prefix)
__device__ __forceinline__ uint64_t lds_hash(const K1Lds& L, int i, int kk) {
  const uint64_t f = rolv(L.ip[i + kk] ^ L.ip[i], kk - 1 + i), r = rorv(L.iq[i + kk] ^ L.iq[i], i);
  return f < r ? f : r;
}

// positions per tile: with a small halo the tile shrinks so that positions + halo fit one scan round of K1WG bases
__device__ __forceinline__ int wg_tile_step(int halo) { return halo <= K1WG / 2 ? K1WG - halo : K1WG; }

__device__ __forceinline__ bool wg_lds_usable(const K1Args& a) {
  if (a.k > 255) return false;
  if (a.mode == 2) return 2 * a.k - (int)a.w_or_s - 1 <= K1H && (int)a.w_or_s >= 1 && (int)a.w_or_s <= a.k;
  if (a.mode == 1) return (int)a.w_or_s >= 1 && (int)a.w_or_s + 1 < K1H;
  return true;
}

__device__ __forceinline__ int wg_sketch_mate_lds(const K1Args& a, const uint8_t* __restrict__ s, int len, const uint64_t* tab, K1Lds& L,
                                                  uint64_t* __restrict__ out, int cnt, int* s_wave, int tid) {
  const int k = a.k;
  const bool scaled = a.scaled != 0;
  const int nk = len - k + 1;  // k-mer positions
  if (nk <= 0) return cnt;
  if (a.mode == 0) {
    const int T = wg_tile_step(k - 1);
    for (int p0 = 0; p0 < nk; p0 += T) {
      wg_prefix(s + p0, min(len - p0, T + k - 1), tab, L, tid);
      const bool v = tid < T && p0 + tid < nk;
      const uint64_t h = v ? lds_hash(L, tid, k) : 0;
      cnt = wg_compact(v && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
    }
    return cnt;
  }
  if (a.mode == 2) {  // closed syncmer (see syncmer_mate)
    const int sm = (int)a.w_or_s, Lw = 2 * k - sm - 1;
    if (len < Lw) return cnt;
    const int wsz = 2 * (k - sm);
    const int nw = wsz > 0 ? len - Lw + 1 : nk;
    const int ns = len - sm + 1;
    const int T = wg_tile_step(Lw);
    for (int p0 = 0; p0 < nw; p0 += T) {
      wg_prefix(s + p0, min(len - p0, T + Lw), tab, L, tid);
      const int nst = min(ns - p0, T + max(wsz - 1, 0));
      for (int i = tid; i < nst; i += K1WG) L.hw[i] = lds_hash(L, i, sm);
      __syncthreads();
      const bool v = tid < T && p0 + tid < nw;
      uint64_t h = 0;
      if (v) {
        int pos = tid;
        if (wsz > 0) {
          const int m = argmin_left(L.hw, tid, wsz);
          pos = (m - tid < k - sm) ? m : m + sm - k;
        }
        h = lds_hash(L, pos, k);
      }
      cnt = wg_compact(v && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
    }
    return cnt;
  }
  // minimizer (see minimizer_mate)
  const int w = (int)a.w_or_s;
  if (len < k + w - 1) return cnt;
  const int nw = nk - w + 1;
  const int T = wg_tile_step(w + k);
  for (int p0 = 0; p0 < nw; p0 += T) {
    const int b0 = p0 > 0 ? p0 - 1 : 0, off = p0 - b0;  // the window before the tile's first one is needed too
    wg_prefix(s + b0, min(len - b0, T + w + k), tab, L, tid);
    const int nkt = min(nk - b0, T + w);
    for (int i = tid; i < nkt; i += K1WG) L.hw[i] = lds_hash(L, i, k);
    __syncthreads();
    const int w0 = p0 + tid;
    const bool v = tid < T && w0 < nw;
    int m = -1, pm = -2;
    if (v) {
      m = argmin_left(L.hw, off + tid, w);
      pm = w0 > 0 ? argmin_left(L.hw, off + tid - 1, w) : -2;
    }
    const uint64_t h = (v && m != pm) ? L.hw[m] : 0;
    cnt = wg_compact(v && m != pm && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
  }
  return cnt;
}

__global__ void __launch_bounds__(K1WG) k1_kmers_wg(const K1Args a) {
  __shared__ uint64_t tab[256];
  __shared__ int s_wave[K1WG / 64];
  __shared__ K1Lds lds;
  const int tid = threadIdx.x;
  if (tid < 256) tab[tid] = seed_of(tid);
  __syncthreads();
  const bool use_lds = wg_lds_usable(a);
  for (uint32_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
    const uint64_t o1 = a.offs[r];
    const int len1 = (int)(a.offs[r + 1] - o1);
    uint64_t o2 = 0;
    int len2 = 0;
    const bool pe = a.offs2 != nullptr;
    if (pe) {
      o2 = a.offs2[r];
      len2 = (int)(a.offs2[r + 1] - o2);
    }
    uint64_t* out = a.hashes + o1 + o2;
    const bool skip = len1 < a.min_qlen && !(pe && len2 >= a.min_qlen);
    int cnt = 0, cnt1 = 0;
    if (!skip) {
      uint64_t* tk = a.scratch ? a.scratch + o1 + o2 : nullptr;
      uint64_t* ts = a.scratch2 ? a.scratch2 + o1 + o2 : nullptr;
      cnt = use_lds ? wg_sketch_mate_lds(a, a.seqs + o1, len1, tab, lds, out, 0, s_wave, tid)
                    : wg_sketch_mate(a, a.seqs + o1, len1, tab, tk, ts, out, 0, s_wave, tid);
      cnt1 = cnt;
      if (pe)
        cnt = use_lds ? wg_sketch_mate_lds(a, a.seqs2 + o2, len2, tab, lds, out, cnt, s_wave, tid)
                      : wg_sketch_mate(a, a.seqs2 + o2, len2, tab, tk, ts, out, cnt, s_wave, tid);
    }
    if (tid == 0) {
      a.nk_raw[r] = cnt;
      a.nk1[r] = cnt1;
      a.qlen[r] = len1 + len2;
    }
  }
}

// ---- whole genomes (plain / FracMinHash k-mers): segments of K1SEG positions on their own workgroups ----------------
// Pass 1 hashes a segment and compacts its kept hashes at scratch[offs[r] + seg*K1SEG ...]; pass 2 moves the segments of a
// read together in order (destination = sum of the counts of the earlier segments).
constexpr int K1SEG = 65536;

__global__ void __launch_bounds__(K1WG) k1_seg_hash(const K1Args a) {
  __shared__ uint64_t tab[256];
  __shared__ int s_wave[K1WG / 64];
  __shared__ K1Lds lds;
  const int tid = threadIdx.x;
  if (tid < 256) tab[tid] = seed_of(tid);
  __syncthreads();
  const uint32_t r = blockIdx.x / a.segs_max, seg = blockIdx.x % a.segs_max;
  const uint64_t o1 = a.offs[r];
  const int len = (int)(a.offs[r + 1] - o1);
  const int npos = len - a.k + 1;
  const int p_lo = (int)seg * K1SEG;
  int cnt = 0;
  if (len >= a.min_qlen && p_lo < npos) {  // (:778-786 gate; ErrShortSeq => no k-mers)
    const uint8_t* __restrict__ s = a.seqs + o1;
    uint64_t* __restrict__ out = a.scratch + o1 + p_lo;
    const int p_hi = min(npos, p_lo + K1SEG);
    const bool scaled = a.scaled != 0;
    const int T = wg_tile_step(a.k - 1);
    for (int p0 = p_lo; p0 < p_hi; p0 += T) {
      wg_prefix(s + p0, min(len - p0, T + a.k - 1), tab, lds, tid);
      const bool v = tid < T && p0 + tid < p_hi;
      const uint64_t h = v ? lds_hash(lds, tid, a.k) : 0;
      cnt = wg_compact(v && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
    }
  }
  if (tid == 0) a.seg_cnt[blockIdx.x] = cnt;
}

__global__ void __launch_bounds__(256) k1_seg_pack(const K1Args a) {
  const uint32_t r = blockIdx.x / a.segs_max, seg = blockIdx.x % a.segs_max;
  const uint64_t o1 = a.offs[r];
  const int len = (int)(a.offs[r + 1] - o1);
  const int npos = len - a.k + 1;
  const int nsegs = npos > 0 ? (npos + K1SEG - 1) / K1SEG : 1;
  if ((int)seg >= nsegs) return;
  const int* __restrict__ sc = a.seg_cnt + (size_t)r * a.segs_max;
  int dest = 0;
  for (uint32_t t = 0; t < seg; t++) dest += sc[t];
  const int cnt = sc[seg];
  const uint64_t* __restrict__ src = a.scratch + o1 + (uint64_t)seg * K1SEG;
  uint64_t* __restrict__ dst = a.hashes + o1 + dest;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) dst[i] = src[i];
  if ((int)seg == nsegs - 1 && threadIdx.x == 0) {
    a.nk_raw[r] = dest + cnt;
    a.nk1[r] = dest + cnt;
    a.qlen[r] = len;
  }
}

// NumKmers when no read of the batch can exceed the dedup threshold.
__global__ void k_nk_simple(const int32_t* nk_raw, int32_t* nk_search, uint32_t n, int32_t min_matched) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int v = nk_raw[i];
    nk_search[i] = v >= min_matched ? v : 0;  // :854-869: too few k-mers => not searched
  }
}

// ------------------------------------------------------------------------------------------------
// K1d: sort + in-place unique (handleQuery :874-908).  One workgroup per read.  Bitonic network in
// its all-ascending form, so indices >= n act as +inf without being stored.
// ------------------------------------------------------------------------------------------------
template <typename P>
__device__ __forceinline__ void bitonic_sort(P a, int n, int tid, int nthreads) {
  int lg = 0;
  while ((1 << lg) < n) lg++;
  const int half = (1 << lg) >> 1;  // compare-exchanges per stage
  for (int ls = 1; ls <= lg; ls++) {
    const int size = 1 << ls, hm = (size >> 1) - 1;
    // flip: i = blk + t, j = blk + size-1-t
    for (int p = tid; p < half; p += nthreads) {
      const int blk = (p >> (ls - 1)) << ls, t = p & hm;
      const int i = blk + t, j = blk + size - 1 - t;
      if (j < n) {
        uint64_t x = a[i], y = a[j];
        if (x > y) { a[i] = y; a[j] = x; }
      }
    }
    __syncthreads();
    for (int lt = ls - 2; lt >= 0; lt--) {
      const int stride = 1 << lt;
      for (int p = tid; p < half; p += nthreads) {
        const int i = ((p >> lt) << (lt + 1)) + (p & (stride - 1)), j = i + stride;
        if (j < n) {
          uint64_t x = a[i], y = a[j];
          if (x > y) { a[i] = y; a[j] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// inclusive +scan over the 64 lanes of a wave (all lanes active), see wave_xor_scan32
__device__ __forceinline__ int wave_add_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// order-preserving removal of adjacent repeats: src[0..n) -> dst (a different array); returns the new length (uniform).
// scan: NT/64 + 1 ints of LDS.
template <int NT, typename P>
__device__ __forceinline__ int block_unique(P src, int n, uint64_t* __restrict__ dst, int* scan, int tid) {
  const int chunk = (n + NT - 1) / NT;
  const int b = min(n, tid * chunk), e = min(n, b + chunk);
  int c = 0;
  for (int i = b; i < e; i++) c += (i == 0 || src[i] != src[i - 1]) ? 1 : 0;
  const int incl = wave_add_scan(c);
  __syncthreads();  // scan[] may still be read by a previous call
  if ((tid & 63) == 63) scan[tid >> 6] = incl;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    const int t = scan[w];
    if (w < (tid >> 6)) before += t;
    total += t;
  }
  int pos = before + incl - c;
  for (int i = b; i < e; i++)
    if (i == 0 || src[i] != src[i - 1]) dst[pos++] = src[i];
  __syncthreads();
  return total;
}

// Queries of at most 512 k-mers (paired-end 2x150 / 2x250 reads just above -u 256): one WAVE per query, 8 elements per lane
// in registers (element e = lane*8 + i).  The same all-ascending bitonic network: compare-exchanges at distance < 8 are
// register moves, the others exchange registers with lane ^ mask (ds_bpermute); no LDS array, no barrier.
constexpr int DW_CAP = 512;

__device__ __forceinline__ void cx64(uint64_t& lo, uint64_t& hi) {
  if (lo > hi) {
    const uint64_t t = lo;
    lo = hi;
    hi = t;
  }
}

__global__ void __launch_bounds__(256) k_dedup_wave(const DedupArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.n_reads) return;
  const int n = a.nk_raw[r];
  if (n <= a.dedup_threshold) {
    if (lane == 0) a.nk_search[r] = n >= a.min_matched ? n : 0;
    return;
  }
  if (n > DW_CAP) return;  // the workgroup classes take it
  uint64_t* __restrict__ g = a.hashes + a.offs[r] + (a.offs2 ? a.offs2[r] : 0);
  uint64_t v[8], p[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int e = lane * 8 + i;
    v[i] = e < n ? g[e] : ~0ULL;  // pads sort to the end; only the first n sorted elements are looked at
  }
#pragma unroll
  for (int S = 2; S <= DW_CAP; S <<= 1) {
    // flip: e <-> e ^ (S-1)
    if (S <= 8) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        if ((i ^ (S - 1)) > i) cx64(v[i], v[i ^ (S - 1)]);
    } else {
      const int mask = S / 8 - 1;
      const bool lower = (lane & (S / 16)) == 0;  // lane < lane ^ mask
#pragma unroll
      for (int i = 0; i < 8; i++) p[i] = __shfl_xor(v[7 - i], mask);
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = lower ? (v[i] < p[i] ? v[i] : p[i]) : (v[i] > p[i] ? v[i] : p[i]);
    }
#pragma unroll
    for (int d = S / 4; d >= 1; d >>= 1) {  // e <-> e ^ d
      if (d < 8) {
#pragma unroll
        for (int i = 0; i < 8; i++)
          if ((i & d) == 0) cx64(v[i], v[i | d]);
      } else {
        const int mask = d / 8;
        const bool lower = (lane & mask) == 0;
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = __shfl_xor(v[i], mask);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = lower ? (v[i] < p[i] ? v[i] : p[i]) : (v[i] > p[i] ? v[i] : p[i]);
      }
    }
  }
  // unique, in place (every element is in a register by now)
  const uint64_t prev_lane = __shfl_up(v[7], 1);
  int c = 0;
  bool f[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int e = lane * 8 + i;
    const uint64_t prev = i == 0 ? prev_lane : v[i - 1];
    f[i] = e < n && (e == 0 || v[i] != prev);
    c += f[i] ? 1 : 0;
  }
  const int incl = wave_add_scan(c);
  int pos = incl - c;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (f[i]) g[pos++] = v[i];
  const int total = __builtin_amdgcn_readlane(incl, 63);
  // MinMatched is tested on the raw count (:854), NumKmers is the unique count (:910)
  if (lane == 0) a.nk_search[r] = n >= a.min_matched ? total : 0;
}

// Window sketches emit the same k-mer for runs of consecutive windows (10 k syncmer emissions of a HiFi read hold ~1.4 k
// distinct adjacent values), so for such databases an order-preserving pass drops adjacent repeats first: hashes -> scratch,
// the shortened length parked in nk_search[r] until the sort of that query overwrites it with NumKmers.
constexpr int ADJ_NT = 512;
__global__ void __launch_bounds__(ADJ_NT) k_adj_unique(const DedupArgs a) {
  __shared__ int s_wave[ADJ_NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (uint32_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
  const int n = a.nk_raw[r];
  if (n <= a.dedup_threshold || n <= a.n_lo || n > a.n_hi) continue;
  const uint64_t koff = a.offs[r] + (a.offs2 ? a.offs2[r] : 0);
  const uint64_t* __restrict__ g = a.hashes + koff;
  uint64_t* __restrict__ dst = a.scratch + koff;
  int m = 0;
  for (int t0 = 0; t0 < n; t0 += ADJ_NT) {  // tiles of consecutive elements: coalesced reads, ordered compaction
    const int i = t0 + tid;
    uint64_t x = 0;
    bool keep = false;
    if (i < n) {
      x = g[i];
      keep = i == 0 || x != g[i - 1];
    }
    const uint64_t mask = __ballot(keep);
    if (lane == 0) s_wave[w] = __popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int j = 0; j < ADJ_NT / 64; j++) {
      const int c = s_wave[j];
      if (j < w) before += c;
      total += c;
    }
    if (keep) dst[m + before + __popcll(mask & ((1ULL << lane) - 1ULL))] = x;
    m += total;
    __syncthreads();
  }
  if (tid == 0) a.nk_search[r] = m;
  }
}

// Workgroup classes for queries of more than 512 k-mers (raw count in (n_lo, n_hi]), chosen by the number m of elements
// left to sort: m <= 4096 in 32 KB of LDS with 256 threads; above that 1024 threads and 128 KB of LDS (m <= 16384), beyond
// that the network runs in global memory.
template <int NT, int CAP>
__global__ void __launch_bounds__(NT) k_dedup(const DedupArgs a) {
  __shared__ uint64_t s[CAP];
  __shared__ int scan[NT / 64];
  const int tid = threadIdx.x;
  for (uint32_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
  const int n = a.nk_raw[r];
  if (n <= a.dedup_threshold || n <= a.n_lo || n > a.n_hi) continue;  // settled by k_dedup_wave / sorted by sort_huge.hip
  const int m = a.pre ? a.nk_search[r] : n;  // (a query the smaller class finished shows its NumKmers <= 4096 here)
  if (m <= a.lo || m > a.hi) continue;
  const uint64_t koff = a.offs[r] + (a.offs2 ? a.offs2[r] : 0);
  uint64_t* g = a.hashes + koff;
  uint64_t* tmp = a.scratch + koff;
  const uint64_t* in = a.pre ? tmp : g;
  int total;
  if (m <= CAP) {
    for (int i = tid; i < m; i += NT) s[i] = in[i];
    __syncthreads();
    bitonic_sort(s, m, tid, NT);
    total = block_unique<NT>(s, m, g, scan, tid);
  } else {
    uint64_t* w = const_cast<uint64_t*>(in);
    if (in == g) {  // sort a copy so that the result can be compacted back into g
      for (int i = tid; i < m; i += NT) tmp[i] = g[i];
      __threadfence_block();
      __syncthreads();
      w = tmp;
    }
    bitonic_sort(w, m, tid, NT);
    __threadfence_block();
    total = block_unique<NT>(w, m, g, scan, tid);
  }
  // MinMatched is tested on the raw count (:854), NumKmers is the unique count (:910)
  if (tid == 0) a.nk_search[r] = n >= a.min_matched ? total : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// K2: the COBS query.
//
// Work unit = (read, slot) with slot = (resident block, tile of LPR*16 bytes of its rows).  LPR lanes
// serve one unit, so a wave carries G = 64/LPR units: LPR = 64 for wide rows (GTDB-scale, 1872 B),
// 16 or 4 for narrow rows (a 312-column block has 39-byte rows).  Each lane owns 16 bytes = 128
// columns of its unit's rows and keeps their match counts as NPL bit-sliced planes (vertical
// counters): 8 rows are reduced with a carry-save adder tree (7 CSAs) and the carry word rippled into
// the upper planes, ~4 VALU ops per loaded dword, which keeps the kernel HBM-bound.
// Row indices of a chunk of CH k-mers are computed cooperatively (one exact fastmod per (k-mer,
// block)) into a per-wave LDS table; k-mers past the end of a read map to the all-zero row appended to
// each block, so the inner loop has no tail code.
// ------------------------------------------------------------------------------------------------
#define CSA(h, l, a_, b_, c_)              \
  {                                        \
    uint32_t u_ = (a_) ^ (b_);             \
    h = ((a_) & (b_)) | (u_ & (c_));       \
    l = u_ ^ (c_);                         \
  }

// 16 bytes of a row.  Index rows are read once and never reused: non-temporal loads keep them from displacing the
// hash/offset lines in L2 (+2 % on the random-gather microbenchmark, profiles/).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_row16(const uint8_t* p, int nt) {
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
  const u32x4 v = nt ? __builtin_nontemporal_load(q) : *q;
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <int NPL>
__device__ __forceinline__ void csa8(uint32_t (&pl)[NPL], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4,
                                     uint32_t x5, uint32_t x6, uint32_t x7) {
  uint32_t ta, tb, fa, fb, e;
  CSA(ta, pl[0], pl[0], x0, x1);
  CSA(tb, pl[0], pl[0], x2, x3);
  CSA(fa, pl[1], pl[1], ta, tb);
  CSA(ta, pl[0], pl[0], x4, x5);
  CSA(tb, pl[0], pl[0], x6, x7);
  CSA(fb, pl[1], pl[1], ta, tb);
  CSA(e, pl[2], pl[2], fa, fb);
#pragma unroll
  for (int p = 3; p < NPL; p++) {
    uint32_t t = pl[p] & e;
    pl[p] ^= e;
    e = t;
  }
}

// SPLIT = true is the long-query form: a unit is (long query, slot, chunk of a.split_chk <= 8192 k-mers); its counts are added to a
// per-query u32 array with atomics and thresholded by k_threshold_long, so a whole genome spreads over the chip instead
// of one wave per (query, slot).

template <int LPR, int NPL, bool MULTI, bool SPLIT>
__global__ void __launch_bounds__(256) k2_cobs(const K2Args a) {
  constexpr int G = 64 / LPR;
  constexpr int PAIRS = MULTI ? 256 : 1024;
  constexpr int CH = (PAIRS / G) > 64 ? 64 : (PAIRS / G);
  constexpr int NHMAX = MULTI ? 4 : 1;
  static_assert(CH % 8 == 0, "chunk must be a multiple of the CSA group");
  __shared__ uint32_t s_rows[4][NHMAX][G * CH];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, li = lane % LPR;
  const uint64_t per_read = SPLIT ? (uint64_t)a.nslots * a.split_chunks : (uint64_t)a.nslots;
  const uint64_t total_units = (SPLIT ? (uint64_t)a.n_long : (uint64_t)a.n_reads) * per_read;
  const uint64_t u = a.unit_base + ((uint64_t)blockIdx.x * 4 + wave) * G + g;
  const bool valid = u < total_units;
  uint32_t r = 0, sidx = 0, li_long = 0;
  int k0 = 0;
  if (valid) {
    if (SPLIT) {
      li_long = (uint32_t)(u / per_read);
      const uint32_t rem = (uint32_t)(u % per_read);
      sidx = rem / a.split_chunks;
      k0 = (int)(rem % a.split_chunks) * (int)a.split_chk;
      r = a.long_list[li_long];
    } else {
      r = (uint32_t)(u / a.nslots);
      sidx = (uint32_t)(u % a.nslots);
    }
  }
  const Slot slot = a.slots[sidx];
  const BlockDev* __restrict__ bd = a.blocks + slot.block;
  int n = valid ? a.nk[r] : 0;
  if (SPLIT) n = max(0, min(n - k0, (int)a.split_chk));
  else if (a.split_min > 0 && n > a.split_min) n = 0;  // long queries are left to the SPLIT launch
  int nmax = n;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
  if (nmax == 0) return;

  const uint32_t stride = bd->stride;
  const uint32_t boff = (slot.tile * LPR + li) * 16u;
  const bool active = n > 0 && boff < stride;
  // integer threshold (:7468-7470): count >= minMatched && float64(count) > nHashes*queryCov
  const double thr = __dmul_rn((double)n, a.min_qcov);
  uint32_t cmin = (uint32_t)thr + 1u;  // smallest integer c with (double)c > thr  (thr >= 0)
  if (cmin < (uint32_t)a.min_matched) cmin = (uint32_t)a.min_matched;
  // Branch and bound: once count + (k-mers still to come) < cmin for every column of a 128-byte sector of the row, nothing
  // in it can become a hit any more and its lanes stop loading.  Unrelated references are dead after ~80 % of a read's
  // k-mers (Bloom density <= fpr), so the tail of the row traffic is never fetched; results are unchanged.
  constexpr int GRP = LPR < 8 ? LPR : 8;  // lanes that share a sector
  bool live = active;
  const uint8_t* __restrict__ base = bd->rows + boff;
  const uint64_t koff = a.offs[r] + (a.offs2 ? a.offs2[r] : 0) + (uint64_t)k0;
  const int nh = MULTI ? a.num_hashes : 1;

  uint32_t pl[4][NPL];
#pragma unroll
  for (int d = 0; d < 4; d++)
#pragma unroll
    for (int p = 0; p < NPL; p++) pl[d][p] = 0;

  for (int c0 = 0; c0 < nmax; c0 += CH) {
    // ---- row indices of this chunk: loc = h % NumSigs (:6811), multi-hash h_i = uint32(a + b*i) (util-hash.go:125-142)
    for (int p = lane; p < G * CH; p += 64) {
      const int q = p / CH, j = p % CH;
      const int srcl = q * LPR;
      const int nq = __shfl(n, srcl);
      const uint64_t koq = __shfl((unsigned long long)koff, srcl);
      const uint32_t bi = __shfl(slot.block, srcl);
      const BlockDev* __restrict__ bq = a.blocks + bi;
      const uint64_t ns = bq->num_sigs;
      const uint64_t s16 = bq->stride >> 4;  // rows are addressed in 16-byte units: 32 bits reach 64 GB per block
      const int kidx = c0 + j;
      if (kidx < nq) {
        const uint64_t h = a.hashes[koq + kidx];
        if (!MULTI) {
          s_rows[wave][0][p] = (uint32_t)(fastmod_u64(h, ns, bq->magic_hi, bq->magic_lo) * s16);
        } else {
          const uint32_t ha = (uint32_t)(h >> 32), hb = (uint32_t)h;
          for (int i = 0; i < nh; i++)
            s_rows[wave][i][p] = (uint32_t)(fastmod_u64((uint64_t)(uint32_t)(ha + hb * (uint32_t)i), ns, bq->magic_hi, bq->magic_lo) * s16);
        }
      } else {
        for (int i = 0; i < nh; i++) s_rows[wave][i][p] = (uint32_t)(ns * s16);  // the appended all-zero row
      }
    }
    wave_lds_fence();

    const int cnt = min(CH, nmax - c0);
    for (int j = 0; j < cnt; j += 8) {
      uint4 x[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (live) {
          const uint32_t row = s_rows[wave][0][g * CH + j + i];
          v = load_row16(base + ((uint64_t)row << 4), a.nt_loads);
          if (MULTI) {
            for (int hh = 1; hh < nh; hh++) {  // AND of the h rows (pand.AndUnsafe, :6639-6646)
              const uint32_t row2 = s_rows[wave][hh][g * CH + j + i];
              const uint4 w = load_row16(base + ((uint64_t)row2 << 4), a.nt_loads);
              v.x &= w.x; v.y &= w.y; v.z &= w.z; v.w &= w.w;
            }
          }
        }
        x[i] = v;
      }
      csa8<NPL>(pl[0], x[0].x, x[1].x, x[2].x, x[3].x, x[4].x, x[5].x, x[6].x, x[7].x);
      csa8<NPL>(pl[1], x[0].y, x[1].y, x[2].y, x[3].y, x[4].y, x[5].y, x[6].y, x[7].y);
      csa8<NPL>(pl[2], x[0].z, x[1].z, x[2].z, x[3].z, x[4].z, x[5].z, x[6].z, x[7].z);
      csa8<NPL>(pl[3], x[0].w, x[1].w, x[2].w, x[3].w, x[4].w, x[5].w, x[6].w, x[7].w);
      if (!SPLIT && a.prune) {
        const int done = min(n, c0 + j + 8);
        const int need = (int)cmin - (n - done);  // a column must already hold this many to stay in the race
        bool lane_alive = live;
        if (live && need > 0) {
          uint32_t any = 0;
          if (NPL >= 32 || ((uint32_t)need >> NPL) == 0) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
              uint32_t ge = 0xffffffffu;
#pragma unroll
              for (int p = 0; p < NPL; p++) ge = (((uint32_t)need >> p) & 1u) ? (ge & pl[d][p]) : (ge | pl[d][p]);
              any |= ge;
            }
          }
          lane_alive = any != 0;
        }
        const uint64_t alive = __ballot(lane_alive);
        live = live && ((alive >> (lane & ~(GRP - 1))) & ((1ULL << GRP) - 1ULL)) != 0;
        if (alive == 0) break;  // the whole wave is done with these rows
      }
    }
    wave_lds_fence();
    if (!SPLIT && a.prune && __ballot(live) == 0) break;
  }

  if (!live) return;
  if (SPLIT) {
    // partial counts of this chunk -> the query's count array (consecutive lanes hit consecutive words)
    uint32_t* __restrict__ acc = a.long_counts + (uint64_t)li_long * a.ncols_total + bd->col_base;
#pragma unroll
    for (int d = 0; d < 4; d++) {
      for (int q = 0; q < 32; q++) {
        uint32_t count = 0;
#pragma unroll
        for (int p = 0; p < NPL; p++) count |= ((pl[d][p] >> q) & 1u) << p;
        const uint32_t col = (boff + (uint32_t)d * 4u + (uint32_t)(q >> 3)) * 8u + (7u - (uint32_t)(q & 7));
        if (count && col < bd->ncols) atomicAdd(acc + col, count);
      }
    }
    return;
  }
  if (NPL < 32 && (cmin >> NPL) != 0) return;  // unreachable count
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t ge = 0xffffffffu;  // bit-sliced (count >= cmin), LSB to MSB
#pragma unroll
    for (int p = 0; p < NPL; p++) ge = ((cmin >> p) & 1u) ? (ge & pl[d][p]) : (ge | pl[d][p]);
    while (ge) {
      const int q = __ffs(ge) - 1;
      ge &= ge - 1;
      uint32_t count = 0;
#pragma unroll
      for (int p = 0; p < NPL; p++) count |= ((pl[d][p] >> q) & 1u) << p;
      // byte (q>>3) of this dword, bit (q&7): bit 7 = first column of the byte (index.go:1157)
      const uint32_t col = (boff + (uint32_t)d * 4u + (uint32_t)(q >> 3)) * 8u + (7u - (uint32_t)(q & 7));
      if (col < bd->ncols) {
        const unsigned long long idx = atomicAdd(a.counter, 1ULL);
        if (idx < a.hit_cap) {
          kmcpg_hit hit;
          hit.read = r;
          hit.col = bd->col_base + col;
          hit.count = count;
          a.hits[idx] = hit;
        }
      }
    }
  }
}

constexpr uint64_t K2_MAX_BLOCKS = 1ull << 23;  // x 256 threads = 2^31

template <int LPR, int NPL>
static void launch_k2_t(const K2Args& a, bool multi, hipStream_t st) {
  constexpr int G = 64 / LPR;
  const uint64_t units = (uint64_t)a.n_reads * a.nslots;
  const uint64_t waves = (units + G - 1) / G;
  const uint64_t blocks = (waves + 3) / 4;
  K2Args b = a;
  for (uint64_t b0 = 0; b0 < blocks; b0 += K2_MAX_BLOCKS) {  // a launch holds fewer than 2^32 threads
    const unsigned nb = (unsigned)std::min<uint64_t>(K2_MAX_BLOCKS, blocks - b0);
    b.unit_base = b0 * 4 * G;
    if (multi)
      hipLaunchKernelGGL((k2_cobs<LPR, NPL, true, false>), dim3(nb), dim3(256), 0, st, b);
    else
      hipLaunchKernelGGL((k2_cobs<LPR, NPL, false, false>), dim3(nb), dim3(256), 0, st, b);
  }
}

template <int LPR>
static int launch_k2_l(const K2Args& a, int npl, bool multi, hipStream_t st) {
  switch (npl) {
    case 8: launch_k2_t<LPR, 8>(a, multi, st); return 0;
    case 16: launch_k2_t<LPR, 16>(a, multi, st); return 0;
    case 24: launch_k2_t<LPR, 24>(a, multi, st); return 0;
    default: return -1;
  }
}

int launch_k2(const K2Args& a, int lpr, int npl, hipStream_t st) {
  const bool multi = a.num_hashes > 1;
  switch (lpr) {
    case 4: return launch_k2_l<4>(a, npl, multi, st);
    case 16: return launch_k2_l<16>(a, npl, multi, st);
    case 64: return launch_k2_l<64>(a, npl, multi, st);
    default: return -1;
  }
}

template <int LPR>
static void launch_k2_split_t(const K2Args& a, bool multi, hipStream_t st) {
  constexpr int G = 64 / LPR;
  const uint64_t units = (uint64_t)a.n_long * a.nslots * a.split_chunks;
  const uint64_t blocks = ((units + G - 1) / G + 3) / 4;
  K2Args b = a;
  for (uint64_t b0 = 0; b0 < blocks; b0 += K2_MAX_BLOCKS) {
    const unsigned nb = (unsigned)std::min<uint64_t>(K2_MAX_BLOCKS, blocks - b0);
    b.unit_base = b0 * 4 * G;
    if (multi)
      hipLaunchKernelGGL((k2_cobs<LPR, 16, true, true>), dim3(nb), dim3(256), 0, st, b);
    else
      hipLaunchKernelGGL((k2_cobs<LPR, 16, false, true>), dim3(nb), dim3(256), 0, st, b);
  }
}

int launch_k2_split(const K2Args& a, int lpr, hipStream_t st) {
  const bool multi = a.num_hashes > 1;
  switch (lpr) {
    case 4: launch_k2_split_t<4>(a, multi, st); return 0;
    case 16: launch_k2_split_t<16>(a, multi, st); return 0;
    case 64: launch_k2_split_t<64>(a, multi, st); return 0;
    default: return -1;
  }
}

// queries with more than split_min k-mers: meta[0] = how many, meta[1] = their largest NumKmers
__global__ void k_list_long(const int32_t* __restrict__ nk, uint32_t n_reads, int32_t split_min, uint32_t* __restrict__ list, uint32_t* __restrict__ meta) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_reads && nk[r] > split_min) {
    list[atomicAdd(&meta[0], 1u)] = r;
    atomicMax(&meta[1], (uint32_t)nk[r]);
  }
}

void launch_list_long(const int32_t* nk, uint32_t n_reads, int32_t split_min, uint32_t* list, uint32_t* meta, hipStream_t st) {
  if (n_reads == 0) return;
  hipLaunchKernelGGL(k_list_long, dim3((n_reads + 255) / 256), dim3(256), 0, st, nk, n_reads, split_min, list, meta);
}

// threshold over the accumulated counts of the long queries (same integer rule as the k2_cobs epilogue)
__global__ void k_threshold_long(const K2Args a) {
  const uint64_t total = (uint64_t)a.n_long * a.ncols_total;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = a.long_counts[i];
    if (c == 0) continue;
    const uint32_t li = (uint32_t)(i / a.ncols_total), col = (uint32_t)(i % a.ncols_total);
    const uint32_t r = a.long_list[li];
    const double thr = __dmul_rn((double)a.nk[r], a.min_qcov);
    uint32_t cmin = (uint32_t)thr + 1u;
    if (cmin < (uint32_t)a.min_matched) cmin = (uint32_t)a.min_matched;
    if (c >= cmin) {
      const unsigned long long idx = atomicAdd(a.counter, 1ULL);
      if (idx < a.hit_cap) {
        kmcpg_hit hit;
        hit.read = r;
        hit.col = col;
        hit.count = c;
        a.hits[idx] = hit;
      }
    }
  }
}

void launch_threshold_long(const K2Args& a, hipStream_t st) {
  const uint64_t total = (uint64_t)a.n_long * a.ncols_total;
  if (total == 0) return;
  unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(k_threshold_long, dim3(blocks), dim3(256), 0, st, a);
}

int k1_segment_len() { return K1SEG; }

void launch_k1(const K1Args& a, uint32_t max_read_len, hipStream_t st) {
  if (a.n_reads == 0) return;
  if (a.seg_cnt && a.segs_max > 1) {  // whole genomes: one workgroup per 65536-position segment, then an ordered pack
    const unsigned blocks = a.n_reads * a.segs_max;
    hipLaunchKernelGGL(k1_seg_hash, dim3(blocks), dim3(K1WG), 0, st, a);
    hipLaunchKernelGGL(k1_seg_pack, dim3(blocks), dim3(256), 0, st, a);
    return;
  }
  if (max_read_len > 2048) {  // long queries: a whole workgroup per read
    unsigned blocks = a.n_reads > 65536 ? 65536 : a.n_reads;
    hipLaunchKernelGGL(k1_kmers_wg, dim3(blocks), dim3(K1WG), 0, st, a);
    return;
  }
  unsigned blocks = (a.n_reads + 3) / 4;
  if (blocks > 32768) blocks = 32768;
  hipLaunchKernelGGL(k1_kmers, dim3(blocks), dim3(256), 0, st, a);
}

void launch_nk_simple(const int32_t* nk_raw, int32_t* nk_search, uint32_t n, int32_t min_matched, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_nk_simple, dim3((n + 255) / 256), dim3(256), 0, st, nk_raw, nk_search, n, min_matched);
}

void launch_dedup(DedupArgs a, uint64_t max_n, hipStream_t st) {
  if (a.n_reads == 0) return;
  // the wave class settles every query at or below the dedup threshold and sorts those of at most 512 k-mers
  hipLaunchKernelGGL(k_dedup_wave, dim3((a.n_reads + 3) / 4), dim3(256), 0, st, a);
  if (max_n <= DW_CAP) return;
  a.n_lo = DW_CAP;
  a.n_hi = max_n > HUGE_MIN ? (int32_t)HUGE_MIN : 0x7fffffff;  // beyond that: device-wide sort (sort_huge.hip)
  const unsigned grid = a.n_reads > (1u << 20) ? (1u << 20) : a.n_reads;  // the workgroup kernels stride over the reads
  if (a.pre) hipLaunchKernelGGL(k_adj_unique, dim3(grid), dim3(ADJ_NT), 0, st, a);
  a.lo = 0;
  a.hi = 4096;
  hipLaunchKernelGGL((k_dedup<256, 4096>), dim3(grid), dim3(256), 0, st, a);
  if (max_n > 4096) {
    a.lo = 4096;
    a.hi = 0x7fffffff;
    hipLaunchKernelGGL((k_dedup<1024, 16384>), dim3(grid), dim3(1024), 0, st, a);
  }
}

// ------------------------------------------------------------------------------------------------
// layout: on-disk rows (NumRowBytes, unpadded — serialization.go:140,379) -> HBM rows (stride)
// ------------------------------------------------------------------------------------------------
__global__ void k_repack(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n_rows, uint32_t row_bytes,
                         uint32_t stride) {
  const uint32_t wpr = stride / 4;  // dwords per dst row
  const uint64_t total = n_rows * wpr;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t row = i / wpr;
    const uint32_t b0 = (uint32_t)(i % wpr) * 4;
    uint32_t v = 0;
    const uint8_t* s = src + row * row_bytes;
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (b0 + t < row_bytes) v |= (uint32_t)s[b0 + t] << (8 * t);
    reinterpret_cast<uint32_t*>(dst + row * stride)[b0 / 4] = v;
  }
}

void launch_repack(const uint8_t* src, uint8_t* dst, uint64_t n_rows, uint32_t row_bytes, uint32_t stride, hipStream_t st) {
  if (n_rows == 0) return;
  uint64_t total = n_rows * (stride / 4);
  unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(k_repack, dim3(blocks), dim3(256), 0, st, src, dst, n_rows, row_bytes, stride);
}

__global__ void k_gather_rows(const uint8_t* __restrict__ rows, uint32_t stride, uint32_t row_bytes, const uint64_t* __restrict__ idx,
                              uint64_t n, uint8_t* __restrict__ out) {
  const uint64_t total = n * row_bytes;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / row_bytes;
    const uint32_t b = (uint32_t)(i % row_bytes);
    out[i] = rows[idx[r] * stride + b];
  }
}

void launch_gather_rows(const uint8_t* rows, uint32_t stride, uint32_t row_bytes, const uint64_t* idx, uint64_t n, uint8_t* out,
                        hipStream_t st) {
  if (n == 0) return;
  uint64_t total = n * row_bytes;
  unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, st, rows, stride, row_bytes, idx, n, out);
}

// ------------------------------------------------------------------------------------------------
// synthetic index (bench / full-size parity only): i.i.d. Bernoulli bits from a counter-based generator
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}

// 64 Bernoulli(p8/256) bits for counter c
__device__ __forceinline__ uint64_t bernoulli64(uint64_t key, uint64_t c, uint32_t p8) {
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {  // LSB of p8 first
    const uint64_t rnd = splitmix64(key ^ (c * 8 + i) * 0xd6e8feb86659fd93ULL);
    acc = ((p8 >> i) & 1u) ? (acc | rnd) : (acc & rnd);
  }
  return acc;
}

__global__ void k_synth_fill(uint8_t* __restrict__ rows, uint64_t n_rows, uint32_t stride, uint32_t ncols, uint64_t key, uint32_t p8) {
  const uint32_t qpr = stride / 8;  // qwords per row
  const uint64_t total = n_rows * qpr;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t row = i / qpr;
    const uint32_t qw = (uint32_t)(i % qpr);
    uint64_t v = bernoulli64(key, i, p8);
    // zero the bits of columns >= ncols (padding columns never match: :7466 scans them but count is 0)
    const uint32_t col0 = qw * 64;
    if (col0 >= ncols) v = 0;
    else if (col0 + 64 > ncols) {
      uint64_t m = 0;
      for (uint32_t c = col0; c < ncols; c++) {
        const uint32_t byte = (c - col0) >> 3, bit = 7 - ((c - col0) & 7);
        m |= 1ULL << (byte * 8 + bit);
      }
      v &= m;
    }
    reinterpret_cast<uint64_t*>(rows + row * stride)[qw] = v;
  }
}

void launch_synth_fill(uint8_t* rows, uint64_t n_rows, uint32_t stride, uint32_t ncols, uint64_t key, uint32_t p8, hipStream_t st) {
  uint64_t total = n_rows * (stride / 8);
  unsigned blocks = (unsigned)((total + 255) / 256 > 262144 ? 262144 : (total + 255) / 256);
  hipLaunchKernelGGL(k_synth_fill, dim3(blocks), dim3(256), 0, st, rows, n_rows, stride, ncols, key, p8);
}

// sigs[h % NumSigs] |= 1 << (7 - col%8)  (index.go:1157) for a list of hashes
__global__ void k_plant(BlockDev bd, uint32_t col, int num_hashes, const uint64_t* __restrict__ hashes, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h = hashes[i];
    const uint32_t ha = (uint32_t)(h >> 32), hb = (uint32_t)h;
    for (int t = 0; t < num_hashes; t++) {
      const uint64_t hv = num_hashes == 1 ? h : (uint64_t)(uint32_t)(ha + hb * (uint32_t)t);
      const uint64_t row = fastmod_u64(hv, bd.num_sigs, bd.magic_hi, bd.magic_lo);
      const uint64_t byte = row * bd.stride + (col >> 3);
      uint32_t* w = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(bd.rows) + (byte & ~3ULL));
      atomicOr(w, (uint32_t)(1u << (7 - (col & 7))) << (8 * (byte & 3)));
    }
  }
}

void launch_plant(const BlockDev& bd, uint32_t col, int num_hashes, const uint64_t* hashes, uint64_t n, hipStream_t st) {
  if (n == 0) return;
  unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(k_plant, dim3(blocks), dim3(256), 0, st, bd, col, num_hashes, hashes, n);
}

// plant every k-mer of read r into global column cols[r] (bench / full-size parity only)
__global__ void __launch_bounds__(256) k_plant_reads(const BlockDev* __restrict__ blocks, uint32_t nblocks, int num_hashes,
                                                     const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offs,
                                                     const int32_t* __restrict__ nk, const uint32_t* __restrict__ cols, uint32_t n_reads) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n_reads; r += nwaves) {
    const uint32_t cg = cols[r];
    if (cg == 0xffffffffu) continue;
    uint32_t bi = nblocks;
    for (uint32_t b = 0; b < nblocks; b++)
      if (cg >= blocks[b].col_base && cg < blocks[b].col_base + blocks[b].ncols) bi = b;
    if (bi == nblocks) continue;  // column lives on another shard
    const BlockDev bd = blocks[bi];
    const uint32_t col = cg - bd.col_base;
    const int n = nk[r];
    for (int j = lane; j < n; j += 64) {
      const uint64_t h = hashes[offs[r] + j];
      const uint32_t ha = (uint32_t)(h >> 32), hb = (uint32_t)h;
      for (int t = 0; t < num_hashes; t++) {
        const uint64_t hv = num_hashes == 1 ? h : (uint64_t)(uint32_t)(ha + hb * (uint32_t)t);
        const uint64_t row = fastmod_u64(hv, bd.num_sigs, bd.magic_hi, bd.magic_lo);
        const uint64_t byte = row * bd.stride + (col >> 3);
        uint32_t* w = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(bd.rows) + (byte & ~3ULL));
        atomicOr(w, (uint32_t)(1u << (7 - (col & 7))) << (8 * (byte & 3)));
      }
    }
  }
}

void launch_plant_reads(const BlockDev* blocks, uint32_t nblocks, int num_hashes, const uint64_t* hashes, const uint64_t* offs,
                        const int32_t* nk, const uint32_t* cols, uint32_t n_reads, hipStream_t st) {
  if (n_reads == 0 || nblocks == 0) return;
  unsigned nb = (n_reads + 3) / 4;
  if (nb > 32768) nb = 32768;
  hipLaunchKernelGGL(k_plant_reads, dim3(nb), dim3(256), 0, st, blocks, nblocks, num_hashes, hashes, offs, nk, cols, n_reads);
}

}  // namespace kmcpg

namespace kmcpg {

// index building: sigs[h_i % NumSigs][col] = 1 for every hash of every column of one block (index.go:1107-1309).
// hashes = the block's columns back to back, col_off[c] = first hash of column c (n_cols+1 entries); the matrix is row-major
// with the on-disk row width (no padding), bit 7 - col%8 of byte col/8 (index.go:1157).
__global__ void k_build_scatter(uint8_t* __restrict__ sigs, uint64_t num_sigs, uint64_t mh, uint64_t ml, uint32_t row_bytes, int num_hashes,
                                const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ col_off, uint32_t col0, uint32_t n_cols, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = n_cols;  // column of hash i: last c with col_off[c] <= i
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (col_off[mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t col = col0 + lo;
    const uint64_t h = hashes[i];
    const uint32_t ha = (uint32_t)(h >> 32), hb = (uint32_t)h;
    for (int t = 0; t < num_hashes; t++) {
      const uint64_t hv = num_hashes == 1 ? h : (uint64_t)(uint32_t)(ha + hb * (uint32_t)t);
      const uint64_t byte = fastmod_u64(hv, num_sigs, mh, ml) * row_bytes + (col >> 3);
      uint32_t* w = reinterpret_cast<uint32_t*>(sigs + (byte & ~3ULL));
      atomicOr(w, (uint32_t)(1u << (7 - (col & 7))) << (8 * (byte & 3)));
    }
  }
}

void launch_build_scatter(uint8_t* sigs, uint64_t num_sigs, uint64_t mh, uint64_t ml, uint32_t row_bytes, int num_hashes, const uint64_t* hashes,
                          const uint64_t* col_off, uint32_t col0, uint32_t n_cols, uint64_t n, hipStream_t st) {
  if (n == 0) return;
  unsigned blocks = (unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
  hipLaunchKernelGGL(k_build_scatter, dim3(blocks), dim3(256), 0, st, sigs, num_sigs, mh, ml, row_bytes, num_hashes, hashes, col_off, col0, n_cols, n);
}

}  // namespace kmcpg
