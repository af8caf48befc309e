// k1_kmers.hip — K1: ntHash canonical k-mer hashes of batched reads, FracMinHash filter, Closed-Syncmer and Minimizer
// selection: one wave per short read (k1_kmers), one 1024-thread workgroup with LDS prefix arrays per long read
// (k1_kmers_wg), one workgroup per 65536-position segment of a genome (k1_seg_hash / k1_seg_pack).  Replaces bio/sketches
// NextHash / NextSyncmer / NextMinimizer behind generateKmers (kmcp/cmd/util-db-search.go:1037-1107).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <algorithm>

#include "common.hpp"
#include "device_utils.hpp"
#include "kernels.hpp"

namespace kmcpg {

// ------------------------------------------------------------------------------------------------
// K1: k-mer generation.  ntHash of the k-mer at position i in closed form:
//     fh(i) = XOR_j rol(F[i+j], k-1-j),   rh(i) = XOR_j rol(R[i+j], j)      (F = seed of the base, R = of its complement)
// Every term is a rotation of a per-position value by an amount that depends on i+j only up to a common rotation, so with
// the prefix XORs  P(n) = XOR_{m<n} ror(F[m], m)  and  Q(n) = XOR_{m<n} rol(R[m], m)
//     fh(i) = rol(P(i+k) ^ P(i), k-1+i),   rh(i) = ror(Q(i+k) ^ Q(i), i)
// i.e. one XOR scan over the bases gives the hashes of every k (and of the s-mers of a syncmer) for two look-ups each,
// instead of k table look-ups per k-mer.  The scans run on DPP within a wave (row_shr 1/2/4/8, row_bcast 15/31).
// ------------------------------------------------------------------------------------------------

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
#pragma unroll 4
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

__device__ __forceinline__ int sketch_mate(const K1Args& a, int mode, const uint8_t* s, int len, const uint64_t* tab, uint64_t* tmp_k, uint64_t* tmp_s,
                                           uint64_t* out, int cnt, int lane) {
  if (mode == 2) return syncmer_mate(s, len, a.k, (int)a.w_or_s, tab, a.scaled != 0, a.max_hash, tmp_k, tmp_s, out, cnt, lane);
  if (mode == 1) return minimizer_mate(s, len, a.k, (int)a.w_or_s, tab, a.scaled != 0, a.max_hash, tmp_k, out, cnt, lane);
  return hash_mate(s, len, a.k, tab, a.scaled != 0, a.max_hash, out, cnt, lane);
}

// one kernel per sketch mode: the plain/FracMinHash form (every short-read search) does not carry the window sketches' registers
template <int MODE>
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
      cnt = sketch_mate(a, MODE, a.seqs + o1, len1, tab, tk, ts, out, 0, lane);
      cnt1 = cnt;
      if (pe) {
        __threadfence_block();
        cnt = sketch_mate(a, MODE, a.seqs2 + o2, len2, tab, tk, ts, out, cnt, lane);
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
constexpr int K1_WAVE_SORT_CAP = DEDUP_WAVE_CAP;  // (kernels.hpp)

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
  return cnt + __builtin_amdgcn_readfirstlane(total);
}

__device__ __forceinline__ uint64_t hash_at(const uint8_t* __restrict__ s, int i, int kk, const uint64_t* tab) {
  uint64_t f = 0, r = 0;
  for (int j = 0; j < kk; j++) {
    f = rol1(f) ^ tab[s[i + j]];
    r = rol1(r) ^ tab[s[i + kk - 1 - j] & 7];
  }
  return f < r ? f : r;
}

__device__ __forceinline__ int wg_sketch_mate(const K1Args& a, int mode, const uint8_t* __restrict__ s, int len, const uint64_t* tab, uint64_t* hk,
                                              uint64_t* hs, uint64_t* __restrict__ out, int cnt, int* s_wave, int tid) {
  const int k = a.k;
  const bool scaled = a.scaled != 0;
  if (mode == 0) {
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
  if (mode == 2) {  // closed syncmer, see syncmer_mate
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
  uint64_t m4v[K1WG + K1H];                 // m4v[i] = min(hw[i..i+3]), m4i[i] = its leftmost position: the windows' arg-min reads
  uint16_t m4i[K1WG + K1H];                 // a quarter of the values (two-level scan)
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

// second level of the window arg-min: leftmost minima of every 4 consecutive entries of hw[0..n); ends with a barrier
__device__ __forceinline__ void wg_min4(K1Lds& L, int n, int tid) {
  for (int i = tid; i + 3 < n; i += K1WG) {
    uint64_t mv = L.hw[i];
    int m = i;
#pragma unroll
    for (int j = 1; j < 4; j++) {
      const uint64_t v = L.hw[i + j];
      if (v < mv) {  // strict: the leftmost of equal values wins
        mv = v;
        m = i + j;
      }
    }
    L.m4v[i] = mv;
    L.m4i[i] = (uint16_t)m;
  }
  __syncthreads();
}

// leftmost minimum of hw[b .. b+n): whole groups of 4 through m4v/m4i, the rest one by one (same result as argmin_left)
__device__ __forceinline__ int argmin_left4(const K1Lds& L, int b, int n) {
  if (n < 4) return argmin_left(L.hw, b, n);
  uint64_t mv = L.m4v[b];
  int m = b;       // group whose minimum is the best so far (its position is looked up once, at the end)
  bool grp = true;  // m names a group of 4 (true) or a single position (false)
  int i = b + 4;
  for (; i + 4 <= b + n; i += 4) {
    const uint64_t v = L.m4v[i];
    if (v < mv) {
      mv = v;
      m = i;
    }
  }
  for (; i < b + n; i++) {
    const uint64_t v = L.hw[i];
    if (v < mv) {
      mv = v;
      m = i;
      grp = false;
    }
  }
  return grp ? (int)L.m4i[m] : m;
}

// Ordered compaction of one tile of K1WG candidates with the adjacent repeats dropped on the way (window sketches emit the same
// k-mer for runs of consecutive windows: ~10 k emissions of a HiFi read hold ~1.4 k distinct adjacent values): a candidate is
// written to out[cnt ...] unless it equals the candidate before it in sequence order (the previous kept lane of its wave, else
// the last candidate of an earlier wave of the tile, else `last` = the last candidate of the tiles — and the mate — before).
// raw counts every candidate.  Returns the new (uniform) count.
struct AdjCarry {
  uint64_t last = 0;
  int have = 0;
  int raw = 0;
};
__device__ __forceinline__ int wg_compact_adj(bool keep, uint64_t h, uint64_t* __restrict__ out, int cnt, AdjCarry& c, int* s_wave, int* s_wave2, uint64_t* s_last,
                                              int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const uint64_t m = __ballot(keep);
  const uint64_t lower = m & ((1ULL << lane) - 1ULL);
  if (lane == 0) s_wave[w] = __popcll(m);
  if (m && lane == 63 - __clzll((unsigned long long)m)) s_last[w] = h;  // the wave's last candidate
  __syncthreads();
  // the candidate before this one
  const int src = lower ? 63 - __clzll((unsigned long long)lower) : lane;
  uint64_t prev = __shfl(h, src);
  bool has_prev = lower != 0;
  int raw_total = 0, last_w = -1;
#pragma unroll
  for (int i = 0; i < K1WG / 64; i++) {
    const int cw = s_wave[i];
    raw_total += cw;
    if (cw > 0) last_w = i;
  }
  if (!has_prev) {
    int pw = -1;
    for (int i = w - 1; i >= 0; i--)
      if (s_wave[i] > 0) {
        pw = i;
        break;
      }
    if (pw >= 0) {
      prev = s_last[pw];
      has_prev = true;
    } else if (c.have) {
      prev = c.last;
      has_prev = true;
    }
  }
  const bool keep2 = keep && !(has_prev && h == prev);
  const uint64_t m2 = __ballot(keep2);
  if (lane == 0) s_wave2[w] = __popcll(m2);
  const uint64_t tile_last = last_w >= 0 ? s_last[last_w] : 0;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int i = 0; i < K1WG / 64; i++) {
    const int cw = s_wave2[i];
    if (i < w) before += cw;
    total += cw;
  }
  if (keep2) out[cnt + before + __popcll(m2 & ((1ULL << lane) - 1ULL))] = h;
  // uniform values: kept in scalar registers (the compiler cannot see that every thread computed the same)
  c.raw += __builtin_amdgcn_readfirstlane(raw_total);
  if (__builtin_amdgcn_readfirstlane(last_w) >= 0) {
    c.last = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tile_last >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tile_last);
    c.have = 1;
  }
  __syncthreads();
  return cnt + __builtin_amdgcn_readfirstlane(total);
}

// positions per tile: with a small halo the tile shrinks so that positions + halo fit one scan round of K1WG bases
__device__ __forceinline__ int wg_tile_step(int halo) { return halo <= K1WG / 2 ? K1WG - halo : K1WG; }

__host__ __device__ __forceinline__ bool wg_lds_usable(const K1Args& a) {
  if (a.k > 255) return false;
  if (a.mode == 2) return 2 * a.k - (int)a.w_or_s - 1 <= K1H && (int)a.w_or_s >= 1 && (int)a.w_or_s <= a.k;
  if (a.mode == 1) return (int)a.w_or_s >= 1 && (int)a.w_or_s + 1 < K1H;
  return true;
}

__device__ __forceinline__ int wg_sketch_mate_lds(const K1Args& a, int mode, const uint8_t* __restrict__ s, int len, const uint64_t* tab, K1Lds& L,
                                                  uint64_t* __restrict__ out, int cnt, int* s_wave, int tid) {
  const int k = a.k;
  const bool scaled = a.scaled != 0;
  const int nk = len - k + 1;  // k-mer positions
  if (nk <= 0) return cnt;
  if (mode == 0) {
    const int T = wg_tile_step(k - 1);
    for (int p0 = 0; p0 < nk; p0 += T) {
      wg_prefix(s + p0, min(len - p0, T + k - 1), tab, L, tid);
      const bool v = tid < T && p0 + tid < nk;
      const uint64_t h = v ? lds_hash(L, tid, k) : 0;
      cnt = wg_compact(v && h != 0 && (!scaled || h <= a.max_hash), h, out, cnt, s_wave, tid);
    }
    return cnt;
  }
  return cnt;
}

// The window sketches (Closed Syncmer, Minimizer) of one mate in the LDS tile form.  ADJ = false: every emission is written to
// out[cnt...] (what the reference's generateKmers returns); ADJ = true: adjacent repeats are dropped on the way and c.raw counts
// the emissions (input of the sort + unique that queries above -u get anyway).  The arg-min of a window reads the 4-wide minima
// of wg_min4 (two-level scan): 5 + 0 look-ups instead of 20 for the 20-s-mer windows of k = 21, s = 11.
template <bool ADJ>
__device__ __forceinline__ int wg_window_mate_lds(const K1Args& a, int mode, const uint8_t* __restrict__ s, int len, const uint64_t* tab, K1Lds& L,
                                                  uint64_t* __restrict__ out, int cnt, AdjCarry& c, int* s_wave, int* s_wave2, uint64_t* s_last, int tid) {
  const int k = a.k;
  const bool scaled = a.scaled != 0;
  const int nk = len - k + 1;  // k-mer positions
  if (nk <= 0) return cnt;
  auto emit = [&](bool keep, uint64_t h) {
    if (ADJ) cnt = wg_compact_adj(keep, h, out, cnt, c, s_wave, s_wave2, s_last, tid);
    else cnt = wg_compact(keep, h, out, cnt, s_wave, tid);
  };
  if (mode == 2) {  // closed syncmer (see syncmer_mate)
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
      const bool two = wsz >= 4 && (a.flags & 1);
      if (two) wg_min4(L, nst, tid);
      const bool v = tid < T && p0 + tid < nw;
      uint64_t h = 0;
      if (v) {
        int pos = tid;
        if (wsz > 0) {
          const int m = two ? argmin_left4(L, tid, wsz) : argmin_left(L.hw, tid, wsz);
          pos = (m - tid < k - sm) ? m : m + sm - k;
        }
        h = lds_hash(L, pos, k);
      }
      emit(v && h != 0 && (!scaled || h <= a.max_hash), h);
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
    const bool two = w >= 4 && (a.flags & 1);
    if (two) wg_min4(L, nkt, tid);
    const int w0 = p0 + tid;
    const bool v = tid < T && w0 < nw;
    int m = -1, pm = -2;
    if (v) {
      m = two ? argmin_left4(L, off + tid, w) : argmin_left(L.hw, off + tid, w);
      pm = w0 <= 0 ? -2 : (two ? argmin_left4(L, off + tid - 1, w) : argmin_left(L.hw, off + tid - 1, w));
    }
    const uint64_t h = (v && m != pm) ? L.hw[m] : 0;
    emit(v && m != pm && h != 0 && (!scaled || h <= a.max_hash), h);
  }
  return cnt;
}

// the per-read loop of the workgroup kernels; LDS = the tile form with prefix arrays in LDS, else the scratch-buffer form
template <bool LDS>
__device__ __forceinline__ void wg_reads(const K1Args& a, int mode, const uint64_t* tab, int* s_wave, K1Lds* lds, int tid) {
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
      if (LDS) {
        cnt = wg_sketch_mate_lds(a, mode, a.seqs + o1, len1, tab, *lds, out, 0, s_wave, tid);
        cnt1 = cnt;
        if (pe) cnt = wg_sketch_mate_lds(a, mode, a.seqs2 + o2, len2, tab, *lds, out, cnt, s_wave, tid);
      } else {
        uint64_t* tk = a.scratch ? a.scratch + o1 + o2 : nullptr;
        uint64_t* ts = a.scratch2 ? a.scratch2 + o1 + o2 : nullptr;
        cnt = wg_sketch_mate(a, mode, a.seqs + o1, len1, tab, tk, ts, out, 0, s_wave, tid);
        cnt1 = cnt;
        if (pe) cnt = wg_sketch_mate(a, mode, a.seqs2 + o2, len2, tab, tk, ts, out, cnt, s_wave, tid);
      }
    }
    if (tid == 0) {
      a.nk_raw[r] = cnt;
      a.nk1[r] = cnt1;
      a.qlen[r] = len1 + len2;
    }
  }
}

// The same for the window sketches in the LDS form, with the adjacent-repeat filter of the dedup path fused in.  A query whose
// emissions exceed -u is sorted and uniqued afterwards (:874-908), and window sketches emit runs of equal values, so what the sort
// needs is the sequence without adjacent repeats: it is written to scratch[...] (its length to nk_adj[r]) while the emissions are
// only counted — 1.4 k values instead of 10 k per HiFi read, and no separate k_adj_unique pass.  The emissions themselves are only
// needed as they are for queries that are not deduplicated (<= -u), for those the wave sort takes (<= K1_WAVE_SORT_CAP) and for
// whole-genome queries (device-wide sort): those — short reads in a long-read batch, rare — are sketched a second time into
// hashes[...].  A query that cannot exceed the bound by its length skips the first pass.
template <int MODE>
__device__ __forceinline__ void wg_reads_windows(const K1Args& a, const uint64_t* tab, int* s_wave, int* s_wave2, uint64_t* s_last, K1Lds& L, int tid) {
  const int raw_bound = a.dedup_threshold > K1_WAVE_SORT_CAP ? a.dedup_threshold : K1_WAVE_SORT_CAP;
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
    const bool skip = len1 < a.min_qlen && !(pe && len2 >= a.min_qlen);
    int raw = 0, raw1 = 0;
    if (!skip) {
      bool need_raw = true;
      if ((a.flags & 2) && a.nk_adj && a.scratch && len1 + len2 > raw_bound) {  // (a mate emits at most one value per base)
        AdjCarry c;
        uint64_t* adj = a.scratch + o1 + o2;
        int m = wg_window_mate_lds<true>(a, MODE, a.seqs + o1, len1, tab, L, adj, 0, c, s_wave, s_wave2, s_last, tid);
        raw1 = c.raw;
        if (pe) m = wg_window_mate_lds<true>(a, MODE, a.seqs2 + o2, len2, tab, L, adj, m, c, s_wave, s_wave2, s_last, tid);
        raw = c.raw;
        need_raw = raw <= raw_bound || raw > (int)HUGE_MIN;
        if (!need_raw && tid == 0) a.nk_adj[r] = m;
      }
      if (need_raw) {
        AdjCarry c;
        uint64_t* out = a.hashes + o1 + o2;
        raw = wg_window_mate_lds<false>(a, MODE, a.seqs + o1, len1, tab, L, out, 0, c, s_wave, s_wave2, s_last, tid);
        raw1 = raw;
        if (pe) raw = wg_window_mate_lds<false>(a, MODE, a.seqs2 + o2, len2, tab, L, out, raw, c, s_wave, s_wave2, s_last, tid);
      }
    }
    if (tid == 0) {
      a.nk_raw[r] = raw;
      a.nk1[r] = raw1;
      a.qlen[r] = len1 + len2;
    }
  }
}

// One kernel per sketch mode (the other modes' code and registers stay out of it); 64 VGPRs => two workgroups per CU.
template <int MODE>
__global__ void __launch_bounds__(K1WG, 8) k1_kmers_wg(const K1Args a) {
  __shared__ uint64_t tab[256];
  __shared__ int s_wave[K1WG / 64];
  __shared__ K1Lds lds;
  const int tid = threadIdx.x;
  if (tid < 256) tab[tid] = seed_of(tid);
  __syncthreads();
  if constexpr (MODE == 0) {
    wg_reads<true>(a, MODE, tab, s_wave, &lds, tid);
  } else {
    __shared__ int s_wave2[K1WG / 64];
    __shared__ uint64_t s_last[K1WG / 64];
    wg_reads_windows<MODE>(a, tab, s_wave, s_wave2, s_last, lds, tid);
  }
}

// ---- window sketches of long reads, barrier-free form (k <= 65, windows of at most 64 hashes: every practical setting) ---------
// The 1024-thread tile form above spends its time in workgroup barriers (eight per tile of ~1000 windows: prefix arrays,
// hashes, minima, ordered compaction).  Here the 16 waves of the workgroup never wait for each other while they sketch: the
// windows of a mate are cut into 16 contiguous segments, one per wave; a wave walks its segment in tiles of 64 windows with the
// wave-level machinery of the short-read kernel (prefix XOR scans on DPP, wave_hash), keeps the s-mer / k-mer hashes of the current
// and the next tile in a 2-slot ring in LDS that only it touches (wave-level fences), finds the windows' leftmost minima there,
// filters adjacent repeats with ballot + shuffle and appends to a private piece of a temporary array.  Two workgroup barriers per
// mate remain: one before and one after the pieces are moved together in order (dropping an element that repeats across a segment
// boundary).  Semantics are those of syncmer_mate / minimizer_mate.
constexpr int K1W_THREADS = 512;
constexpr int K1W_WAVES = K1W_THREADS / 64;
// Per wave: a ring over two tiles (position p relative to the segment start lives at p & 127) of the inclusive prefix XORs, the
// hashes the windows scan (s-mers for syncmers, k-mers for minimizers), the k-mer hashes, and the second level of the window
// arg-min: m4v[p] = min(hw[p..p+3]), m4i[p] = offset of its leftmost position.  The kernel is bound by VALU issue (a wave64
// instruction takes four cycles on a 16-lane SIMD), so what counts is instructions per window: hashes come from two LDS look-ups
// of the prefix ring instead of eight cross-lane permutes, and a 20-wide window costs 5 + 8 look-ups instead of 20.
struct K1WRing {
  uint64_t ip[128], iq[128];
  uint64_t hw[128];
  uint64_t hk[128];
  uint64_t m4v[128];
  uint8_t m4i[128];
};
struct K1WShared {
  int cnt[K1W_WAVES], raw[K1W_WAVES];
  uint64_t first[K1W_WAVES], last[K1W_WAVES];
};
struct SeqCarry {  // uniform over the workgroup
  uint64_t last = 0;
  int have = 0, raw = 0, cnt = 0;
};

__host__ __device__ __forceinline__ bool wave_windows_usable(const K1Args& a) {
  if (a.k > K1_SCAN_MAX_K || a.k < 1) return false;
  const int ws = (int)a.w_or_s;
  if (a.mode == 2) return ws >= 1 && ws <= a.k && 2 * (a.k - ws) <= 60;
  if (a.mode == 1) return ws >= 1 && ws <= 60;
  return false;
}

__device__ __forceinline__ uint64_t readfirst64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// canonical hash of the kk-mer at ring position p = 64 T + lane of the tile whose scan is `t` (its exclusive prefixes are in
// registers, the inclusive prefix at the kk-mer's last base comes from the ring)
__device__ __forceinline__ uint64_t ring_hash(const K1WRing& ring, const WTile& t, int p, int kk, int lane) {
  const int e = (p + kk - 1) & 127;
  const uint64_t f = rolv(ring.ip[e] ^ t.ip ^ t.xp, kk - 1 + lane), r = rorv(ring.iq[e] ^ t.iq ^ t.xq, lane);
  return f < r ? f : r;
}

// m4v / m4i of ring position p (needs hw[p .. p+3])
__device__ __forceinline__ void ring_min4(K1WRing& ring, int p) {
  uint64_t mv = ring.hw[p & 127];
  int m = 0;
#pragma unroll
  for (int j = 1; j < 4; j++) {
    const uint64_t x = ring.hw[(p + j) & 127];
    if (x < mv) {  // strict: the leftmost of equal values wins
      mv = x;
      m = j;
    }
  }
  ring.m4v[p & 127] = mv;
  ring.m4i[p & 127] = (uint8_t)m;
}

// leftmost minimum of hw[p0 .. p0+n) through the 4-wide minima (n >= 1); returns the position, *val = the value
__device__ __forceinline__ int ring_argmin(const K1WRing& ring, int p0, int n, uint64_t* val) {
  int i = 0;
  uint64_t mv;
  int m;       // start of the best group (grp) or the best single position
  bool grp;
  if (n >= 4) {
    mv = ring.m4v[p0 & 127];
    m = p0;
    grp = true;
    for (i = 4; i + 4 <= n; i += 4) {
      const uint64_t x = ring.m4v[(p0 + i) & 127];
      if (x < mv) {
        mv = x;
        m = p0 + i;
      }
    }
  } else {
    mv = ring.hw[p0 & 127];
    m = p0;
    grp = false;
    i = 1;
  }
  for (; i < n; i++) {
    const uint64_t x = ring.hw[(p0 + i) & 127];
    if (x < mv) {
      mv = x;
      m = p0 + i;
      grp = false;
    }
  }
  *val = mv;
  return grp ? m + (int)ring.m4i[m & 127] : m;
}

template <int MODE, bool ADJ>
__device__ __forceinline__ void wgw_mate(const K1Args& a, const uint8_t* __restrict__ s, int len, const uint64_t* tab, K1WRing& ring, K1WShared& sh,
                                         uint64_t* __restrict__ temp, uint64_t* __restrict__ fin, SeqCarry& c, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const int k = a.k;
  const bool scaled = a.scaled != 0;
  const int nk = len - k + 1;
  int wsz = 0, nw = 0, sm = 0;
  if (MODE == 2) {
    sm = (int)a.w_or_s;
    const int Lw = 2 * k - sm - 1;
    wsz = 2 * (k - sm);
    nw = (nk <= 0 || len < Lw) ? 0 : (wsz > 0 ? len - Lw + 1 : nk);
  } else {
    wsz = (int)a.w_or_s;
    nw = (nk <= 0 || len < k + wsz - 1) ? 0 : nk - wsz + 1;
  }
  if (nw <= 0) return;  // ErrShortSeq (uniform)
  const int segw = ((nw + K1W_WAVES * 64 - 1) / (K1W_WAVES * 64)) * 64;  // windows per wave, a multiple of 64
  const int w0s = w * segw, w0e = min(nw, w0s + segw);
  int cnt = 0, raw = 0;
  uint64_t first = 0, last = 0;
  bool any = false;
  if (w0s < w0e) {
    const int ntiles = (w0e - w0s + 63) / 64;
    uint64_t cp = 0, cq = 0;
    WTile B = wave_tile(s, len, w0s, tab, cp, cq, lane);
    ring.ip[lane] = B.ip;
    ring.iq[lane] = B.iq;
    int pm_carry = -2;  // minimizer: arg-min of the window before the tile's first one
    // iteration t: scan tile t+1, hash tile t, windows of tile t-1 (their hashes reach into tile t)
    for (int t = 0; t <= ntiles; t++) {
      const WTile C = wave_tile(s, len, w0s + 64 * (t + 1), tab, cp, cq, lane);
      const int nslot = ((t + 1) & 1) * 64;
      ring.ip[nslot + lane] = C.ip;
      ring.iq[nslot + lane] = C.iq;
      wave_lds_fence();
      const int pt = 64 * t + lane;
      const uint64_t hkv = ring_hash(ring, B, pt, k, lane);
      ring.hk[pt & 127] = hkv;
      ring.hw[pt & 127] = MODE == 2 ? ring_hash(ring, B, pt, sm, lane) : hkv;
      B = C;
      wave_lds_fence();
      if (wsz >= 4) {
        // one pass: the 4-wide minima of this tile's positions 0..60 and of the previous tile's 61..63 (whose last elements
        // have just arrived); the windows below read the previous tile's and this tile's first 57 (wsz <= 60)
        const int pm4 = lane <= 60 ? pt : pt - 64;
        if (pm4 >= 0) ring_min4(ring, pm4);
        wave_lds_fence();
      }
      if (t == 0) continue;
      const int p0 = 64 * (t - 1) + lane;  // this lane's window, relative to w0s
      const bool v = w0s + p0 < w0e;
      uint64_t h = 0;
      bool keep = false;
      if (MODE == 2) {
        int pos = p0;
        if (wsz > 0) {
          uint64_t mv;
          const int m = ring_argmin(ring, p0, wsz, &mv);
          pos = (m - p0 < k - sm) ? m : m + sm - k;
        }
        h = ring.hk[pos & 127];
        keep = v && h != 0 && (!scaled || h <= a.max_hash);
      } else {
        uint64_t mv;
        const int m = ring_argmin(ring, p0, wsz, &mv);
        if (t == 1 && w0s > 0) {  // (uniform branch) window w0s-1 = {k-mer w0s-1} + the first wsz-1 of window w0s
          int pm0 = -1;
          if (lane == 0) {
            const uint64_t x = hash_at(s, w0s - 1, k, tab);
            int m2 = -1;
            uint64_t mv2 = ~0ULL;
            for (int j = 0; j < wsz - 1; j++) {
              const uint64_t y = ring.hk[j & 127];
              if (m2 < 0 || y < mv2) {
                mv2 = y;
                m2 = j;
              }
            }
            pm0 = (m2 < 0 || x <= mv2) ? -1 : m2;  // relative position; -1 = k-mer w0s-1 itself
          }
          pm_carry = __builtin_amdgcn_readfirstlane(pm0);
        }
        int pm = __shfl_up(m, 1);
        if (lane == 0) pm = pm_carry;  // -2 at the very first window of the mate: always emitted
        pm_carry = __builtin_amdgcn_readlane(m, 63);
        h = mv;
        keep = v && m != pm && h != 0 && (!scaled || h <= a.max_hash);
      }
      const uint64_t mk = __ballot(keep);
      const uint64_t lower = mk & ((1ULL << lane) - 1ULL);
      bool keep2 = keep;
      if (ADJ) {
        const int src = lower ? 63 - __clzll((unsigned long long)lower) : lane;
        const uint64_t prev = __shfl(h, src);
        const bool has_prev = lower != 0 || any;
        keep2 = keep && !(has_prev && h == (lower ? prev : last));
      }
      if (mk) {
        const int hi = 63 - __clzll((unsigned long long)mk), lo = __ffsll((unsigned long long)mk) - 1;
        const uint64_t lastv = readfirst64(__shfl(h, hi));
        if (!any) first = readfirst64(__shfl(h, lo));
        last = lastv;
        any = true;
      }
      const uint64_t mk2 = ADJ ? __ballot(keep2) : mk;
      if (keep2) temp[w0s + cnt + __popcll(mk2 & ((1ULL << lane) - 1ULL))] = h;
      cnt += __popcll(mk2);
      raw += __popcll(mk);
      wave_lds_fence();  // the next iteration's hashes and minima overwrite what these windows read
    }
  }
  if (lane == 0) {
    sh.cnt[w] = cnt;
    sh.raw[w] = raw;
    sh.first[w] = first;
    sh.last[w] = last;
  }
  __threadfence_block();
  __syncthreads();
  // the segments' pieces move together, in order
  int off = c.cnt, have = c.have, my_off = 0, my_drop = 0, rawsum = 0;
  uint64_t plast = c.last;
#pragma unroll
  for (int j = 0; j < K1W_WAVES; j++) {
    const int cj = sh.cnt[j];
    rawsum += sh.raw[j];
    const int drop = (ADJ && cj > 0 && have && sh.first[j] == plast) ? 1 : 0;
    if (j == w) {
      my_off = off;
      my_drop = drop;
    }
    off += cj - drop;
    if (sh.raw[j] > 0) {
      plast = sh.last[j];
      have = 1;
    }
  }
  for (int i = lane + my_drop; i < cnt; i += 64) fin[my_off + i - my_drop] = temp[w0s + i];
  c.cnt = __builtin_amdgcn_readfirstlane(off);
  c.have = __builtin_amdgcn_readfirstlane(have);
  c.last = readfirst64(plast);
  c.raw += __builtin_amdgcn_readfirstlane(rawsum);
  __syncthreads();
}

// the per-read loop of the barrier-free window kernel: same contract as wg_reads_windows (scratch[] + nk_adj[] for queries that
// will be sorted, hashes[] for the others)
template <int MODE>
__global__ void __launch_bounds__(K1W_THREADS) k1_windows_wave(const K1Args a) {
  __shared__ uint64_t tab[256];
  __shared__ K1WRing rings[K1W_WAVES];
  __shared__ K1WShared sh;
  const int tid = threadIdx.x;
  if (tid < 256) tab[tid] = seed_of(tid);
  __syncthreads();
  K1WRing& ring = rings[tid >> 6];
  const int raw_bound = a.dedup_threshold > K1_WAVE_SORT_CAP ? a.dedup_threshold : K1_WAVE_SORT_CAP;
  // (behind k1_windows_roll: only the reads that kernel put on its list)
  const uint32_t n_todo = a.seg_only_flagged ? *a.seg_nflag : a.n_reads;
  for (uint32_t it = blockIdx.x; it < n_todo; it += gridDim.x) {
    const uint32_t r = a.seg_only_flagged ? a.seg_list[it] : it;
    const uint64_t o1 = a.offs[r];
    const int len1 = (int)(a.offs[r + 1] - o1);
    uint64_t o2 = 0;
    int len2 = 0;
    const bool pe = a.offs2 != nullptr;
    if (pe) {
      o2 = a.offs2[r];
      len2 = (int)(a.offs2[r + 1] - o2);
    }
    const bool skip = len1 < a.min_qlen && !(pe && len2 >= a.min_qlen);
    int raw = 0, raw1 = 0;
    if (!skip) {
      uint64_t* hs_out = a.hashes + o1 + o2;
      uint64_t* sc_out = a.scratch + o1 + o2;
      bool need_raw = true;
      if (a.nk_adj && len1 + len2 > raw_bound) {  // (a mate emits at most one value per base)
        SeqCarry c;
        wgw_mate<MODE, true>(a, a.seqs + o1, len1, tab, ring, sh, hs_out, sc_out, c, tid);
        raw1 = c.raw;
        if (pe) wgw_mate<MODE, true>(a, a.seqs2 + o2, len2, tab, ring, sh, hs_out + len1, sc_out, c, tid);
        raw = c.raw;
        need_raw = raw <= raw_bound || raw > (int)HUGE_MIN;
        if (!need_raw && tid == 0) a.nk_adj[r] = c.cnt;
      }
      if (need_raw) {
        SeqCarry c;
        wgw_mate<MODE, false>(a, a.seqs + o1, len1, tab, ring, sh, sc_out, hs_out, c, tid);
        raw1 = c.raw;
        if (pe) wgw_mate<MODE, false>(a, a.seqs2 + o2, len2, tab, ring, sh, sc_out + len1, hs_out, c, tid);
        raw = c.raw;
      }
    }
    if (tid == 0) {
      a.nk_raw[r] = raw;
      a.nk1[r] = raw1;
      a.qlen[r] = len1 + len2;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Closed-syncmer sketches of long reads by ROLLING (round 6; docs/ROUND_NOTES.md has the costing).  k1_windows_wave gives a lane one
// position and pays ~4.2 wave instructions per window (two 64-bit XOR scans, two closed-form hashes, a two-level arg-min); here a lane
// owns a RUN of consecutive windows of one read (one wave per read, run = windows / 64 rounded up to 16) and walks it on 2-bit codes:
//   * two rolling ntHash states (s-mers and k-mers; nthash.hpp: fh' = rol1(fh) ^ F2[out][in], rh' = ror1(rh ^ R2[out][in]), the pair
//     tables as ONE 16-byte LDS look-up per roll); the codes of the four bases a step needs come from four 16-code shift registers that
//     are refilled every 16 steps (runs start at multiples of 16, so the refill is uniform over the wave);
//   * the leftmost minimum of the window's WSZ = 2 (k - s) s-mer hashes without a data-dependent branch: the s-mer hashes are cut into
//     blocks of WSZ; a block's suffix minima (ties to the left) are scanned in place, in registers, once the block is complete, its
//     prefix minima grow as the next block's hashes arrive, and window i = min(suffix[i], prefix[i + WSZ - 1]) with ties to the suffix
//     (a per-lane monotonic queue would make SOME lane rescan at almost every step);
//   * the k-mer hash of the emission position (within k - s of the window start) from a per-lane ring in LDS ([slot][lane]: every lane
//     keeps to its own two banks whatever slot it reads);
//   * emissions without adjacent repeats go to the lane's own stretch of hashes[] and are moved together in order afterwards, the runs
//     stitched on their first / last values (what k1_windows_wave does across waves).
// Same outputs as k1_windows_wave's fused path: scratch[offs[r] ...] + nk_adj[r], nk_raw / nk1 / qlen.  A read this kernel cannot take
// — any byte other than A/C/G/T in either case, fewer windows than WR_MIN_WINDOWS, longer than the LDS holds, or an emission count
// outside (raw_bound, HUGE_MIN] (the fused path does not apply) — is put on a list for k1_windows_wave, launched right behind.
// Reference: sketches.NewSyncmerSketch / NextSyncmer behind generateKmers (util-db-search.go:1053,1068).
// ------------------------------------------------------------------------------------------------
constexpr int WR_MIN_WINDOWS = 1024;
constexpr int WR_PAD_BASES = 1024 + 256;  // what the last lanes' (predicated-off) steps and the refills may still read: zeros

__host__ __device__ __forceinline__ int wr_words_for(int max_read_len) { return (max_read_len + WR_PAD_BASES + 15) / 16 + 4; }
__host__ __device__ __forceinline__ int wr_ring(int wsz) { return wsz <= 30 ? 16 : 32; }  // >= k - s + 1 = wsz / 2 + 1 slots
__host__ __device__ __forceinline__ size_t wr_lds_bytes(int wsz, int words, int waves) { return 1024 + (size_t)waves * ((size_t)wr_ring(wsz) * 64 * 8 + (size_t)words * 4); }

template <int WSZ, int WR_WAVES>
__global__ void __launch_bounds__(64 * WR_WAVES) k1_windows_roll(const K1Args a, int words) {
  extern __shared__ __attribute__((aligned(16))) uint8_t wr_lds[];
  constexpr int KR = WSZ <= 30 ? 16 : 32;
  constexpr int D = WSZ / 2;  // k - s
  uint4* const TK = reinterpret_cast<uint4*>(wr_lds);         // [16] {F2, R2} of the k-mer roll
  uint4* const TS = TK + 16;                                   // [16] ... of the s-mer roll
  uint64_t* const SD = reinterpret_cast<uint64_t*>(TS + 16);   // [4] seeds by code, [4] seeds of the complements
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint8_t* const mine_lds = wr_lds + 1024 + (size_t)w * ((size_t)KR * 64 * 8 + (size_t)words * 4);
  uint64_t* const ring = reinterpret_cast<uint64_t*>(mine_lds);                    // [KR][64]
  uint32_t* const Wd = reinterpret_cast<uint32_t*>(mine_lds + (size_t)KR * 64 * 8);  // [words]: 16 codes each
  const int k = a.k, sm = (int)a.w_or_s;
  if (tid < 16) {
    const uint64_t fk = nt2_f2(tid >> 2, tid & 3, k), rk = nt2_r2(tid >> 2, tid & 3, k);
    const uint64_t fs = nt2_f2(tid >> 2, tid & 3, sm), rs = nt2_r2(tid >> 2, tid & 3, sm);
    TK[tid] = make_uint4((uint32_t)fk, (uint32_t)(fk >> 32), (uint32_t)rk, (uint32_t)(rk >> 32));
    TS[tid] = make_uint4((uint32_t)fs, (uint32_t)(fs >> 32), (uint32_t)rs, (uint32_t)(rs >> 32));
    if (tid < 4) {
      SD[tid] = seed_of(nt2_letter(tid));
      SD[4 + tid] = seed_of(nt2_letter(tid) & 7);
    }
  }
  __syncthreads();
  const uint32_t r = blockIdx.x * WR_WAVES + (uint32_t)w;
  if (r >= a.n_reads) return;  // (no barrier below: the waves of a workgroup are independent from here on)
  const uint64_t o1 = a.offs[r];
  const int len = (int)(a.offs[r + 1] - o1);
  const int Lw = 2 * k - sm - 1;
  const int nw = len - Lw + 1;  // windows (WSZ > 0)
  const int raw_bound = a.dedup_threshold > K1_WAVE_SORT_CAP ? a.dedup_threshold : K1_WAVE_SORT_CAP;
  auto leave_to_wave_kernel = [&]() {
    if (lane == 0) a.seg_list[atomicAdd(a.seg_nflag, 1u)] = r;
  };
  if (len < a.min_qlen || len <= raw_bound || nw < WR_MIN_WINDOWS || len + WR_PAD_BASES > (words - 4) * 16) {
    leave_to_wave_kernel();
    return;
  }
  // ---- the read as 2-bit codes (zeros behind its end)
  {
    const uint8_t* __restrict__ s = a.seqs + o1;
    typedef uint32_t u32x4_any __attribute__((ext_vector_type(4), aligned(1)));
    bool bad = false;
    for (int gi = lane; gi < words; gi += 64) {
      const int b0 = gi * 16;
      uint32_t word = 0;
      if (b0 + 16 <= len) {
        const u32x4_any v = *reinterpret_cast<const u32x4_any*>(s + b0);
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t c = nt2_codes4(in[d]);
          bad |= !nt2_valid4(in[d], c);
          word |= nt2_fold4(c) << (8 * d);
        }
      } else if (b0 < len) {
        for (int j = 0; b0 + j < len; j++) {
          const uint32_t ch = s[b0 + j], c = (ch >> 1) & 3u;
          bad |= (ch & 0xDFu) != (uint32_t)nt2_letter((int)c);
          word |= c << (2 * j);
        }
      }
      Wd[gi] = word;
    }
    if (__ballot(bad) != 0) {
      leave_to_wave_kernel();
      return;
    }
  }
  wave_lds_fence();
  const int L = (((nw + 63) / 64) + 15) & ~15;  // windows per lane, a multiple of 16
  const int q0 = lane * L, q1 = min(nw, q0 + L);
  const bool scaled = a.scaled != 0;
  const uint64_t max_hash = a.max_hash;
  auto code_at = [&](int q) -> uint32_t { return (Wd[q >> 4] >> (2 * (q & 15))) & 3u; };
  auto start_up = [&](int q, int kk, uint64_t& fh, uint64_t& rh) {  // fh = XOR_j rol(F[j], kk-1-j), rh = XOR_j rol(R[j], j) of the kk-mer at q
    fh = 0;
    rh = 0;
    for (int j = 0; j < kk; j++) {
      const uint32_t c = code_at(q + j);
      fh = nt2_rol1(fh) ^ SD[c];
      rh ^= rolv(SD[4 + c], j);
    }
  };
  auto roll = [&](uint64_t& fh, uint64_t& rh, uint32_t oc, uint32_t ic, const uint4* T) __attribute__((always_inline)) {
    const uint4 t = T[(oc << 2) | ic];
    fh = nt2_rol1(fh) ^ (((uint64_t)t.y << 32) | t.x);
    rh = nt2_ror1(rh ^ (((uint64_t)t.w << 32) | t.z));
  };
  // ---- warm-up: the first block of s-mer hashes, the first D k-mer hashes
  uint64_t x[WSZ];
  int pidx[WSZ];
  uint64_t sfh, srh, kfh, krh;
  start_up(q0, sm, sfh, srh);
  x[0] = sfh < srh ? sfh : srh;
#pragma unroll
  for (int j = 1; j < WSZ; j++) {
    roll(sfh, srh, code_at(q0 + j - 1), code_at(q0 + j - 1 + sm), TS);
    x[j] = sfh < srh ? sfh : srh;
  }
  start_up(q0, k, kfh, krh);
  ring[(q0 & (KR - 1)) * 64 + lane] = kfh < krh ? kfh : krh;
  for (int t = 1; t <= D; t++) {  // k-mer hashes of q0 .. q0 + D: the roll stays ONE position ahead of what the next window may ask for
    roll(kfh, krh, code_at(q0 + t - 1), code_at(q0 + t - 1 + k), TK);
    ring[((q0 + t) & (KR - 1)) * 64 + lane] = kfh < krh ? kfh : krh;
  }
  auto suffix_scan = [&]() __attribute__((always_inline)) {  // x[j] <- min(x[j .. WSZ-1]), pidx[j] <- its leftmost position in the block
    pidx[WSZ - 1] = WSZ - 1;
#pragma unroll
    for (int j = WSZ - 2; j >= 0; j--) {
      const bool left = x[j] <= x[j + 1];
      pidx[j] = left ? j : pidx[j + 1];
      x[j] = left ? x[j] : x[j + 1];
    }
  };
  suffix_scan();
  // the codes a step needs, relative to its window i: s-mer roll out / in (to position i + WSZ), k-mer roll out / in (to i + D + 1)
  const int d_so = WSZ - 1, d_si = WSZ - 1 + sm, d_ko = D, d_ki = D + k;
  uint32_t c_so = 0, c_si = 0, c_ko = 0, c_ki = 0;
  auto refill = [&](int i, int d) -> uint32_t {  // 16 codes from base i + d on (i is a multiple of 16)
    const int b = i + d;
    return nt2_funnel(Wd[(b >> 4) + 1], Wd[b >> 4], (uint32_t)(2 * (b & 15)));
  };
  uint64_t* __restrict__ temp = a.hashes + o1 + q0;
  int cnt = 0, raw = 0;
  uint64_t first = 0, last = 0, pv = ~0ULL;
  int pp = 0;
  auto emit = [&](uint64_t h, bool on) __attribute__((always_inline)) {
    // (almost every window keeps its value — a hash is 0 or above maxHash rarely — so what is kept is tracked with selects; the one
    // branch is the store of a value that differs from the lane's previous one, a fifth of the steps per lane)
    const bool keep = on && h != 0 && (!scaled || h <= max_hash);
    if (keep && (raw == 0 || h != last)) {
      if (cnt == 0) first = h;  // (the first kept value is always stored)
      temp[cnt++] = h;
    }
    last = keep ? h : last;
    raw += keep ? 1 : 0;
  };
  // The step is software-pipelined by hand: the k-mer hash of window i is ASKED for at the top of step i and USED at the top of step
  // i + 1, and the pair-table entries of step i + 1's rolls are asked for at the end of step i — an LDS round trip is ~100 cycles and
  // the first version (three of them waited for in every step, two waves per SIMD) ran at a fifth of the issue rate.
  c_so = refill(q0, d_so);
  c_si = refill(q0, d_si);
  c_ko = refill(q0, d_ko);
  c_ki = refill(q0, d_ki);
  uint4 ts = TS[((c_so & 3u) << 2) | (c_si & 3u)], tk = TK[((c_ko & 3u) << 2) | (c_ki & 3u)];
  uint64_t hq = 0;   // the k-mer hash window i - 1 emits (arriving)
  bool on_q = false;  // ... and whether that window exists
  const int nblk = (L + WSZ - 1) / WSZ;
  int n = 0;  // windows done by every lane (uniform)
  for (int b = 0; b < nblk; b++) {
    const int base = q0 + b * WSZ;
#pragma unroll
    for (int j = 0; j < WSZ; j++, n++) {
      const int i = base + j;
      // window i: the suffix of this block from j on, the first j hashes of the next block
      const bool right = pv < x[j];
      const int m = right ? pp : base + pidx[j];
      const int pos = (m - i < D) ? m : m + sm - k;
      const uint64_t hcur = ring[(pos & (KR - 1)) * 64 + lane];
      emit(hq, on_q);
      // the s-mer hash at i + WSZ joins the next block's prefix; the k-mer hash at i + D + 1 enters the ring
      sfh = nt2_rol1(sfh) ^ (((uint64_t)ts.y << 32) | ts.x);
      srh = nt2_ror1(srh ^ (((uint64_t)ts.w << 32) | ts.z));
      kfh = nt2_rol1(kfh) ^ (((uint64_t)tk.y << 32) | tk.x);
      krh = nt2_ror1(krh ^ (((uint64_t)tk.w << 32) | tk.z));
      const uint64_t hs = sfh < srh ? sfh : srh;
      if (hs < pv) {
        pv = hs;
        pp = i + WSZ;
      }
      x[j] = hs;
      ring[((i + D + 1) & (KR - 1)) * 64 + lane] = kfh < krh ? kfh : krh;
      c_so >>= 2;
      c_si >>= 2;
      c_ko >>= 2;
      c_ki >>= 2;
      if (((n + 1) & 15) == 0) {  // uniform: the next 16 steps' codes
        c_so = refill(q0 + n + 1, d_so);
        c_si = refill(q0 + n + 1, d_si);
        c_ko = refill(q0 + n + 1, d_ko);
        c_ki = refill(q0 + n + 1, d_ki);
      }
      ts = TS[((c_so & 3u) << 2) | (c_si & 3u)];
      tk = TK[((c_ko & 3u) << 2) | (c_ki & 3u)];
      hq = hcur;
      on_q = i < q1;
    }
    suffix_scan();
    pv = ~0ULL;
  }
  emit(hq, on_q);
  // ---- the runs' pieces move together, in order; a run whose first kept value repeats the previous run's last one drops it
  const uint64_t has = __ballot(raw > 0);
  const uint64_t lower = has & ((1ULL << lane) - 1ULL);
  const int src = lower ? 63 - __clzll((unsigned long long)lower) : lane;
  const uint64_t prev_last = __shfl(last, src);
  const int drop = (raw > 0 && lower != 0 && first == prev_last) ? 1 : 0;
  const int mine = cnt - drop;
  int incl = mine, raw_all = raw;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t1 = __shfl_up(incl, off), t2 = __shfl_up(raw_all, off);
    if (lane >= off) {
      incl += t1;
      raw_all += t2;
    }
  }
  const int total = __builtin_amdgcn_readlane(incl, 63), raw_total = __builtin_amdgcn_readlane(raw_all, 63);
  if (raw_total <= raw_bound || raw_total > (int)HUGE_MIN) {  // the fused path does not apply (k1_windows_wave decides the same way)
    leave_to_wave_kernel();
    return;
  }
  // every lane moves its own piece (its stretch of hashes[] -> its place in scratch[]), eight values in flight: a loop over the 64
  // pieces with the whole wave copying one piece at a time was 64 dependent global round trips, ~100 us per read
  {
    uint64_t* __restrict__ fin = a.scratch + o1 + (incl - mine);
    const uint64_t* __restrict__ src = temp + drop;
    int most = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) most = max(most, __shfl_xor(most, off));
    for (int t = 0; t < most; t += 8) {
      uint64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = t + u < mine ? src[t + u] : 0;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (t + u < mine) fin[t + u] = v[u];
    }
  }
  if (lane == 0) {
    a.nk_adj[r] = total;
    a.nk_raw[r] = raw_total;
    a.nk1[r] = raw_total;
    a.qlen[r] = len;
  }
}

// halo too large for the LDS tiles (2k-s-1 > 512, w >= 511, k > 255): window scans over scratch arrays in global memory
__global__ void __launch_bounds__(K1WG) k1_kmers_wg_global(const K1Args a) {
  __shared__ uint64_t tab[256];
  __shared__ int s_wave[K1WG / 64];
  const int tid = threadIdx.x;
  if (tid < 256) tab[tid] = seed_of(tid);
  __syncthreads();
  wg_reads<false>(a, a.mode, tab, s_wave, nullptr, tid);
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

// The same segments with ROLLING hashes (round 4).  The prefix-XOR form above prices every k-mer at two 64-bit scans, four variable
// rotates and eight workgroup barriers per 1024 positions — ~150 lane-operations per base, 250 Gbase/s, which had become 42 % of a
// genome-search step.  A genome has what a read lacks: long runs.  Here every lane owns ROLL_L = 128 consecutive k-mer positions and
// rolls ntHash along them as the recurrence is meant to be used,
//     fh(i+1) = rol1(fh(i) ^ rol(F[i], k-1)) ^ F[i+k]        rh(i+1) = ror1(rh(i) ^ R[i] ^ rol(R[i+k], k)),
// after k steps of start-up: two byte look-ups and four seed look-ups (LDS) per k-mer and no cross-lane traffic at all.  A wave takes
// 64 x 128 = 8192 positions, a workgroup of 8 waves one K1SEG segment.  The bases of a wave's stretch are staged in LDS with the
// lanes' runs 132 bytes apart (the lanes walk their runs in step: a 128-byte pitch would put all of them on one bank).  Kept hashes
// must come out in position order, i.e. lane after lane: a first walk counts what each lane keeps (wave prefix sum, then the
// workgroup's over its 8 waves) and holds on to the first two hashes; a lane that kept more walks its run again to write them.  Same outputs as k1_seg_hash (out[], seg_cnt).
constexpr int ROLL_L = 128, ROLL_WAVES = 8, ROLL_PITCH = ROLL_L + 4, ROLL_HALO = 256;
static_assert(64 * ROLL_L * ROLL_WAVES == K1SEG, "a workgroup covers one segment");

__global__ void __launch_bounds__(64 * ROLL_WAVES) k1_seg_roll(const K1Args a) {
  // seeds, and the two rotated copies the recurrence asks for (a look-up instead of two variable 64-bit rotates per k-mer):
  // tab_out[b] = rol(F[b], k-1) leaves fh with the base that drops out, tab_in[b] = rol(R[b], k) enters rh with the new one
  __shared__ uint64_t tab[256], tab_out[256], tab_in[8];
  __shared__ __attribute__((aligned(16))) uint8_t bases[ROLL_WAVES][64 * ROLL_PITCH + ROLL_HALO];
  __shared__ int s_cnt[ROLL_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 256) {
    const uint64_t sd = seed_of(tid);
    tab[tid] = sd;
    tab_out[tid] = nt_tab_out(sd, a.k);
    if (tid < 8) tab_in[tid] = nt_tab_in(sd, a.k);
  }
  // As the fallback pass behind k1_seg_roll2 (seg_only_flagged) the launch is a small fixed grid that walks the LIST of segments that
  // kernel marked (a.seg_nflag entries of a.seg_list; none at all for a clean batch); alone, workgroup b takes segment b, once.
  const uint32_t n_todo = a.seg_only_flagged ? *a.seg_nflag : gridDim.x;
  for (uint32_t it = blockIdx.x; it < n_todo; it += gridDim.x) {
  const uint32_t bid = a.seg_only_flagged ? a.seg_list[it] : it;
  const uint32_t r = bid / a.segs_max, seg = bid % a.segs_max;
  const uint64_t o1 = a.offs[r];
  const int len = (int)(a.offs[r + 1] - o1);
  const int k = a.k;
  const int npos = len - k + 1;
  const int p_lo = (int)seg * K1SEG;
  const bool on = len >= a.min_qlen && p_lo < npos;  // (:778-786 gate; ErrShortSeq => no k-mers) — uniform over the workgroup
  const int P0 = p_lo + w * 64 * ROLL_L;             // first position of this wave
  const int wpos = on ? max(0, min(npos - P0, 64 * ROLL_L)) : 0;  // positions of this wave
  uint8_t* __restrict__ B = bases[w];
  auto at = [&](int q) -> int { return (q / ROLL_L) * ROLL_PITCH + (q % ROLL_L); };
  if (wpos > 0) {
    const uint8_t* __restrict__ s = a.seqs + o1 + P0;
    const int nb = wpos + k - 1;  // bases this wave needs (k <= 128, the launcher checks: they end inside the 65th run slot)
    // 16 bases per lane and turn (a 16-byte piece at a multiple of 16 never crosses the end of a 128-byte run, so it stays in one
    // piece in LDS too; the source is wherever the read starts: unaligned loads), then what is left byte by byte
    typedef uint32_t u32x4_any __attribute__((ext_vector_type(4), aligned(1)));
    const int nb16 = nb & ~15;
    for (int g = lane * 16; g < nb16; g += 64 * 16) {
      const u32x4_any v = *reinterpret_cast<const u32x4_any*>(s + g);
      uint32_t* d = reinterpret_cast<uint32_t*>(B + at(g));
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int g = nb16 + lane; g < nb; g += 64) B[at(g)] = s[g];
  }
  __syncthreads();  // tab[] and the staged bases
  const bool scaled = a.scaled != 0;
  const uint64_t max_hash = a.max_hash;
  const int q0 = lane * ROLL_L;
  const int mine = max(0, min(wpos - q0, ROLL_L));  // k-mer positions of this lane
  // one walk over the lane's run; `emit(h)` sees the kept hashes in position order
  auto walk = [&](auto&& emit) {
    if (mine <= 0) return;
    uint64_t fh = 0, rh = 0;
    for (int j = 0; j < k; j++) {  // start-up (hash_at): fh = XOR_j rol(F[j], k-1-j), rh = XOR_j rol(R[j], j)
      nt_start_step(fh, rh, B[at(q0 + j)], j, tab);
    }
    const uint8_t* __restrict__ run = B + lane * ROLL_PITCH;  // the lane's own 128 bases; what follows them starts 4 bytes later
    for (int t = 0;; t++) {
      const uint64_t h = fh < rh ? fh : rh;
      if (h != 0 && (!scaled || h <= max_hash)) emit(h);
      if (t + 1 >= mine) break;
      const int ti = t + k;  // < 256: k <= 128
      const uint8_t bo = run[t], bi = run[ti + ((ti >> 7) << 2)];
      nt_roll_step(fh, rh, bo, bi, tab, tab_out, tab_in);
    }
  };
  // (a FracMinHash database keeps one hash in hundreds: the first two a lane keeps stay in registers, and the second walk is only
  // taken by a lane that kept more — every lane of an unscaled query, hardly any of a scaled one)
  int c = 0;
  uint64_t h0 = 0, h1 = 0;
  walk([&](uint64_t h) {
    if (c == 0) h0 = h;
    else if (c == 1) h1 = h;
    c++;
  });
  // where this lane's hashes go: behind those of the lower lanes of its wave and of the lower waves of the workgroup
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_cnt[w] = incl;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int i = 0; i < ROLL_WAVES; i++) {
    const int ci = s_cnt[i];
    if (i < w) before += ci;
    total += ci;
  }
  if (c > 0) {
    uint64_t* __restrict__ out = a.scratch + o1 + p_lo + before + (incl - c);
    if (c <= 2) {
      out[0] = h0;
      if (c == 2) out[1] = h1;
    } else {
      int i = 0;
      walk([&](uint64_t h) { out[i++] = h; });
    }
  }
  if (tid == 0) a.seg_cnt[bid] = total;
  __syncthreads();  // bases[] and s_cnt[] are free for the next segment of the list
  }
}

// The same segments once more (round 5), on 2-BIT CODES.  k1_seg_roll above spends ~60 lane-operations per base, most of them on getting
// at its operands: two byte look-ups in LDS with their address arithmetic, four 64-bit table look-ups with theirs.  A genome is A, C, G, T
// almost everywhere, and for those four letters (either case) everything the recurrence needs is a function of two 2-bit codes:
//     fh' = rol1(fh) ^ F2[out][in],   F2[o][i] = rol1(rol(F[o], k-1)) ^ F[i]          rh' = ror1(rh ^ R2[out][in]),   R2[o][i] = R[o] ^ rol(R[i], k)
// Here a wave converts its stretch of bases to codes while it stages them (16 bases per lane and step: (w >> 1) & 3 per byte, one multiply
// folds four codes into a byte, one v_perm rebuilds the canonical letters to check that nothing else was there) into 2 KB of LDS instead
// of 8.4 KB of bytes; a lane then takes its codes 16 at a time from one dword, the incoming ones from a funnel shift of two (the
// offset k is the same for every group), and a roll costs two bit-field extracts, one table address, two look-ups and the 64-bit
// arithmetic: ~22 operations per base.  A segment that holds ANY other byte (N, IUPAC codes, U: their hashes depend on the exact byte)
// is left to k1_seg_roll: this kernel marks it (seg_cnt = -1) and the launcher runs the byte kernel behind it for the marked segments only.
// Same outputs as k1_seg_roll / k1_seg_hash (out[], seg_cnt): tests/test_gpu_parity.py::test_k1_all_forms_across_k and the fuzz sweeps.
// A lane walks R2_L = 256 consecutive positions (round 6; 128 before): what a lane pays once per run — the k start-up steps, its share of
// the scans and of the copy-out — is paid half as often, and a workgroup is 4 waves for the same segment and the same LDS.  512 (2 waves
// per workgroup: half the waves per SIMD) executes the same number of instructions 15 % slower; table indices from two bit-field shifts
// per roll instead of the nibble words: +5 % instructions (profiles/r06_roll2_min.txt).
#ifndef KMCPG_R2_L
#define KMCPG_R2_L 256
#endif
constexpr int R2_L = KMCPG_R2_L, R2_WAVES = K1SEG / (64 * R2_L), R2_G = R2_L / 16;  // positions per lane, waves per workgroup, code words per lane run
constexpr int R2_KEEP = R2_L > 256 ? 6 : R2_L > 128 ? 4 : 2;  // kept hashes a lane holds in registers (a FracMinHash lane keeps R2_L / scale of them: more is a second walk)
static_assert(64 * R2_L * R2_WAVES == K1SEG, "a workgroup covers one segment");
constexpr int R2_PITCH = R2_G + 1;             // dwords per lane run (+ 1: an odd pitch puts the lanes' j-th words on different banks)
constexpr int R2_RUNS = 64 + (9 + R2_G - 1) / R2_G;  // 64 runs + the 9 words that k - 1 <= 127 further bases and the funnel's upper word reach into
constexpr int R2_WORDS = R2_RUNS * R2_PITCH;

__global__ void __launch_bounds__(64 * R2_WAVES) k1_seg_roll2(const K1Args a) {
  // one block of LDS with the pair tables in front: their addresses — table index x 8 — then fit the offset fields of one ds_read2_b64
  // and the walk's inner loop needs no base address (an add per roll)
  struct Sh {
    uint64_t F2[16], R2[16];  // the two pair tables
    uint64_t S[4], RC[4];     // seeds by code (A 0, C 1, T 2, G 3 = (ascii >> 1) & 3), complements
    uint32_t codes[R2_WAVES][R2_WORDS];
    int s_cnt[R2_WAVES];
    int s_bad;
  };
  __shared__ __attribute__((aligned(16))) Sh sh;
  uint64_t* const F2 = sh.F2;
  uint64_t* const R2 = sh.R2;
  uint64_t* const S = sh.S;
  uint64_t* const RC = sh.RC;
  int* const s_cnt = sh.s_cnt;
  int& s_bad = sh.s_bad;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int k = a.k;
  if (tid < 16) {
    F2[tid] = nt2_f2(tid >> 2, tid & 3, k);
    R2[tid] = nt2_r2(tid >> 2, tid & 3, k);
    if (tid < 4) {
      S[tid] = seed_of(nt2_letter(tid));
      RC[tid] = seed_of(nt2_letter(tid) & 7);  // complement of base b = tab[b & 7] (nthash.hpp)
    }
  }
  if (tid == 0) s_bad = 0;
  const uint32_t r = blockIdx.x / a.segs_max, seg = blockIdx.x % a.segs_max;
  const uint64_t o1 = a.offs[r];
  const int len = (int)(a.offs[r + 1] - o1);
  const int npos = len - k + 1;
  const int p_lo = (int)seg * K1SEG;
  const bool on = len >= a.min_qlen && p_lo < npos;  // (:778-786 gate; ErrShortSeq => no k-mers) — uniform over the workgroup
  const int P0 = p_lo + w * 64 * R2_L;               // first position of this wave
  const int wpos = on ? max(0, min(npos - P0, 64 * R2_L)) : 0;  // positions of this wave
  uint32_t* __restrict__ Wd = sh.codes[w];
  auto slot = [&](int word) -> int { return (word / R2_G) * R2_PITCH + (word % R2_G); };  // word = base / 16 within the wave's stretch
  __syncthreads();  // s_bad = 0 before anybody raises it
  if (a.codes) {
    // the batch came as 2-bit codes: 16 of them are the 32 bits at bit 2 G of the packed stream (G = the word's first base in the batch) —
    // two dwords and a funnel shift; a foreign byte among the segment's bases has been noted per segment by k_mark_exc
    const uint32_t* __restrict__ C32 = reinterpret_cast<const uint32_t*>(a.codes);
    const uint64_t G0 = o1 + (uint64_t)P0;
    const int nb = wpos > 0 ? wpos + k - 1 : 0;  // bases this wave needs
    for (int gi = lane; gi < R2_G * R2_RUNS; gi += 64) {
      const int b0 = gi * 16;
      uint32_t word = 0;
      if (b0 < nb) {
        const uint64_t G = G0 + (uint64_t)b0;
        const uint64_t d = G >> 4;
        const uint32_t sh = 2u * (uint32_t)(G & 15u);
        const uint32_t lo = C32[d], hi = C32[d + 1];  // (the device copy of the codes ends 16 bytes behind the last base)
        word = nt2_funnel(hi, lo, sh);
        if (b0 + 16 > nb) word &= (1u << (2 * (nb - b0))) - 1u;  // what follows belongs to the next query
      }
      Wd[slot(gi)] = word;
    }
    if (tid == 0 && a.seg_exc && a.seg_exc[blockIdx.x]) s_bad = 1;
  } else {
    const uint8_t* __restrict__ s = a.seqs + o1 + P0;
    const int nb = wpos > 0 ? wpos + k - 1 : 0;  // bases this wave needs
    typedef uint32_t u32x4_any __attribute__((ext_vector_type(4), aligned(1)));
    bool bad = false;
    for (int gi = lane; gi < R2_G * R2_RUNS; gi += 64) {  // every word the walks may touch gets a value (zero past the stretch)
      const int b0 = gi * 16;
      uint32_t word = 0;
      if (b0 + 16 <= nb) {
        const u32x4_any v = *reinterpret_cast<const u32x4_any*>(s + b0);
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t c = nt2_codes4(in[d]);   // the code of each byte, in its byte
          bad |= !nt2_valid4(in[d], c);            // ... and back: anything but A/C/G/T in either case differs
          word |= nt2_fold4(c) << (8 * d);         // four codes into one byte
        }
      } else if (b0 < nb) {
        for (int j = 0; b0 + j < nb; j++) {
          const uint32_t ch = s[b0 + j], c = (ch >> 1) & 3u;
          bad |= (ch & 0xDFu) != (uint32_t)nt2_letter((int)c);
          word |= c << (2 * j);
        }
      }
      Wd[slot(gi)] = word;
    }
    if (__ballot(bad) != 0 && lane == 0) s_bad = 1;
  }
  __syncthreads();  // tables, codes, s_bad
  if (s_bad) {  // a byte that is not A/C/G/T: the byte kernel takes this segment
    if (tid == 0) {  // (-1 never reaches k1_seg_pack: the byte kernel, launched right behind over this list, overwrites it)
      a.seg_cnt[blockIdx.x] = -1;
      a.seg_list[atomicAdd(a.seg_nflag, 1u)] = blockIdx.x;
    }
    return;
  }
  const bool scaled = a.scaled != 0;
  const uint64_t max_hash = a.max_hash;
  const int q0 = lane * R2_L;
  const int mine = max(0, min(wpos - q0, R2_L));  // k-mer positions of this lane
  const int kw = k >> 4, ksh = 2 * (k & 15);
  const uint32_t mh_hi = (uint32_t)(max_hash >> 32);
  // SC = FracMinHash database (a constant inside the loops).  The filter `0 < h <= maxHash` (util-db-search.go:1072-1077) keeps one k-mer in
  // ~500 at scale 1000, so the 64-bit minimum and the two 64-bit compares of every step are replaced by a 32-bit test that a kept hash
  // must pass — the upper word of min(fh, rh) is min of the upper words — and only a step where some lane passes it computes h exactly.
  auto walk = [&](auto sc_tag, auto&& emit) __attribute__((always_inline)) {
    constexpr bool SC = decltype(sc_tag)::value;
    if (mine <= 0) return;
    // start-up: fh = XOR_j rol(F[j], k-1-j), rh = XOR_j rol(R[j], j) over the run's first k bases, a code word at a time; the reverse strand
    // is gathered as XOR_j ror(R[j], k-1-j) — one fixed rotate per base like the forward strand — and turned by k - 1 at the end
    uint64_t fh = 0, rh = 0;
    const int w_out = R2_G * lane;
    for (int j0 = 0; j0 < k; j0 += 16) {
      const uint32_t wd = Wd[slot(w_out + (j0 >> 4))];
      const int jn = min(16, k - j0);
      for (int j = 0; j < jn; j++) {
        const uint32_t c = (wd >> (2 * j)) & 3u;
        fh = nt2_rol1(fh) ^ S[c];
        rh = nt2_ror1(rh) ^ RC[c];
      }
    }
    rh = rolv(rh, k - 1);
    auto test = [&](uint64_t f, uint64_t r_) __attribute__((always_inline)) {
      if constexpr (SC) {
        const uint32_t fhi = (uint32_t)(f >> 32), rhi = (uint32_t)(r_ >> 32);
        uint32_t mn;  // (written out: the compiler folds the plain expression into the 64-bit compares below — six instructions for these two)
        asm("v_min_u32 %0, %1, %2" : "=v"(mn) : "v"(fhi), "v"(rhi));
        if (mn <= mh_hi) {
          const uint64_t h = f < r_ ? f : r_;
          if (h != 0 && h <= max_hash) emit(h);
        }
      } else {
        const uint64_t h = f < r_ ? f : r_;
        if (h != 0) emit(h);
      }
    };
    int t = 0;
    for (int g = 0; g < R2_G; g++) {
      const uint32_t outw = Wd[slot(w_out + g)];
      const uint32_t inw = nt2_funnel(Wd[slot(w_out + g + kw + 1)], Wd[slot(w_out + g + kw)], (uint32_t)ksh);
      // the table index of every roll of the group, a nibble each: (code going out << 2) | code coming in
      const uint32_t m[2] = {(nt2_spread(outw & 0xFFFFu) << 2) | nt2_spread(inw & 0xFFFFu), (nt2_spread(outw >> 16) << 2) | nt2_spread(inw >> 16)};
      if (mine - t >= 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          test(fh, rh);
          const uint32_t idx = (m[j >> 3] >> (4 * (j & 7))) & 15u;
          fh = nt2_rol1(fh) ^ F2[idx];
          rh = nt2_ror1(rh ^ R2[idx]);
        }
        t += 16;
        if (t >= mine) return;
      } else {
        for (int j = 0; t < mine; j++, t++) {
          test(fh, rh);
          const uint32_t idx = (m[j >> 3] >> (4 * (j & 7))) & 15u;
          fh = nt2_rol1(fh) ^ F2[idx];
          rh = nt2_ror1(rh ^ R2[idx]);
        }
        return;
      }
    }
  };
  int c = 0;
  uint64_t hk[R2_KEEP] = {};
  auto count2 = [&](uint64_t h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < R2_KEEP; i++) hk[i] = c == i ? h : hk[i];  // (selects, not branches into an array on the stack)
    c++;
  };
  if (scaled) walk(std::true_type{}, count2);
  else walk(std::false_type{}, count2);
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t_ = __shfl_up(incl, off);
    if (lane >= off) incl += t_;
  }
  if (lane == 63) s_cnt[w] = incl;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int i = 0; i < R2_WAVES; i++) {
    const int ci = s_cnt[i];
    if (i < w) before += ci;
    total += ci;
  }
  if (c > 0) {
    uint64_t* __restrict__ out = a.scratch + o1 + p_lo + before + (incl - c);
    if (c <= R2_KEEP) {
#pragma unroll
      for (int i = 0; i < R2_KEEP; i++)
        if (i < c) out[i] = hk[i];
    } else {
      int i = 0;
      auto store = [&](uint64_t h) __attribute__((always_inline)) { out[i++] = h; };
      if (scaled) walk(std::true_type{}, store);
      else walk(std::false_type{}, store);
    }
  }
  if (tid == 0) a.seg_cnt[blockIdx.x] = total;
}

// Packed whole-genome batches (a.codes): which segments hold a foreign byte, and the text of exactly those for the byte kernel.
// k_mark_exc: one thread per run of foreign bytes (positions count from the start of the batch; a run of equal bytes may continue from the
// end of one query into the next) -> seg_exc[(read, segment)] = 1 for every segment whose k-mers cover one of its bytes: segment g of a
// read owns the k-mer positions [g K1SEG, (g + 1) K1SEG), i.e. the bases [g K1SEG, (g + 1) K1SEG + k - 1).
__global__ void __launch_bounds__(256) k_mark_exc(const K1Args a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_exc) return;
  uint64_t pos = a.exc[i].pos;
  const uint64_t end = pos + a.exc[i].len;
  uint32_t lo = 0, hi = a.n_reads;  // the read r with offs[r] <= pos < offs[r + 1]: the last r with offs[r] <= pos
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.offs[mid] <= pos) lo = mid;
    else hi = mid;
  }
  uint32_t r = lo;
  while (pos < end && r < a.n_reads) {
    const uint64_t o1 = a.offs[r], o2 = a.offs[r + 1];
    if (pos >= o2) {  // (empty queries in between)
      r++;
      continue;
    }
    const uint64_t s = pos - o1, e = (end < o2 ? end : o2) - o1;  // the run's bases [s, e) of read r
    const uint64_t g_lo = s >= (uint64_t)(a.k - 1) ? (s - (uint64_t)(a.k - 1)) / K1SEG : 0, g_hi = (e - 1) / K1SEG;
    for (uint64_t g = g_lo; g <= g_hi && g < a.segs_max; g++) a.seg_exc[(uint64_t)r * a.segs_max + g] = 1u;
    pos = o2;
    r++;
  }
}

// the bases of the listed segments (k1_seg_roll2's list: the segments k_mark_exc marked) as text, four per packed byte and thread; a byte
// that straddles the segment's ends is expanded whole (its neighbours' bases get the value they have anyway); k_apply_exc behind this
// launch restores the foreign bytes, the byte kernel behind that reads the text
__global__ void __launch_bounds__(256) k_unpack2_list(const K1Args a) {
  constexpr uint32_t LUT = NT2_LETTERS;
  const uint32_t n_todo = *a.seg_nflag;
  for (uint32_t it = blockIdx.x; it < n_todo; it += gridDim.x) {
    const uint32_t bid = a.seg_list[it];
    const uint32_t r = bid / a.segs_max, seg = bid % a.segs_max;
    const uint64_t o1 = a.offs[r], o2 = a.offs[r + 1];
    const uint64_t g0 = o1 + (uint64_t)seg * K1SEG;
    uint64_t g1 = g0 + K1SEG + (uint64_t)(a.k - 1);
    if (g1 > o2) g1 = o2;
    if (g0 >= g1) continue;
    const uint64_t b0 = g0 >> 2, b1 = (g1 + 3) >> 2;  // packed bytes [b0, b1)
    for (uint64_t b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
      const uint32_t byte = a.codes[b];
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) v |= ((LUT >> (8 * ((byte >> (2 * j)) & 3u))) & 0xffu) << (8 * j);
      reinterpret_cast<uint32_t*>(a.seqs_w)[b] = v;
    }
  }
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
  const int cnt = sc[seg];  // >= 0: a segment k1_seg_roll2 marked -1 has been redone by the byte kernel before this launch
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



int k1_segment_len() { return K1SEG; }

// true when the kernel that ran drops the adjacent repeats itself (window sketches of long reads: scratch[] + nk_adj[])
bool launch_k1(const K1Args& a, uint32_t max_read_len, hipStream_t st) {
  if (a.n_reads == 0) return false;
  if (a.seg_cnt && a.segs_max > 1) {  // whole genomes: one workgroup per 65536-position segment, then an ordered pack
    const unsigned blocks = a.n_reads * a.segs_max;
    // rolling hashes for every k the staging halo holds (flags bit 3 = 8: the prefix-XOR form, for A/B runs)
    // (flags bit 4 = 16: the byte kernel alone, for A/B runs)
    if (a.k <= 128 && !(a.flags & 8) && !(a.flags & 16)) {
      // the list of segments the 2-bit kernel leaves to the byte kernel lives behind seg_cnt[] (run_kmers sizes it: counts, the counter, the list and — packed batches — a mark per segment: 3 * blocks + 2 words)
      K1Args b = a;
      b.seg_nflag = (uint32_t*)(a.seg_cnt + blocks);
      b.seg_list = b.seg_nflag + 1;
      (void)hipMemsetAsync(b.seg_nflag, 0, sizeof(uint32_t), st);
      if (a.codes) {  // packed batch (run_kmers sized seg_cnt[] for it: the marks live behind the list)
        b.seg_exc = nullptr;
        if (a.n_exc) {
          b.seg_exc = b.seg_list + blocks;
          (void)hipMemsetAsync(b.seg_exc, 0, (size_t)blocks * sizeof(uint32_t), st);
          hipLaunchKernelGGL(k_mark_exc, dim3((a.n_exc + 255) / 256), dim3(256), 0, st, b);
        }
        hipLaunchKernelGGL(k1_seg_roll2, dim3(blocks), dim3(64 * R2_WAVES), 0, st, b);
        if (a.n_exc) {  // (no foreign byte in the batch: nothing is on the list, nothing reads text)
          hipLaunchKernelGGL(k_unpack2_list, dim3(std::min(blocks, 512u)), dim3(256), 0, st, b);
          launch_apply_exc(a.exc, a.n_exc, a.seqs_w, st);
          b.seg_only_flagged = 1;
          hipLaunchKernelGGL(k1_seg_roll, dim3(std::min(blocks, 512u)), dim3(64 * ROLL_WAVES), 0, st, b);
        }
        hipLaunchKernelGGL(k1_seg_pack, dim3(blocks), dim3(256), 0, st, a);
        return false;
      }
      hipLaunchKernelGGL(k1_seg_roll2, dim3(blocks), dim3(64 * R2_WAVES), 0, st, b);  // 2-bit codes; lists the segments it cannot take
      b.seg_only_flagged = 1;
      // ... and the byte kernel does those: a grid that fills the chip once (2 workgroups of 8 waves per CU) walks the list — nothing
      // but the read of one counter for a clean batch, where a launch over all segments was up to 2^21 workgroups exiting at once
      hipLaunchKernelGGL(k1_seg_roll, dim3(std::min(blocks, 512u)), dim3(64 * ROLL_WAVES), 0, st, b);
    } else if (a.k <= 128 && !(a.flags & 8)) hipLaunchKernelGGL(k1_seg_roll, dim3(blocks), dim3(64 * ROLL_WAVES), 0, st, a);
    else hipLaunchKernelGGL(k1_seg_hash, dim3(blocks), dim3(K1WG), 0, st, a);
    hipLaunchKernelGGL(k1_seg_pack, dim3(blocks), dim3(256), 0, st, a);
    return false;
  }
  if (max_read_len > 2048) {  // long queries: a whole workgroup per read
    unsigned blocks = a.n_reads > 65536 ? 65536 : a.n_reads;
    if (a.mode != 0 && a.scratch && wave_windows_usable(a) && !(a.flags & 4)) {  // window sketches: the barrier-free form
      // closed syncmers with a window of 12 / 16 / 20 / 24 / 32 s-mers (k - s = 6 .. 16), single-end, the fused adjacent-repeat path wanted:
      // the rolling kernel first, k1_windows_wave behind it for the reads it leaves on its list (flags bit 5 = 32: the old kernel alone)
      const int wsz = a.mode == 2 ? 2 * (a.k - (int)a.w_or_s) : 0;
      const int words = wr_words_for((int)max_read_len);
      // reads (= waves) per workgroup: LDS per wave (k-mer ring 8 KB + 2-bit codes of the longest read) decides how many waves a CU holds —
      // two per SIMD for 20-kb reads whatever the grouping (tools/ubench_lds_occ.cpp); 2 measured 0.7 % ahead of 4 and 1
      int waves = getenv("KMCPG_WR_WAVES") ? atoi(getenv("KMCPG_WR_WAVES")) : 2;
      if (waves != 1 && waves != 4) waves = 2;
      while (waves > 1 && wr_lds_bytes(wsz, words, waves) > 65536) waves >>= 1;
      // (no read of the batch can exceed the -u / wave-sort bound — planting, a huge -u —: the fused path is nobody's, the old kernel alone)
      const bool any_fused = (long long)max_read_len > (long long)std::max(a.dedup_threshold, K1_WAVE_SORT_CAP);
      const bool wsz_ok = wsz == 12 || wsz == 16 || wsz == 20 || wsz == 24 || wsz == 32;  // k - s = 6, 8, 10, 12, 16 (21/11, 31/15, 21/13, 31/19 ...)
      if (a.mode == 2 && wsz_ok && a.k <= 64 && !a.offs2 && a.nk_adj && a.seg_list && !(a.flags & 32) && any_fused &&
          wr_lds_bytes(wsz, words, waves) <= 65536) {
        K1Args b = a;
        (void)hipMemsetAsync(b.seg_nflag, 0, sizeof(uint32_t), st);
        const unsigned wg = (a.n_reads + waves - 1) / waves;
        const size_t lds = wr_lds_bytes(wsz, words, waves);
#define KMCPG_WR_LAUNCH(WSZ_, WV_) hipLaunchKernelGGL((k1_windows_roll<WSZ_, WV_>), dim3(wg), dim3(64 * WV_), lds, st, b, words)
#define KMCPG_WR_WAVES_OF(WSZ_)                \
  do {                                         \
    if (waves == 4) KMCPG_WR_LAUNCH(WSZ_, 4);  \
    else if (waves == 2) KMCPG_WR_LAUNCH(WSZ_, 2); \
    else KMCPG_WR_LAUNCH(WSZ_, 1);             \
  } while (0)
        switch (wsz) {
          case 12: KMCPG_WR_WAVES_OF(12); break;
          case 16: KMCPG_WR_WAVES_OF(16); break;
          case 20: KMCPG_WR_WAVES_OF(20); break;
          case 24: KMCPG_WR_WAVES_OF(24); break;
          default: KMCPG_WR_WAVES_OF(32); break;
        }
#undef KMCPG_WR_WAVES_OF
#undef KMCPG_WR_LAUNCH
        b.seg_only_flagged = 1;
        hipLaunchKernelGGL(k1_windows_wave<2>, dim3(std::min(blocks, 1024u)), dim3(K1W_THREADS), 0, st, b);
        if (getenv("KMCPG_K1_DEBUG")) {  // how many reads the rolling kernel left to k1_windows_wave
          uint32_t nf = 0;
          (void)hipStreamSynchronize(st);
          (void)hipMemcpy(&nf, b.seg_nflag, sizeof nf, hipMemcpyDeviceToHost);
          fprintf(stderr, "k1_windows_roll<%d>: %u of %u reads left to k1_windows_wave (max_read_len %u, %d words)\n", wsz, nf, a.n_reads, max_read_len, words);
        }
        return true;
      }
      if (a.mode == 2) hipLaunchKernelGGL(k1_windows_wave<2>, dim3(blocks), dim3(K1W_THREADS), 0, st, a);
      else hipLaunchKernelGGL(k1_windows_wave<1>, dim3(blocks), dim3(K1W_THREADS), 0, st, a);
      return a.nk_adj != nullptr;
    }
    if (!wg_lds_usable(a)) hipLaunchKernelGGL(k1_kmers_wg_global, dim3(blocks), dim3(K1WG), 0, st, a);
    else if (a.mode == 2) hipLaunchKernelGGL(k1_kmers_wg<2>, dim3(blocks), dim3(K1WG), 0, st, a);
    else if (a.mode == 1) hipLaunchKernelGGL(k1_kmers_wg<1>, dim3(blocks), dim3(K1WG), 0, st, a);
    else hipLaunchKernelGGL(k1_kmers_wg<0>, dim3(blocks), dim3(K1WG), 0, st, a);
    return wg_lds_usable(a) && a.mode != 0 && a.nk_adj && a.scratch && (a.flags & 2);
  }
  unsigned blocks = (a.n_reads + 3) / 4;
  if (blocks > 32768) blocks = 32768;
  if (a.mode == 2) hipLaunchKernelGGL(k1_kmers<2>, dim3(blocks), dim3(256), 0, st, a);
  else if (a.mode == 1) hipLaunchKernelGGL(k1_kmers<1>, dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k1_kmers<0>, dim3(blocks), dim3(256), 0, st, a);
  return false;
}

void launch_nk_simple(const int32_t* nk_raw, int32_t* nk_search, uint32_t n, int32_t min_matched, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_nk_simple, dim3((n + 255) / 256), dim3(256), 0, st, nk_raw, nk_search, n, min_matched);
}

}  // namespace kmcpg
