// k1_dedup.hip — K1d: per-read sort + unique when #k-mers > -u (kmcp/cmd/util-db-search.go:874-908).  Whole genomes
// (> 65536 k-mers) go to sort_huge.hip.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.hpp"
#include "device_utils.hpp"
#include "kernels.hpp"

namespace kmcpg {

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
constexpr int DW_CAP = DEDUP_WAVE_CAP;

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


// Queries of up to 2048 elements to sort (a 10-kb HiFi read leaves ~1 400 after the adjacent-repeat filter): the bitonic network
// above needs 66 barrier-separated stages for them; the hashes are uniform, so one distribution pass gets them almost sorted:
// 1024 buckets by the top bits (after shifting the largest value of the query up to bit 63, which also covers FracMinHash
// databases whose hashes all lie below maxHash), a scan of the bucket sizes, a scatter, and an insertion sort of every bucket
// (2 elements on average) — ten barriers in all, same output as the sort (ascending, then unique).  A query whose values are
// not spread out (some bucket above 24 elements: low-complexity sequence) takes the bitonic network instead, in this kernel.
constexpr int DB_MAXB = 24;
template <int CAP, int NB, int DB_NT = 256>
__global__ void __launch_bounds__(DB_NT) k_dedup_bucket(const DedupArgs a) {
  constexpr int BPT = NB / DB_NT;  // buckets per thread
  constexpr int SHIFT = 64 - (NB == 1024 ? 10 : 11);
  static_assert(NB == 1024 || NB == 2048, "bucket bits");
  static_assert(NB % DB_NT == 0 && BPT >= 1, "whole buckets per thread");
  __shared__ uint64_t o[CAP];
  __shared__ int cnt[NB];
  __shared__ int st[NB + 1];
  __shared__ int scan[DB_NT / 64];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (uint32_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
    const int n = a.nk_raw[r];
    if (n <= a.dedup_threshold || n <= a.n_lo || n > a.n_hi) continue;
    const int m = a.pre ? a.nk_search[r] : n;
    if (m <= a.lo || m > a.hi) continue;
    const uint64_t koff = a.offs[r] + (a.offs2 ? a.offs2[r] : 0);
    uint64_t* g = a.hashes + koff;
    const uint64_t* __restrict__ in = a.pre ? a.scratch + koff : g;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < NB; i += DB_NT) cnt[i] = 0;
    __syncthreads();
    // a thread's elements (<= CAP / DB_NT of them) are fetched once, all loads in flight together, and both passes work from registers
    // (round 6: one exposed memory round trip per query instead of one per element and pass); LDS holds the sorted copy only.
    // lz: the hashes are uniform over [0, maxHash] (FracMinHash) or the whole 64-bit range; the host passes the shift that
    // brings the top of that range to bit 63
    const int lz = a.key_shift;
    constexpr int EPT = CAP / DB_NT;
    uint64_t v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const int i = tid + e * DB_NT;
      v[e] = i < m ? in[i] : 0;
    }
#pragma unroll
    for (int e = 0; e < EPT; e++)
      if (tid + e * DB_NT < m) atomicAdd(&cnt[(int)((v[e] << lz) >> SHIFT)], 1);
    __syncthreads();
    // exclusive scan of the bucket sizes (BPT buckets per thread)
    int c4[BPT], sum = 0, big = 0;
#pragma unroll
    for (int j = 0; j < BPT; j++) {
      c4[j] = cnt[tid * BPT + j];
      sum += c4[j];
      big |= c4[j] > DB_MAXB;
    }
    if (big) s_bad = 1;
    const int incl = wave_add_scan(sum);
    if (lane == 63) scan[w] = incl;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int j = 0; j < DB_NT / 64; j++)
      if (j < w) before += scan[j];
    int pos = before + incl - sum;
#pragma unroll
    for (int j = 0; j < BPT; j++) {
      st[tid * BPT + j] = pos;
      cnt[tid * BPT + j] = pos;  // the scatter's cursor
      pos += c4[j];
    }
    if (tid == DB_NT - 1) st[NB] = pos;
    const bool bad = s_bad != 0;
    __syncthreads();
    if (!bad) {
#pragma unroll
      for (int e = 0; e < EPT; e++)
        if (tid + e * DB_NT < m) o[atomicAdd(&cnt[(int)((v[e] << lz) >> SHIFT)], 1)] = v[e];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < BPT; j++) {  // insertion sort of this thread's buckets
        const int b0 = st[tid * BPT + j], b1 = st[tid * BPT + j + 1];
        for (int i = b0 + 1; i < b1; i++) {
          const uint64_t x = o[i];
          int q = i - 1;
          while (q >= b0 && o[q] > x) {
            o[q + 1] = o[q];
            q--;
          }
          o[q + 1] = x;
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int e = 0; e < EPT; e++)
        if (tid + e * DB_NT < m) o[tid + e * DB_NT] = v[e];
      __syncthreads();
      bitonic_sort(o, m, tid, DB_NT);
    }
    const int total = block_unique<DB_NT>(o, m, g, scan, tid);
    // MinMatched is tested on the raw count (:854), NumKmers is the unique count (:910)
    if (tid == 0) a.nk_search[r] = n >= a.min_matched ? total : 0;
    __syncthreads();
  }
}

void launch_dedup(DedupArgs a, uint64_t max_n, hipStream_t st) {
  if (a.n_reads == 0) return;
  // the wave class settles every query at or below the dedup threshold and sorts those of at most 512 k-mers
  hipLaunchKernelGGL(k_dedup_wave, dim3((a.n_reads + 3) / 4), dim3(256), 0, st, a);
  if (max_n <= DW_CAP) return;
  a.n_lo = DW_CAP;
  a.n_hi = max_n > HUGE_MIN ? (int32_t)HUGE_MIN : 0x7fffffff;  // beyond that: device-wide sort (sort_huge.hip)
  const unsigned grid = a.n_reads > (1u << 20) ? (1u << 20) : a.n_reads;  // the workgroup kernels stride over the reads
  if (a.pre && !a.pre_done) hipLaunchKernelGGL(k_adj_unique, dim3(grid), dim3(ADJ_NT), 0, st, a);
  // classes by the number m of elements to sort; the grids stride over the reads, sized for what the chip holds at once
  a.lo = 0;
  a.hi = 2048;
  hipLaunchKernelGGL((k_dedup_bucket<2048, 1024>), dim3(std::min(grid, 256u * 12)), dim3(256), 0, st, a);
  if (max_n <= 2048) return;
  a.lo = 2048;  // (a query the class before finished shows its NumKmers <= lo here)
  a.hi = 4096;
  hipLaunchKernelGGL((k_dedup_bucket<4096, 2048>), dim3(std::min(grid, 256u * 6)), dim3(256), 0, st, a);
  if (max_n > 4096) {
    // FracMinHash sketches of whole genomes (4 000 - 16 000 hashes, uniform below maxHash): the distribution pass again, 1024 threads and
    // 144 KB of LDS per query (one workgroup per CU) instead of the bitonic network's 105 barrier-separated stages
    static const bool big_bucket = !(getenv("KMCPG_DEDUP_BIG_BUCKET") && atoi(getenv("KMCPG_DEDUP_BIG_BUCKET")) == 0);
    a.lo = 4096;
    if (big_bucket) {
      a.hi = 16384;
      hipLaunchKernelGGL((k_dedup_bucket<16384, 2048, 1024>), dim3(std::min(grid, 256u)), dim3(1024), 0, st, a);
      if (max_n <= 16384) return;
      a.lo = 16384;
    }
    a.hi = 0x7fffffff;
    hipLaunchKernelGGL((k_dedup<1024, 16384>), dim3(std::min(grid, 1024u)), dim3(1024), 0, st, a);
  }
}

}  // namespace kmcpg
