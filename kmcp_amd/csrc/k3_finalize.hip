// k3_finalize.hip — K3: the hit list of a batch grouped by read, filtered by -T and ordered per read ON THE DEVICE.
//
// Reference counterpart: the tail of the UnikIndex worker (the target-coverage test, util-db-search.go:7471-7473) and the per-query
// sort of handleQuerySingleDB (:260-283, Matches.Less / SortByTCov / SortByJacc :105-145).  K2 emits (read, column, count) tuples in
// no particular order; until round 4 the host partitioned them by read, filtered and sorted them (finalize.cpp) — on databases full of
// close relatives (hundreds of matches per read) that host half was 3x the time of the kernels.  Here:
//   k3_count    surviving hits (count / size >= -T in float64, the reference's own division) per read: one atomic per RUN of hits
//               of one read (K2 emits them side by side)
//   k3_scan_*   exclusive scan of the counters -> CSR offsets of the reads (uint64)
//   k3_scatter  (column, count) pairs into their read's segment
//   k3_sort_*   every segment ordered as the reference orders a query's matches; a wave per 16 reads for segments of up to 512
//               matches (bitonic network in a wave-private LDS tile, no workgroup barrier), one workgroup for up to 4096; longer segments are left to
//               the host (kmcpg_finalize_grouped knows the same constant).
// What goes back over PCIe is 8 bytes per match in final order plus the offsets; the host only expands pairs to Match records.
//
// Sort keys (ascending order of the 128-bit key (A, B) = the reference's order; ties broken by column, as finalize.cpp does):
//   -s qcov: qcov = c / n with one n per read, so the order of qcov is the order of the integer c (two counts that differ give
//            quotients far more than an ulp apart); ties by tcov = c / size descending = size ascending (equal c; sizes below 2^52
//            give distinct quotients), then column:            A = ~c : size[63:32]      B = size[31:0] : column
//   -s tcov: (c / size) descending as float64 bit patterns (positive doubles order like their bits), then c descending, column:
//                                                              A = ~bits(c / size)       B = ~c : column
//   -s jacc: c / (n + size - c) likewise:                      A = ~bits(jacc)           B = ~c : column
//   -S     : column order (this build's deterministic stand-in for the reference's arrival order): A = column, B = ~c : column
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "device_utils.hpp"
#include "k3_keys.hpp"
#include "kernels.hpp"

namespace kmcpg {

namespace {

constexpr int SCAN_ITEMS = 16, SCAN_THREADS = 256, SCAN_TILE = SCAN_ITEMS * SCAN_THREADS;

__device__ __forceinline__ uint64_t n_hits_of(const K3Args& a) {
  const unsigned long long n = *a.n_hits;
  return n < a.hit_cap ? n : a.hit_cap;
}

// the -T test exactly as the reference makes it: float64(count) / float64(size) >= minTCov (:7471-7473)
__device__ __forceinline__ bool passes(const K3Args& a, const kmcpg_hit& h) {
  if (h.read >= a.n_reads || h.col >= a.n_cols) return false;  // counted in k3_count
  return a.min_tcov <= 0.0 || passes_tcov(h.count, a.col_size[h.col], a.min_tcov);
}

// Hits come in runs: K2 emits the hits of one (read, slot) unit side by side (k2_cobs.hip, wave-aggregated emission), so 64
// consecutive hits name a handful of reads — dozens of hits each on a database of close relatives.  A wave therefore finds the
// runs of equal reads among its 64 hits (ballots), and only the first lane of a run touches the read's counter, with the length of
// the run: 26.6 M hits of a 203-matches-per-read batch cost ~4 M atomics instead of 26.6 M (twice: count, then scatter).
struct Run {
  bool pass;      // this lane's hit takes part
  bool head;      // ... and is the first of its run
  int hpos;       // lane of the run's first hit
  uint32_t len;   // hits in the run (valid on every lane of it)
};
__device__ __forceinline__ Run find_run(bool pass, uint32_t read, int lane) {
  const uint32_t key = pass ? read : 0xffffffffu;  // (a read index is below 2^32 - 1: n_reads is a uint32)
  const uint32_t prev = __shfl_up(key, 1);
  Run r;
  r.pass = pass;
  r.head = pass && (lane == 0 || key != prev);
  const uint64_t H = __ballot(r.head), P = __ballot(pass);
  const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  const uint64_t hm = H & upto;
  r.hpos = hm ? 63 - __clzll((long long)hm) : 0;
  // the run ends before the next head or the next hit that does not take part
  const uint64_t above = (r.hpos == 63) ? 0ull : ((H | ~P) & ~((2ull << r.hpos) - 1ull));
  const int end = above ? __ffsll((long long)above) - 1 : 64;
  r.len = (uint32_t)(end - r.hpos);
  return r;
}

__global__ void k3_count(K3Args a) {
  const uint64_t n = n_hits_of(a);
  const int lane = threadIdx.x & 63;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += step) {  // wave-uniform trip count
    const uint64_t i = i0 + (uint64_t)lane;
    kmcpg_hit h{0xffffffffu, 0, 0};
    if (i < n) h = a.hits[i];
    if (i < n && (h.read >= a.n_reads || h.col >= a.n_cols) && !(h.read == 0xffffffffu && h.col == 0xffffffffu)) atomicAdd(a.bad, 1u);  // (tombstones of K2 are skipped)
    const Run r = find_run(i < n && passes(a, h), h.read, lane);
    if (r.head) atomicAdd(&a.cnt[h.read], r.len);
  }
}

// exclusive scan of cnt[0 .. n) into offs, in three launches: tile-local scans + tile totals, the scan of the totals, the add
__global__ void __launch_bounds__(SCAN_THREADS) k3_scan_tiles(const uint32_t* __restrict__ cnt, uint64_t* __restrict__ offs, uint64_t* __restrict__ sums, uint32_t n) {
  __shared__ uint64_t wsum[SCAN_THREADS / 64];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t t = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = base + i < n ? cnt[base + i] : 0u;
    t += v[i];
  }
  // inclusive scan of the threads' totals: within the wave by shuffles, across the 4 waves through LDS
  uint64_t inc = t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint64_t before = 0;
  for (int w = 0; w < wave; w++) before += wsum[w];
  uint64_t run = before + inc - t;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) offs[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == SCAN_THREADS - 1) sums[blockIdx.x] = before + inc;
}

__global__ void __launch_bounds__(SCAN_THREADS) k3_scan_sums(uint64_t* __restrict__ sums, uint32_t n_tiles) {
  __shared__ uint64_t wsum[SCAN_THREADS / 64];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t b0 = 0; b0 < n_tiles; b0 += SCAN_THREADS) {
    const uint32_t i = b0 + threadIdx.x;
    const uint64_t t = i < n_tiles ? sums[i] : 0;
    uint64_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t before = carry_s;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (i < n_tiles) sums[i] = before + inc - t;
    __syncthreads();
    if (threadIdx.x == SCAN_THREADS - 1) carry_s = before + inc;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(SCAN_THREADS) k3_scan_add(uint64_t* __restrict__ offs, const uint64_t* __restrict__ sums, uint32_t n) {
  const uint64_t add = sums[blockIdx.x];
  const uint32_t base = blockIdx.x * SCAN_TILE;
  for (uint32_t i = threadIdx.x; i < SCAN_TILE && base + i < n; i += SCAN_THREADS) offs[base + i] += add;
}

// segments are filled from the back (the counter of a read runs down to zero): the order inside a segment is settled by the sort
__global__ void k3_scatter(K3Args a) {
  const uint64_t n = n_hits_of(a);
  const int lane = threadIdx.x & 63;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += step) {
    const uint64_t i = i0 + (uint64_t)lane;
    kmcpg_hit h{0xffffffffu, 0, 0};
    if (i < n) h = a.hits[i];
    const Run r = find_run(i < n && passes(a, h), h.read, lane);
    uint32_t left = 0;
    if (r.head) left = atomicSub(&a.cnt[h.read], r.len);  // the run takes places left - len .. left - 1 of its read's segment
    left = __shfl(left, r.hpos);
    if (r.pass) a.pairs[a.offs[h.read] + left - 1 - (uint32_t)(lane - r.hpos)] = kmcpg_pair{h.col, h.count};
  }
}

// one compare-exchange step of the bitonic network over P keys in `t`, done by `nthreads` threads (tid = this thread's index)
__device__ __forceinline__ void bitonic_step(Key* t, uint32_t P, uint32_t k, uint32_t j, uint32_t tid, uint32_t nthreads) {
  for (uint32_t q = tid; q < P / 2; q += nthreads) {
    const uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
    const uint32_t l = i | j;
    const Key x = t[i], y = t[l];
    const bool up = (i & k) == 0;
    if (key_less(y, x) == up) {
      t[i] = y;
      t[l] = x;
    }
  }
}

constexpr int WAVE_CAP = K3_WAVE_CAP, WG_CAP = K3_WG_CAP;

// One wave per RPW consecutive reads, 4 waves per workgroup: segments of 2 .. WAVE_CAP matches are sorted one after the other.
// (One wave per READ made the common batch — a million reads with 0 or 1 match each — pay a wave launch and two dependent
// loads per read for nothing to do: 0.5 ms of a 22 ms step.  Here a wave fetches the offsets of its RPW reads with one
// coalesced load and skips the short ones by ballot.)
constexpr int RPW = 16;
__global__ void __launch_bounds__(256) k3_sort_wave(K3Args a) {
  __shared__ Key tile[4][WAVE_CAP];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint64_t r0 = ((uint64_t)blockIdx.x * 4 + wave) * RPW;
  if (r0 >= a.n_reads) return;
  const uint64_t mine = a.offs[min(r0 + lane, (uint64_t)a.n_reads)];  // lanes 0..RPW hold offs[r0 + lane]
  Key* t = tile[wave];
  for (int q = 0; q < RPW && r0 + q < a.n_reads; q++) {
    const uint64_t s0 = __shfl(mine, q), s1 = __shfl(mine, q + 1);
    const uint64_t m = s1 - s0;
    if (m < 2 || m > WAVE_CAP) continue;  // wave-uniform
    uint32_t P = 2;
    while (P < m) P <<= 1;
    const double nh = (double)a.nk[r0 + q];
    for (uint32_t i = lane; i < P; i += 64) t[i] = i < m ? make_key(a.sort_mode, a.col_size, a.pairs[s0 + i], nh) : Key{~0ull, ~0ull};
    wave_lds_fence();
    for (uint32_t k = 2; k <= P; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        bitonic_step(t, P, k, j, lane, 64);
        wave_lds_fence();
      }
    for (uint32_t i = lane; i < m; i += 64) a.pairs[s0 + i] = pair_of(a.sort_mode, t[i]);
    wave_lds_fence();  // the tile is reused by the next read
  }
}

// Segments of WAVE_CAP < matches <= WG_CAP: one workgroup sorts one of them in LDS.  They are rare, so the workgroups first LOOK
// for them 256 reads at a time (one coalesced load of the offsets per thread; a walk of one read per iteration with its two dependent
// loads made a million-read batch wait 0.5 ms for nothing) and sort what they find.
__global__ void __launch_bounds__(256) k3_sort_wg(K3Args a) {
  __shared__ Key big[WG_CAP];  // 64 KB of the CU's 160 KB
  __shared__ uint32_t n_found, found[256];
  for (uint64_t r0 = (uint64_t)blockIdx.x * 256; r0 < a.n_reads; r0 += (uint64_t)gridDim.x * 256) {
    if (threadIdx.x == 0) n_found = 0;
    __syncthreads();
    const uint64_t rr = r0 + threadIdx.x;
    if (rr < a.n_reads) {
      const uint64_t mm = a.offs[rr + 1] - a.offs[rr];
      if (mm > WAVE_CAP && mm <= WG_CAP) found[atomicAdd(&n_found, 1u)] = threadIdx.x;
    }
    __syncthreads();
    const uint32_t nf = n_found;
    for (uint32_t f = 0; f < nf; f++) {
      const uint64_t r = r0 + found[f];
      const uint64_t s0 = a.offs[r], s1 = a.offs[r + 1];
      const uint64_t m = s1 - s0;
      uint32_t P = 2;
      while (P < m) P <<= 1;
      const double nh = (double)a.nk[r];
      for (uint32_t i = threadIdx.x; i < P; i += 256) big[i] = i < m ? make_key(a.sort_mode, a.col_size, a.pairs[s0 + i], nh) : Key{~0ull, ~0ull};
      __syncthreads();
      for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
          bitonic_step(big, P, k, j, threadIdx.x, 256);
          __syncthreads();
        }
      for (uint32_t i = threadIdx.x; i < m; i += 256) a.pairs[s0 + i] = pair_of(a.sort_mode, big[i]);
      __syncthreads();
    }
    __syncthreads();
  }
}

}  // namespace

uint32_t k3_scan_tiles_for(uint32_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

// cnt[0 .. n_reads] must be zero on entry (it is again on exit); offs gets n_reads + 1 entries, offs[n_reads] = matches kept
void launch_k3(const K3Args& a, uint64_t hits_hint, hipStream_t st) {
  if (a.n_reads == 0) return;
  const uint64_t work = hits_hint ? hits_hint : a.hit_cap;
  const unsigned gblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((work + 255) / 256, 16384));
  hipLaunchKernelGGL(k3_count, dim3(gblocks), dim3(256), 0, st, a);
  const uint32_t n = a.n_reads + 1, tiles = k3_scan_tiles_for(n);
  hipLaunchKernelGGL(k3_scan_tiles, dim3(tiles), dim3(SCAN_THREADS), 0, st, a.cnt, a.offs, a.sums, n);
  hipLaunchKernelGGL(k3_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, a.sums, tiles);
  hipLaunchKernelGGL(k3_scan_add, dim3(tiles), dim3(SCAN_THREADS), 0, st, a.offs, a.sums, n);
  hipLaunchKernelGGL(k3_scatter, dim3(gblocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k3_sort_wave, dim3((a.n_reads + 4 * RPW - 1) / (4 * RPW)), dim3(256), 0, st, a);
  const unsigned wgb = (unsigned)std::min<uint32_t>((a.n_reads + 255) / 256, 2048);
  hipLaunchKernelGGL(k3_sort_wg, dim3(wgb), dim3(256), 0, st, a);
}

}  // namespace kmcpg
