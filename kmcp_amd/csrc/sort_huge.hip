// sort_huge.hip — sort + unique (handleQuery, kmcp/cmd/util-db-search.go:874-908) for queries with more than 65 536 k-mers
// (whole genomes under -g): a device-wide LSD radix sort (8 passes of 8 bits over the 64-bit hashes) and an adjacent-unique
// pass.  The per-read workgroup kernels in k1_dedup.hip cover everything smaller; this file exists so that one 5-M-k-mer query
// does not run on a single workgroup.
//
// Everything is wave-granular: wave w owns the 4096 consecutive keys [4096 w, 4096 (w+1)) of the current buffer and walks them
// in 64 rounds of 64, so "earlier in the buffer" is (wave, round, lane) order and every pass is stable without any
// cross-wave coordination:
//   k_rs_hist     per-wave digit histogram (LDS atomics)                         -> hist[digit][wave]
//   scan_u32      exclusive scan of that table in (digit, wave) order (tile sums, scan of the sums, scan inside the tiles)
//   k_rs_scatter  the wave re-reads its keys; lanes with equal digits find each other with 8 ballots, rank = popcount of the
//                 lower lanes in the group, destination = scanned base + keys of that digit the wave has already placed
// unique: k_uq_count (heads per wave) -> k_scan_u32 -> k_uq_scatter.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_utils.hpp"
#include "kernels.hpp"

namespace kmcpg {

namespace {
constexpr uint32_t KEYS_PER_WAVE = 4096;  // 64 rounds of 64: a 5-M-key query is 1221 waves, its histogram table 312 k entries
constexpr int ROUNDS = KEYS_PER_WAVE / 64;

inline uint32_t n_waves(uint32_t n) { return (n + KEYS_PER_WAVE - 1) / KEYS_PER_WAVE; }
}  // namespace

size_t huge_dedup_temp_bytes(uint32_t max_n) {
  const size_t table = (size_t)256 * n_waves(max_n);  // histogram table + the per-wave head counts + the scan's tile sums
  return (table + n_waves(max_n) + table / 4096 + 64) * sizeof(uint32_t) + 256;
}

__global__ void __launch_bounds__(256) k_rs_hist(const uint64_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t nw) {
  __shared__ uint32_t cnt[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t w = blockIdx.x * 4 + wv;
  for (int d = lane; d < 256; d += 64) cnt[wv][d] = 0;
  wave_lds_fence();
  if (w < nw) {
    const uint64_t base = (uint64_t)w * KEYS_PER_WAVE;
    for (int r = 0; r < ROUNDS; r++) {
      const uint64_t i = base + (uint64_t)r * 64 + lane;
      if (i < n) atomicAdd(&cnt[wv][(keys[i] >> shift) & 255], 1u);
    }
    wave_lds_fence();
    for (int d = lane; d < 256; d += 64) hist[(size_t)d * nw + w] = cnt[wv][d];
  }
}

// ---- exclusive scan of a u32 array in place, three launches: sums of 4096-element tiles -> scan of the tile sums (one
// workgroup) -> scan inside every tile seeded with its sum.  A thread owns 16 consecutive elements (one 64-byte line).
constexpr uint32_t SCAN_TILE = 4096;

__device__ __forceinline__ uint32_t wg_exclusive_scan_256(uint32_t v, uint32_t* lds /*[4]*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(inc, off);
    if (lane >= off) inc += u;
  }
  if (lane == 63) lds[wv] = inc;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t t = lds[w];
    if (w < wv) base += t;
    all += t;
  }
  __syncthreads();  // lds is reused by the caller's next round
  if (total) *total = all;
  return base + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t* __restrict__ data, uint32_t total, uint32_t* __restrict__ tile_sum) {
  __shared__ uint32_t lds[4];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
    if (base + i < total) s += data[base + i];
  uint32_t all = 0;
  (void)wg_exclusive_scan_256(s, lds, &all);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = all;
}

// exclusive scan of data[0..total) in place by ONE workgroup of 1024 threads (the tile sums: total / 4096 entries); the grand
// total goes to *total_out (if given)
__global__ void __launch_bounds__(1024) k_scan_u32(uint32_t* __restrict__ data, uint32_t total, uint32_t* __restrict__ total_out) {
  __shared__ uint32_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t seg = (total + 1023) / 1024;
  const uint64_t lo = (uint64_t)t * seg, hi = lo + seg < total ? lo + seg : total;
  uint32_t s = 0;
  for (uint64_t i = lo; i < hi; i++) s += data[i];
  part[t] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele over the 1024 segment sums
    const uint32_t v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? part[t - 1] : 0;
  for (uint64_t i = lo; i < hi; i++) {
    const uint32_t v = data[i];
    data[i] = run;
    run += v;
  }
  if (total_out && t == 1023) *total_out = part[1023];
}

__global__ void __launch_bounds__(256) k_scan_tiles(uint32_t* __restrict__ data, uint32_t total, const uint32_t* __restrict__ tile_base) {
  __shared__ uint32_t lds[4];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16;
  uint32_t v[16], s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    v[i] = base + i < total ? data[base + i] : 0;
    s += v[i];
  }
  uint32_t run = tile_base[blockIdx.x] + wg_exclusive_scan_256(s, lds, nullptr);
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (base + i < total) data[base + i] = run;
    run += v[i];
  }
}

// data[0..total) -> exclusive prefix sums in place; tile_sum: (total + 4095) / 4096 words of scratch; *total_out = grand total
static void scan_u32(uint32_t* data, uint32_t total, uint32_t* tile_sum, uint32_t* total_out, hipStream_t st) {
  const uint32_t tiles = (total + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(256), 0, st, data, total, tile_sum);
  hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, tile_sum, tiles, total_out);
  hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(256), 0, st, data, total, tile_sum);
}

__global__ void __launch_bounds__(256) k_rs_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n, int shift,
                                                    const uint32_t* __restrict__ hist, uint32_t nw) {
  __shared__ uint32_t run[4][256];  // next free destination of every digit for this wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t w = blockIdx.x * 4 + wv;
  if (w >= nw) return;
  for (int d = lane; d < 256; d += 64) run[wv][d] = hist[(size_t)d * nw + w];
  wave_lds_fence();
  const uint64_t base = (uint64_t)w * KEYS_PER_WAVE;
  const uint64_t below = (1ULL << lane) - 1ULL;
  for (int r = 0; r < ROUNDS; r++) {
    const uint64_t i = base + (uint64_t)r * 64 + lane;
    const bool valid = i < n;
    const uint64_t key = valid ? in[i] : 0;
    const uint32_t d = (uint32_t)(key >> shift) & 255u;
    uint64_t same = __ballot(valid);  // lanes holding a key with my digit
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint64_t bal = __ballot(valid && ((d >> b) & 1u));
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    uint32_t dst = 0;
    if (valid) dst = run[wv][d] + (uint32_t)__popcll(same & below);
    wave_lds_fence();  // every lane has read its digit's counter before the group leaders move it on
    if (valid) {
      out[dst] = key;
      if ((same & below) == 0) run[wv][d] += (uint32_t)__popcll(same);  // the lowest lane of the group
    }
    wave_lds_fence();
  }
}

// number of keys that differ from their predecessor, per wave
__global__ void __launch_bounds__(256) k_uq_count(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ wave_cnt, uint32_t nw) {
  const int lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nw) return;
  const uint64_t base = (uint64_t)w * KEYS_PER_WAVE;
  uint32_t c = 0;
  for (int r = 0; r < ROUNDS; r++) {
    const uint64_t i = base + (uint64_t)r * 64 + lane;
    const bool head = i < n && (i == 0 || keys[i] != keys[i - 1]);
    c += (uint32_t)__popcll(__ballot(head));
  }
  if (lane == 0) wave_cnt[w] = c;
}

__global__ void __launch_bounds__(256) k_uq_scatter(const uint64_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ wave_off,
                                                    uint64_t* __restrict__ out, uint32_t nw) {
  const int lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nw) return;
  const uint64_t base = (uint64_t)w * KEYS_PER_WAVE;
  const uint64_t below = (1ULL << lane) - 1ULL;
  uint32_t run = wave_off[w];
  for (int r = 0; r < ROUNDS; r++) {
    const uint64_t i = base + (uint64_t)r * 64 + lane;
    const uint64_t key = i < n ? keys[i] : 0;
    const bool head = i < n && (i == 0 || key != keys[i - 1]);
    const uint64_t heads = __ballot(head);
    if (head) out[run + (uint32_t)__popcll(heads & below)] = key;
    run += (uint32_t)__popcll(heads);
  }
}

__global__ void k_set_nk_huge(int32_t* nk_search, uint32_t r, const int* d_num, int n_raw, int min_matched) {
  // MinMatched is tested on the raw count (:854), NumKmers is the unique count (:910)
  nk_search[r] = n_raw >= min_matched ? *d_num : 0;
}

// (read index, raw k-mer count, offset of its hashes) of the listed queries, for the host loop
__global__ void k_gather_huge(const uint32_t* __restrict__ list, uint32_t n, const int32_t* __restrict__ nk_raw, const uint64_t* __restrict__ offs,
                              const uint64_t* __restrict__ offs2, uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t r = list[i];
    out[3 * i] = r;
    out[3 * i + 1] = (uint64_t)nk_raw[r];
    out[3 * i + 2] = offs[r] + (offs2 ? offs2[r] : 0);
  }
}

void launch_gather_huge(const uint32_t* list, uint32_t n, const int32_t* nk_raw, const uint64_t* offs, const uint64_t* offs2, uint64_t* out,
                        hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_gather_huge, dim3((n + 255) / 256), dim3(256), 0, st, list, n, nk_raw, offs, offs2, out);
}

// keys[0..n) -> sorted, duplicates removed, back in keys[0..*d_num); tmp: n keys of scratch; d_temp: huge_dedup_temp_bytes(n)
int huge_dedup(uint64_t* keys, uint64_t* tmp, uint32_t n, int* d_num, void* d_temp, size_t temp_bytes, int32_t* nk_search, uint32_t r, int min_matched,
               hipStream_t st) {
  if (n == 0 || temp_bytes < huge_dedup_temp_bytes(n)) return -1;
  const uint32_t nw = n_waves(n);
  const unsigned blocks = (nw + 3) / 4;
  uint32_t* hist = (uint32_t*)d_temp;           // 256 * nw
  uint32_t* wave_cnt = hist + (size_t)256 * nw; // nw
  uint32_t* tile_sum = wave_cnt + nw;           // (256 * nw + 4095) / 4096
  uint64_t *src = keys, *dst = tmp;
  for (int pass = 0; pass < 8; pass++) {
    hipLaunchKernelGGL(k_rs_hist, dim3(blocks), dim3(256), 0, st, src, n, pass * 8, hist, nw);
    scan_u32(hist, 256u * nw, tile_sum, nullptr, st);
    hipLaunchKernelGGL(k_rs_scatter, dim3(blocks), dim3(256), 0, st, src, dst, n, pass * 8, hist, nw);
    std::swap(src, dst);
  }
  // eight passes: the sorted keys are back in `keys`
  hipLaunchKernelGGL(k_uq_count, dim3(blocks), dim3(256), 0, st, keys, n, wave_cnt, nw);
  scan_u32(wave_cnt, nw, tile_sum, (uint32_t*)d_num, st);
  hipLaunchKernelGGL(k_uq_scatter, dim3(blocks), dim3(256), 0, st, keys, n, wave_cnt, tmp, nw);
  if (hipMemcpyAsync(keys, tmp, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_set_nk_huge, dim3(1), dim3(1), 0, st, nk_search, r, d_num, (int)n, min_matched);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace kmcpg
