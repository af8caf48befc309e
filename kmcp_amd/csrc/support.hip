// support.hip — not on the query path: load-time layout (k_repack), parity helpers (k_gather_rows, k_plant*), the synthetic
// index of bench.py (k_synth_fill) and the `kmcp index` scatter (k_build_scatter, index.go:1107-1309).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.hpp"
#include "device_utils.hpp"
#include "kernels.hpp"

namespace kmcpg {

// ------------------------------------------------------------------------------------------------
// layout: on-disk rows (NumRowBytes, unpadded — serialization.go:140,379) -> HBM rows (stride)
// ------------------------------------------------------------------------------------------------
// dst = the group's rows (zero-filled beforehand); this block's bytes land at [byte_off, byte_off + row_bytes) of every row.
// Interior dwords are stored, the (at most two) dwords a block shares with its neighbours in the group are OR-ed in.
// The padding bits of a row's last byte (columns ncols .. 8*row_bytes-1; zero in every file `kmcp index` writes, index.go:1157)
// are cleared here, so the query kernel may rely on "count > 0 => real column" whatever a file holds.
__global__ void k_repack(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n_rows, uint32_t row_bytes,
                         uint32_t stride, uint32_t byte_off, uint32_t last_mask) {
  const uint32_t w0 = byte_off / 4, w1 = (byte_off + row_bytes + 3) / 4;  // dst dwords touched per row
  const uint32_t wpr = w1 - w0;
  const uint64_t total = n_rows * wpr;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t row = i / wpr;
    const uint32_t w = w0 + (uint32_t)(i % wpr);
    const uint8_t* s = src + row * row_bytes;
    uint32_t v = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const uint32_t b = w * 4 + t;  // byte of the dst row
      if (b >= byte_off && b < byte_off + row_bytes) {
        uint32_t x = s[b - byte_off];
        if (b - byte_off == row_bytes - 1) x &= last_mask;
        v |= x << (8 * t);
      }
    }
    uint32_t* d = reinterpret_cast<uint32_t*>(dst + row * stride) + w;
    if (w * 4 >= byte_off && w * 4 + 4 <= byte_off + row_bytes) *d = v;
    else if (v) atomicOr(d, v);
  }
}

void launch_repack(const uint8_t* src, uint8_t* dst, uint64_t n_rows, uint32_t row_bytes, uint32_t stride, uint32_t byte_off, uint32_t ncols, hipStream_t st) {
  if (n_rows == 0) return;
  const uint32_t pad = row_bytes * 8u - ncols;                      // 0..7 unused low bits in the last byte (bit 7 = first column)
  const uint32_t last_mask = pad < 8 ? (0xffu << pad) & 0xffu : 0;  // a header whose counts disagree is refused at open
  uint64_t total = n_rows * ((byte_off + row_bytes + 3) / 4 - byte_off / 4);
  unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(k_repack, dim3(blocks), dim3(256), 0, st, src, dst, n_rows, row_bytes, stride, byte_off, last_mask);
}

// rows idx[0..n) — or first .. first+n-1 when idx is null — of a block, packed at the on-disk width
__global__ void k_gather_rows(const uint8_t* __restrict__ rows, uint32_t stride, uint32_t row_bytes, const uint64_t* __restrict__ idx, uint64_t first,
                              uint64_t n, uint8_t* __restrict__ out) {
  const uint64_t total = n * row_bytes;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / row_bytes;
    const uint32_t b = (uint32_t)(i % row_bytes);
    out[i] = rows[(idx ? idx[r] : first + r) * stride + b];
  }
}

void launch_gather_rows(const uint8_t* rows, uint32_t stride, uint32_t row_bytes, const uint64_t* idx, uint64_t first, uint64_t n, uint8_t* out,
                        hipStream_t st) {
  if (n == 0) return;
  uint64_t total = n * row_bytes;
  unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, st, rows, stride, row_bytes, idx, first, n, out);
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

// `rows` = the block's first byte inside its group's rows (row pitch `stride`); the bits depend on (key, row, qword of the block's
// own padded row) only, so a block holds the same bits however it is grouped with others.  own_stride = the block's row
// padded as if it stood alone.
__global__ void k_synth_fill(uint8_t* __restrict__ rows, uint64_t n_rows, uint32_t stride, uint32_t own_stride, uint32_t ncols, uint64_t key, uint32_t p8) {
  const uint32_t qpr = own_stride / 8;  // qwords per row of the block
  const uint32_t row_bytes = (ncols + 7) / 8;
  const uint32_t qused = (row_bytes + 7) / 8;
  const uint64_t total = n_rows * qused;
  const bool aligned = ((uintptr_t)rows & 7) == 0 && (stride & 7) == 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t row = i / qused;
    const uint32_t qw = (uint32_t)(i % qused);
    uint64_t v = bernoulli64(key, row * qpr + qw, p8);
    // zero the bits of columns >= ncols (padding columns never match: :7466 scans them but count is 0)
    const uint32_t col0 = qw * 64;
    if (col0 + 64 > ncols) {
      uint64_t m = 0;
      for (uint32_t c = col0; c < ncols; c++) {
        const uint32_t byte = (c - col0) >> 3, bit = 7 - ((c - col0) & 7);
        m |= 1ULL << (byte * 8 + bit);
      }
      v &= m;
    }
    uint8_t* d = rows + row * stride + (uint64_t)qw * 8;
    if (aligned && qw * 8 + 8 <= row_bytes) *reinterpret_cast<uint64_t*>(d) = v;
    else
      for (uint32_t t = 0; t < 8 && qw * 8 + t < row_bytes; t++) d[t] = (uint8_t)(v >> (8 * t));
  }
}

void launch_synth_fill(uint8_t* rows, uint64_t n_rows, uint32_t stride, uint32_t own_stride, uint32_t ncols, uint64_t key, uint32_t p8, hipStream_t st) {
  uint64_t total = n_rows * (((ncols + 7) / 8 + 7) / 8);
  unsigned blocks = (unsigned)((total + 255) / 256 > 262144 ? 262144 : (total + 255) / 256);
  hipLaunchKernelGGL(k_synth_fill, dim3(blocks), dim3(256), 0, st, rows, n_rows, stride, own_stride, ncols, key, p8);
}

// sigs[h % NumSigs] |= 1 << (7 - col%8)  (index.go:1157) for a list of hashes
__global__ void k_plant(BlockDev bd, uint32_t col, int num_hashes, const uint64_t* __restrict__ hashes, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h = hashes[i];
    const uint32_t ha = (uint32_t)(h >> 32), hb = (uint32_t)h;
    for (int t = 0; t < num_hashes; t++) {
      const uint64_t hv = num_hashes == 1 ? h : (uint64_t)(uint32_t)(ha + hb * (uint32_t)t);
      const uint64_t row = fastmod_u64(hv, bd.num_sigs, bd.magic_hi);
      // bd.rows is the block's first byte inside its group's row: any alignment
      const uintptr_t byte = (uintptr_t)bd.rows + row * bd.stride + (col >> 3);
      atomicOr(reinterpret_cast<uint32_t*>(byte & ~(uintptr_t)3), (uint32_t)(1u << (7 - (col & 7))) << (8 * (byte & 3)));
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
        const uint64_t row = fastmod_u64(hv, bd.num_sigs, bd.magic_hi);
        const uintptr_t byte = (uintptr_t)bd.rows + row * bd.stride + (col >> 3);
        atomicOr(reinterpret_cast<uint32_t*>(byte & ~(uintptr_t)3), (uint32_t)(1u << (7 - (col & 7))) << (8 * (byte & 3)));
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
__global__ void k_build_scatter(uint8_t* __restrict__ sigs, uint64_t num_sigs, uint64_t mh, uint32_t row_bytes, int num_hashes,
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
      const uint64_t byte = fastmod_u64(hv, num_sigs, mh) * row_bytes + (col >> 3);
      uint32_t* w = reinterpret_cast<uint32_t*>(sigs + (byte & ~3ULL));
      atomicOr(w, (uint32_t)(1u << (7 - (col & 7))) << (8 * (byte & 3)));
    }
  }
}

void launch_build_scatter(uint8_t* sigs, uint64_t num_sigs, uint64_t mh, uint32_t row_bytes, int num_hashes, const uint64_t* hashes,
                          const uint64_t* col_off, uint32_t col0, uint32_t n_cols, uint64_t n, hipStream_t st) {
  if (n == 0) return;
  unsigned blocks = (unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
  hipLaunchKernelGGL(k_build_scatter, dim3(blocks), dim3(256), 0, st, sigs, num_sigs, mh, row_bytes, num_hashes, hashes, col_off, col0, n_cols, n);
}

}  // namespace kmcpg

namespace kmcpg {

// ------------------------------------------------------------------------------------------------
// 2-bit packed upload (host.cpp stage): a batch of long queries is 4x smaller over PCIe as 2-bit codes; K1 reads ASCII as before.
// One thread expands 4 packed bytes to 16 bases (one 16-byte store); the few bytes that are not A/C/G/T/U come as runs and are
// written over the result (the hash of such a byte depends on its exact value: seed 0, complement by its low 3 bits — nthash.hpp).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unpack2(const uint8_t* __restrict__ packed, uint8_t* __restrict__ out, uint64_t n_bases) {
  const uint64_t n16 = n_bases / 16;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr uint32_t LUT = ('A') | ('C' << 8) | ('T' << 16) | ('G' << 24);  // code -> base
  if (i < n16) {
    const uint32_t w = reinterpret_cast<const uint32_t*>(packed)[i];
    uint32_t o[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t byte = (w >> (8 * b)) & 0xffu;
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) v |= ((LUT >> (8 * ((byte >> (2 * j)) & 3u))) & 0xffu) << (8 * j);
      o[b] = v;
    }
    reinterpret_cast<uint4*>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  } else if (i == n16) {  // the last < 16 bases
    for (uint64_t j = n16 * 16; j < n_bases; j++) out[j] = (uint8_t)((LUT >> (8 * ((packed[j >> 2] >> (2 * (j & 3))) & 3u))) & 0xffu);
  }
}

__global__ void __launch_bounds__(256) k_apply_exc(const ExcRun* __restrict__ runs, uint32_t n_runs, uint8_t* __restrict__ out) {
  for (uint32_t r = blockIdx.x; r < n_runs; r += gridDim.x) {
    const ExcRun e = runs[r];
    for (uint32_t j = threadIdx.x; j < e.len; j += blockDim.x) out[e.pos + j] = (uint8_t)e.byte;
  }
}

void launch_apply_exc(const ExcRun* runs, uint32_t n_runs, uint8_t* out, hipStream_t st) {
  if (n_runs) hipLaunchKernelGGL(k_apply_exc, dim3(std::min<unsigned>(n_runs, 65536u)), dim3(256), 0, st, runs, n_runs, out);
}

void launch_unpack2(const uint8_t* packed, uint8_t* out, uint64_t n_bases, const ExcRun* runs, uint32_t n_runs, hipStream_t st) {
  if (n_bases == 0) return;
  const uint64_t threads = n_bases / 16 + 1;
  hipLaunchKernelGGL(k_unpack2, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, packed, out, n_bases);
  if (n_runs) hipLaunchKernelGGL(k_apply_exc, dim3(std::min<unsigned>(n_runs, 65536u)), dim3(256), 0, st, runs, n_runs, out);
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT ONLY (KMCPG_DEBUG_ROWSORT, profiles/r05_rowsort_gate.txt): the hashes of every query re-ordered by the row they
// address in ONE block (h % num_sigs), so that all units in flight sweep that block's rows in the same direction.  mode 2
// rotates each query's sorted list by a pseudo-random offset: the same per-unit locality without the phase coherence.
// One workgroup per query of at most 4096 k-mers (longer ones are left as they are); bitonic sort of (row, index) keys in LDS.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_debug_rowsort(uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offs, const int32_t* __restrict__ nk,
                                                       uint64_t num_sigs, uint64_t mh, int mode) {
  __shared__ uint64_t key[4096];
  __shared__ uint64_t val[4096];
  const uint32_t r = blockIdx.x;
  const int n = nk[r];
  if (n <= 1 || n > 4096) return;
  uint64_t* h = hashes + offs[r];
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = threadIdx.x; i < m; i += 256) {
    if (i < n) {
      val[i] = h[i];
      key[i] = (fastmod_u64(val[i], num_sigs, mh) << 12) | (uint64_t)i;
    } else {
      key[i] = ~0ULL;
    }
  }
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < m / 2; i += 256) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t a = key[lo], b = key[hi];
        if ((a > b) == up) {
          key[lo] = b;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  const int rot = mode == 2 ? (int)(splitmix64(r) % (uint64_t)n) : 0;
  for (int i = threadIdx.x; i < n; i += 256) h[(i + rot) % n] = val[key[i] & 4095u];
}

void launch_debug_rowsort(uint64_t* hashes, const uint64_t* offs, const int32_t* nk, uint32_t n_reads, uint64_t num_sigs, uint64_t mh, int mode, hipStream_t st) {
  if (n_reads == 0) return;
  hipLaunchKernelGGL(k_debug_rowsort, dim3(n_reads), dim3(256), 0, st, hashes, offs, nk, num_sigs, mh, mode);
}

}  // namespace kmcpg
