// ubench_cache.cpp — does the cache hierarchy (4 MiB L2 per XCD, 256 MiB Infinity Cache) serve random row gathers faster
// than HBM does?  Random 16/64/128/256-byte gathers over working sets of 1 MiB ... 4 GiB, temporal and non-temporal loads,
// device-wide working set ("shared": what a slot-major K2 launch would see in the Infinity Cache) and one working set per
// XCD ("xcd": blockIdx % 8 picks the region; what an XCD-partitioned launch would see in L2).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_cache.cpp -o ubench_cache
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int LPR, int ROWS, bool NT>
__global__ void __launch_bounds__(256) gather(const uint8_t* __restrict__ buf, uint64_t nrows, uint32_t stride, uint64_t region_bytes,
                                              int per_xcd, int iters, uint32_t* out, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const uint64_t gid = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / LPR) + g;
  const uint8_t* base = buf + (per_xcd ? (uint64_t)(blockIdx.x & 7) * region_bytes : 0) + li * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    u32x4 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const uint64_t row = mix(seed + gid * 1000003ULL + (uint64_t)it * ROWS + r) % nrows;
      const u32x4* q = reinterpret_cast<const u32x4*>(base + row * stride);
      v[r] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) acc ^= v[r];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
template <int LPR, bool NT>
static void run(const uint8_t* buf, uint64_t region_bytes, int per_xcd, uint32_t* out) {
  const uint32_t stride = LPR * 16;
  const uint64_t nrows = region_bytes / stride;
  const uint64_t groups = (1ull << 24) * (LPR >= 8 ? 1 : 2);
  const unsigned blocks = (unsigned)(groups / (64 / LPR) / 4);
  const int iters = 16;
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  // warm: two passes so the region is resident wherever it can be
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((gather<LPR, 8, NT>), dim3(blocks), dim3(256), 0, 0, buf, nrows, stride, region_bytes, per_xcd, iters, out, 1ULL + w);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<LPR, 8, NT>), dim3(blocks), dim3(256), 0, 0, buf, nrows, stride, region_bytes, per_xcd, iters, out, 7ULL);
  CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  const double rows = (double)blocks * 4 * (64 / LPR) * iters * 8;
  printf("%-6s %s  req=%4d B  set=%8.1f MiB%s: %8.2f ms  %6.1f Greq/s  %7.0f GB/s\n", per_xcd ? "xcd" : "shared", NT ? "nt " : "tmp", stride,
         region_bytes / 1048576.0, per_xcd ? " per XCD" : "", ms, rows / ms / 1e6, rows * stride / ms / 1e6);
  fflush(stdout);
}
int main() {
  const uint64_t cap = 8ull << 30;
  uint8_t* buf; uint32_t* out;
  CHK(hipMalloc(&buf, cap)); CHK(hipMalloc(&out, 64)); CHK(hipMemset(buf, 1, cap));
  const uint64_t MiB = 1ull << 20;
  const uint64_t shared_sets[] = {2 * MiB, 8 * MiB, 24 * MiB, 48 * MiB, 64 * MiB, 96 * MiB, 128 * MiB, 160 * MiB, 192 * MiB, 256 * MiB, 384 * MiB, 1024 * MiB, 4096 * MiB, 8192 * MiB};
  for (uint64_t s : shared_sets) {
    run<1, false>(buf, s, 0, out); run<4, false>(buf, s, 0, out); run<8, false>(buf, s, 0, out); run<16, false>(buf, s, 0, out);
    run<4, true>(buf, s, 0, out); run<8, true>(buf, s, 0, out);
  }
  const uint64_t xcd_sets[] = {MiB / 2, 1 * MiB, 2 * MiB, 3 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB};
  for (uint64_t s : xcd_sets) {
    run<1, false>(buf, s, 1, out); run<4, false>(buf, s, 1, out); run<8, false>(buf, s, 1, out); run<16, false>(buf, s, 1, out);
  }
  return 0;
}
