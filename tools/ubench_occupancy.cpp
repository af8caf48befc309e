// ubench_occupancy.cpp — how many random 128-byte requests must be in flight to reach the device's request rate?
// Random 128-B (8 lanes x 16 B) non-temporal gathers over a 2 GiB / 55 GiB-like set with the number of resident waves per SIMD
// limited through the workgroup's LDS footprint (160 KB per CU) and 4 / 8 / 16 rows in flight per wave.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_occupancy.cpp -o ubench_occupancy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ uint32_t dyn_lds[];
template <int LPR, int ROWS>
__global__ void __launch_bounds__(256) gather(const uint8_t* __restrict__ buf, uint64_t nrows, int iters, uint32_t* out, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const uint64_t gid = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / LPR) + g;
  const uint8_t* base = buf + li * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    u32x4 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const uint64_t row = mix(seed + gid * 1000003ULL + (uint64_t)it * ROWS + r) % nrows;
      v[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + row * (LPR * 16)));
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) acc ^= v[r];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) { out[0] = 1; dyn_lds[0] = 1; }
}
template <int LPR, int ROWS>
static void run(const uint8_t* buf, uint64_t set_bytes, int waves_per_simd, uint32_t* out) {
  const uint64_t nrows = set_bytes / (LPR * 16);
  const unsigned blocks = (1u << 24) / (64 / LPR) / 4;
  const int iters = 128 / ROWS;
  const size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / waves_per_simd) & ~(size_t)1023;  // W workgroups of 4 waves per CU
  auto k = gather<LPR, ROWS>;
  CHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, buf, nrows, iters, out, 1ULL);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, buf, nrows, iters, out, 7ULL);
  CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  const double rows = (double)blocks * 4 * (64 / LPR) * iters * ROWS;
  printf("req=%4d B  set=%7.0f MiB  waves/SIMD=%d  rows in flight/wave=%2d: %8.2f ms  %6.1f Greq/s  %7.0f GB/s\n", LPR * 16, set_bytes / 1048576.0,
         waves_per_simd, ROWS, ms, rows / ms / 1e6, rows * LPR * 16 / ms / 1e6);
  fflush(stdout);
}
int main() {
  const uint64_t cap = 48ull << 30;
  uint8_t* buf; uint32_t* out;
  CHK(hipMalloc(&buf, cap)); CHK(hipMalloc(&out, 64)); CHK(hipMemset(buf, 1, cap));
  for (uint64_t set : {2ull << 30, 48ull << 30})
    for (int w : {8, 5, 4, 3, 2, 1}) {
      run<8, 4>(buf, set, w, out); run<8, 8>(buf, set, w, out); run<8, 16>(buf, set, w, out);
      if (w == 5 || w == 2) { run<64, 4>(buf, set, w, out); run<64, 8>(buf, set, w, out); }
    }
  return 0;
}
