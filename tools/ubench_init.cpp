// ubench_init.cpp — what the HIP runtime costs a short-lived process before its first kernel: the floor under kmcp-search's
// "before the search started" figure (profiles/r06_cli_e2e.txt).  hipcc -O2 -o ubench_init tools/ubench_init.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <chrono>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int* p) { if (p) *p = 1; }
int main() {
  double t0 = now(), t;
  hipInit(0);
  t = now(); printf("hipInit %.1f ms\n", t - t0); t0 = t;
  hipSetDevice(0);
  t = now(); printf("hipSetDevice %.1f ms\n", t - t0); t0 = t;
  void* d = nullptr;
  hipMalloc(&d, 1 << 20);
  t = now(); printf("first hipMalloc (1 MB) %.1f ms\n", t - t0); t0 = t;
  void* big = nullptr;
  hipMalloc(&big, 1500ull << 20);
  t = now(); printf("hipMalloc 1.5 GB %.1f ms\n", t - t0); t0 = t;
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  t = now(); printf("first stream %.1f ms\n", t - t0); t0 = t;
  void* h = nullptr;
  hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
  t = now(); printf("hipHostMalloc 64 MB %.1f ms\n", t - t0); t0 = t;
  void* h2 = nullptr;
  hipHostMalloc(&h2, 64 << 20, hipHostMallocDefault);
  t = now(); printf("hipHostMalloc 64 MB again %.1f ms\n", t - t0); t0 = t;
  hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int*)d);
  hipStreamSynchronize(st);
  t = now(); printf("first kernel (code object load) %.1f ms\n", t - t0); t0 = t;
  hipMemcpyAsync(big, h, 64 << 20, hipMemcpyHostToDevice, st);
  hipStreamSynchronize(st);
  t = now(); printf("first H2D 64 MB %.1f ms\n", t - t0); t0 = t;
  for (int i = 0; i < 4; i++) hipMemcpyAsync((char*)big + ((size_t)i << 26), h, 64 << 20, hipMemcpyHostToDevice, st);
  hipStreamSynchronize(st);
  t = now(); printf("4 x H2D 64 MB %.1f ms (%.1f GB/s)\n", t - t0, 4 * 64.0 / 1024 / ((t - t0) * 1e-3)); t0 = t;
  hipFree(big);
  t = now(); printf("hipFree 1.5 GB %.1f ms\n", t - t0); t0 = t;
  hipHostFree(h);
  t = now(); printf("hipHostFree 64 MB %.1f ms\n", t - t0); t0 = t;
  return 0;
}
