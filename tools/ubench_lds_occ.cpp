// ubench_lds_occ.cpp — how many workgroups of a given dynamic LDS size does one CU really hold?  (profiles/r06_k1_roll.txt)
// A kernel of one wave per workgroup spins for a fixed number of clocks; grid = CUs x 64 workgroups; time vs LDS size gives residency.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long cycles, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (lds[threadIdx.x] == -1) *sink = 1;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlockOptin %zu\n", p.multiProcessorCount, p.sharedMemPerBlock,
         p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
  int* sink;
  hipMalloc(&sink, 4);
  const int wgs = p.multiProcessorCount * 64;
  for (int kb : {1, 4, 8, 14, 16, 20, 32, 40, 55, 64}) {
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 64, (size_t)kb * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), (size_t)kb * 1024, 0, 1000LL, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), (size_t)kb * 1024, 0, 200000LL, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // each workgroup spins 200 000 clocks (100 MHz counter? s_memtime runs at a fixed rate): rounds = 64 / resident per CU
    printf("LDS %2d KB per workgroup of 64 threads: occupancy API %2d per CU, %d workgroups in %.3f ms\n", kb, occ, wgs, ms);
  }
  return 0;
}
