// sort_huge.hip — sort + unique (handleQuery, kmcp/cmd/util-db-search.go:874-908) for queries with more than 65 536 k-mers
// (whole genomes under -g): a device-wide radix sort and an adjacent-unique pass from hipCUB/rocPRIM.  The per-read
// workgroup kernels in k1_dedup.hip cover everything smaller; this file exists so that one 5-M-k-mer query does not run on a
// single workgroup.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "kernels.hpp"

namespace kmcpg {

size_t huge_dedup_temp_bytes(uint32_t max_n) {
  size_t a = 0, b = 0;
  uint64_t* p = nullptr;
  int* d = nullptr;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, a, p, p, (int)max_n, 0, 64, (hipStream_t) nullptr);
  (void)hipcub::DeviceSelect::Unique(nullptr, b, p, p, d, (int)max_n, (hipStream_t) nullptr);
  return (a > b ? a : b) + 256;
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

int huge_dedup(uint64_t* keys, uint64_t* tmp, uint32_t n, int* d_num, void* d_temp, size_t temp_bytes, int32_t* nk_search, uint32_t r, int min_matched,
               hipStream_t st) {
  size_t tb = temp_bytes;
  if (hipcub::DeviceRadixSort::SortKeys(d_temp, tb, keys, tmp, (int)n, 0, 64, st) != hipSuccess) return -1;
  tb = temp_bytes;
  if (hipcub::DeviceSelect::Unique(d_temp, tb, tmp, keys, d_num, (int)n, st) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_set_nk_huge, dim3(1), dim3(1), 0, st, nk_search, r, d_num, (int)n, min_matched);
  return 0;
}

}  // namespace kmcpg
