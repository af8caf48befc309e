// k3_keys.hpp — what K3 decides per match, as plain functions for device and host: the -T test (util-db-search.go:7471-7473) and the
// 128-bit sort keys whose ascending order is the reference's order of a query's matches (Matches.Less / SortByTCov / SortByJacc,
// util-db-search.go:105-145; ties broken by column, as finalize.cpp does).  The layout of the keys is described at the top of
// k3_finalize.hip.  Host instantiation: tests/k3_keys_check.cpp against the host half's own order (tests/test_k3_keys_cpu.py).
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/kmcp_gpu.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KMCPG_K3_HD __host__ __device__ __forceinline__
#else
#define KMCPG_K3_HD inline
#endif

namespace kmcpg {

struct Key {
  uint64_t a, b;
};
KMCPG_K3_HD bool key_less(const Key& x, const Key& y) { return x.a < y.a || (x.a == y.a && x.b < y.b); }

KMCPG_K3_HD uint64_t bits_of_double(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__double_as_longlong(v);
#else
  uint64_t u;
  memcpy(&u, &v, sizeof u);
  return u;
#endif
}

// the -T test exactly as the reference makes it: float64(count) / float64(size) >= minTCov
KMCPG_K3_HD bool passes_tcov(uint32_t count, uint64_t size, double min_tcov) {
  if (min_tcov <= 0.0) return true;
  return (double)count / (double)size >= min_tcov;
}

// sort_mode: 0 qcov, 1 tcov, 2 jacc (-s), 3 column order (-S); nh = float64(NumKmers of the read)
KMCPG_K3_HD Key make_key(int32_t sort_mode, const uint64_t* __restrict__ col_size, kmcpg_pair p, double nh) {
  const uint32_t inv = ~p.count;
  Key k;
  if (sort_mode == 0) {
    const uint64_t s = col_size[p.col];
    k.a = ((uint64_t)inv << 32) | (s >> 32);
    k.b = (s << 32) | p.col;
  } else {
    const double c = (double)p.count;
    if (sort_mode == 3) k.a = p.col;
    else {
      const double nt = (double)col_size[p.col];
      const double score = sort_mode == 1 ? c / nt : c / (nh + nt - c);  // :7487-7489, left to right as Go evaluates it
      k.a = ~bits_of_double(score);
    }
    k.b = ((uint64_t)inv << 32) | p.col;
  }
  return k;
}

KMCPG_K3_HD kmcpg_pair pair_of(int32_t sort_mode, const Key& k) {
  kmcpg_pair p;
  p.col = (uint32_t)k.b;
  p.count = ~(uint32_t)((sort_mode == 0 ? k.a : k.b) >> 32);
  return p;
}

}  // namespace kmcpg
