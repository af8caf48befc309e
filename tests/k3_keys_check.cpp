// k3_keys_check.cpp — host instantiation of kmcp_amd/csrc/k3_keys.hpp (K3's -T test and sort keys) behind one C function: the
// pairs of ONE read filtered and ordered the way k3_sort_* does it (keys, ascending 128-bit order, pairs recovered from the keys).
// Built as a shared object and driven by tests/test_k3_keys_cpu.py against the host half's own order.
#include <algorithm>
#include <vector>

#include "../kmcp_amd/csrc/k3_keys.hpp"

extern "C" uint32_t k3_order(int32_t sort_mode, const uint64_t* col_size, const uint32_t* pairs, uint32_t m, double nh, double min_tcov, uint32_t* out) {
  std::vector<kmcpg::Key> keys;
  for (uint32_t i = 0; i < m; i++) {
    kmcpg_pair p{pairs[2 * i], pairs[2 * i + 1]};
    if (!kmcpg::passes_tcov(p.count, col_size[p.col], min_tcov)) continue;
    keys.push_back(kmcpg::make_key(sort_mode, col_size, p, nh));
  }
  std::sort(keys.begin(), keys.end(), [](const kmcpg::Key& x, const kmcpg::Key& y) { return kmcpg::key_less(x, y); });
  for (size_t i = 0; i < keys.size(); i++) {
    const kmcpg_pair p = kmcpg::pair_of(sort_mode, keys[i]);
    out[2 * i] = p.col;
    out[2 * i + 1] = p.count;
  }
  return (uint32_t)keys.size();
}
