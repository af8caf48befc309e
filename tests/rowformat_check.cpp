// rowformat_check.cpp — kmcp-search's row formatter against itself and against printf (CPU only; built by tests/test_rowformat_cpu.py
// with ASan/UBSan): RowFormatter::rows (a query's constant fields and a column's text formatted once) must give the bytes of
// RowFormatter::row called per match, for short and oversized IDs / target names, every count of matches around the switch
// between the two paths, FPR strings from the table and from the map; w_f4 must print what "%.4f" prints, ties and large
// values included.
#define main kmcp_search_cli_main
#include "../cli/kmcp_search.cpp"
#undef main

#include <random>

int main() {
  std::mt19937_64 g(7);
  auto rnd = [&](uint64_t n) { return (uint64_t)(g() % n); };
  // targets: ordinary, empty, and one longer than the scratch line
  std::vector<std::string> target;
  for (int c = 0; c < 300; c++) {
    std::string t = "GCF_" + std::to_string(100000000 + c * 7919) + ".1";
    if (c % 97 == 0) t.clear();
    if (c % 131 == 5) t.assign(9000, 'x');
    target.push_back(t);
  }
  std::vector<uint32_t> tidx(300);
  std::vector<uint64_t> gsize(300);
  for (int c = 0; c < 300; c++) {
    tidx[c] = (uint32_t)(rnd(10) | (10u << 16));
    gsize[c] = rnd(3) ? rnd(20000000) : (rnd(2) ? 0 : ~0ull >> rnd(30));
  }
  unsigned long long rows = 0;
  RowFormatter A, B;
  for (int it = 0; it < 6000; it++) {
    const uint64_t cnt = it < 40 ? (uint64_t)it % 10 : 1 + rnd(it % 50 == 0 ? 400 : 12);
    std::string id = "read" + std::to_string(g());
    if (it % 211 == 0) id.assign(700 + rnd(900), 'q');  // around the 1024-byte limit of the pre-formatted prefix
    if (it % 977 == 0) id.assign(9000, 'I');             // longer than the scratch line
    const int qkmers = it % 5 == 0 ? 4097 + (int)rnd(100000) : 1 + (int)rnd(500);
    const int qlen = qkmers + 20, k = 21 + (int)rnd(11);
    std::vector<kmcpg_match> ms(cnt);
    for (auto& m : ms) {
      memset(&m, 0, sizeof m);
      m.col = (uint32_t)rnd(300);
      m.target_idx = tidx[m.col];
      m.gsize = gsize[m.col];
      m.mkmers = 1 + (int)rnd((uint64_t)qkmers);
      m.fpr = it % 3 ? ldexp((double)rnd(1 << 20), -(int)rnd(80)) : 0.0;
      m.qcov = (double)m.mkmers / qkmers;
      m.tcov = rnd(50) ? (double)m.mkmers / (double)(1 + rnd(5000000)) : (double)rnd(1000000) * 1000.5;  // also >= 1e5: the printf path
      m.jacc = rnd(7) ? m.qcov * 0.5 : (double)rnd(100000) / 10000.0 + 0.00005;                           // exact-looking ties
    }
    std::string a, b;
    A.rows(a, id, qlen, qkmers, ms.data(), cnt, target, k, (uint64_t)it * 131);
    for (uint64_t j = 0; j < cnt; j++) B.row(b, id, qlen, qkmers, cnt, target[ms[j].col], ms[j], k, (uint64_t)it * 131);
    if (a != b) {
      printf("rows() and row() differ for query %d (%llu matches)\n", it, (unsigned long long)cnt);
      return 1;
    }
    rows += cnt;
  }
  // "%.4f"
  unsigned long long nf = 0;
  for (int it = 0; it < 3000000; it++) {
    double v;
    switch (it % 6) {
      case 0: v = (double)rnd(1000000) / 10000.0 + 0.00005; break;      // decimal ties as doubles see them
      case 1: v = ldexp((double)rnd(1ull << 53), -(int)rnd(70)); break;  // all magnitudes
      case 2: v = (double)rnd(1000) / 1000.0; break;
      case 3: v = 99999.0 + (double)rnd(30000) / 10000.0; break;         // around the fast path's bound
      case 4: v = (double)rnd(1ull << 40) + 0.5 * (double)rnd(2); break;
      default: v = (double)(1 + rnd(255)) / (double)(1 + rnd(255)); break;
    }
    char x[512], y[512];
    const size_t nx = (size_t)(RowFormatter::w_f4(x, v) - x);
    const size_t ny = (size_t)snprintf(y, sizeof y, "%.4f", v);
    if (nx != ny || memcmp(x, y, nx) != 0) {
      printf("w_f4(%a) = %.*s, printf says %s\n", v, (int)nx, x, y);
      return 1;
    }
    nf++;
  }
  printf("%llu rows identical, %llu values printed like %%.4f\n", rows, nf);
  return 0;
}
