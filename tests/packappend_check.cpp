// packappend_check.cpp — kmcp-search's batch assembly for -g queries (cli/kmcp_search.cpp, round 6): a query packed on its own and moved
// behind a batch with Batch::append_packed (byte copy when the batch ends on a byte, shifted otherwise, runs re-based) must leave the batch
// exactly as packing the same text straight into it with Batch::pack_append does — codes, runs, base count — for every alignment, with N
// gaps, IUPAC bytes and empty queries in between; and kmcpg_unpack2 of the result must spell the concatenated text.  CPU only.
#define main kmcp_search_cli_main
#include "../cli/kmcp_search.cpp"
#undef main

#include <random>

int main() {
  std::mt19937_64 g(11);
  const char alpha[] = "ACGTACGTACGTacgtNNNRYU-";
  for (int it = 0; it < 300; it++) {
    Batch direct, joined;
    direct.packed = joined.packed = true;
    std::string text;
    const int nq = 1 + (int)(g() % 9);
    for (int q = 0; q < nq; q++) {
      const size_t n = it % 7 == 0 ? (size_t)(g() % 5) : (size_t)(g() % 3000);
      std::string s(n, 'A');
      for (auto& c : s) c = (it % 3 == 0) ? "ACGT"[g() & 3] : alpha[g() % (sizeof alpha - 1)];
      if (n > 40 && q % 2) std::fill(s.begin() + 10, s.begin() + 31, 'N');  // the k - 1 gap
      text += s;
      direct.pack_append(s.data(), s.size());
      Batch one;
      one.packed = true;
      // a query arrives in pieces (records), as from the reader
      const size_t cut = n ? (size_t)(g() % n) : 0;
      one.pack_append(s.data(), cut);
      one.pack_append(s.data() + cut, n - cut);
      joined.append_packed(one.codes.data(), one.n_bases, one.exc.data(), one.n_exc);
      if (joined.n_bases != direct.n_bases) { printf("FAIL base count, case %d\n", it); return 1; }
    }
    const size_t nb = (size_t)((direct.n_bases + 3) / 4);
    if (nb && memcmp(direct.codes.data(), joined.codes.data(), nb) != 0) { printf("FAIL codes differ, case %d\n", it); return 1; }
    // runs: the joined batch may hold a run in two pieces where a query boundary cuts it; compare what they spell
    std::vector<uint8_t> a(direct.n_bases + 1), b(joined.n_bases + 1);
    if (kmcpg_unpack2(direct.codes.data(), direct.n_bases, direct.exc.data(), direct.n_exc, a.data()) != 0 ||
        kmcpg_unpack2(joined.codes.data(), joined.n_bases, joined.exc.data(), joined.n_exc, b.data()) != 0) { printf("FAIL unpack: %s\n", kmcpg_last_error()); return 1; }
    if (a != b) { printf("FAIL text differs, case %d\n", it); return 1; }
    for (size_t i = 0; i < text.size(); i++) {
      char c = text[i];
      if (c == 'a' || c == 'c' || c == 'g' || c == 't') c = (char)(c - 32);
      if (c == 'U' || c == 'u') c = 'T';
      if ((char)b[i] != c) { printf("FAIL spelling at %zu, case %d: %c vs %c\n", i, it, (char)b[i], c); return 1; }
    }
  }
  // Batch::append (batches the reader cut before the database was open, joined for a paged index): ids, bases and both CSRs re-based
  for (int pe = 0; pe < 2; pe++) {
    Batch all, acc;
    all.paired = acc.paired = pe != 0;
    bool first = true;
    for (int part = 0; part < 5; part++) {
      Batch b;
      b.paired = pe != 0;
      const int n = part == 2 ? 0 : 1 + (int)(g() % 40);
      for (int i = 0; i < n; i++) {
        std::string id = "q" + std::to_string(g() % 100000), s1((size_t)(g() % 200), 'C'), s2((size_t)(g() % 200), 'G');
        for (Batch* t : {&b, &all}) {
          t->id_buf.insert(t->id_buf.end(), id.begin(), id.end());
          t->id_offs.push_back(t->id_buf.size());
          t->seqs.insert(t->seqs.end(), s1.begin(), s1.end());
          t->offs.push_back(t->seqs.size());
          if (pe) {
            t->seqs2.insert(t->seqs2.end(), s2.begin(), s2.end());
            t->offs2.push_back(t->seqs2.size());
          }
        }
      }
      if (first) {
        acc.id_buf = b.id_buf; acc.id_offs = b.id_offs; acc.seqs = b.seqs; acc.offs = b.offs; acc.seqs2 = b.seqs2; acc.offs2 = b.offs2;
        first = false;
      } else {
        acc.append(b);
      }
    }
    if (acc.id_buf != all.id_buf || acc.id_offs != all.id_offs || acc.seqs != all.seqs || acc.offs != all.offs || acc.seqs2 != all.seqs2 ||
        (pe && acc.offs2 != all.offs2) || acc.n_seq != 5) {
      printf("FAIL Batch::append, paired %d\n", pe);
      return 1;
    }
  }
  printf("ok\n");
  return 0;
}
