// gunzip_check.cpp — cli/fast_gunzip.hpp against zlib (built by tests/test_gunzip_cpu.py with ASan/UBSan): streams of every
// compression level and strategy over FASTQ-like, repetitive, random and tiny inputs, stored and fixed-Huffman blocks, several
// members in a row, header fields, trailing bytes; then damaged and truncated copies: whatever zlib's inflate accepts must come
// out identical, and what it rejects must be rejected (never a crash, never a short read).
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <string>
#include <vector>

#include "../cli/fast_gunzip.hpp"

static std::mt19937_64 g(99);
static uint64_t rnd(uint64_t n) { return g() % n; }

static std::string make_text(int kind, size_t n) {
  std::string t;
  t.reserve(n + 400);
  if (kind == 0) {  // FASTQ
    uint64_t i = 0;
    while (t.size() < n) {
      t += "@read" + std::to_string(i++) + "/1\n";
      const int len = 100 + (int)rnd(60);
      for (int j = 0; j < len; j++) t.push_back("ACGT"[rnd(4)]);
      t += "\n+\n";
      for (int j = 0; j < len; j++) t.push_back((char)(33 + (rnd(10) ? 37 : rnd(41))));
      t.push_back('\n');
    }
  } else if (kind == 1) {  // long runs and short distances
    while (t.size() < n) t.append((size_t)(1 + rnd(600)), (char)('a' + rnd(3)));
  } else if (kind == 2) {  // incompressible
    while (t.size() < n) t.push_back((char)rnd(256));
  } else {  // repeats at all distances up to the window
    std::string unit;
    for (int j = 0; j < 1 + (int)rnd(40000); j++) unit.push_back((char)('A' + rnd(20)));
    while (t.size() < n) t += unit;
  }
  t.resize(n);
  return t;
}

static std::string gz_member(const std::string& text, int level, int strategy, int flags) {
  z_stream z{};
  if (deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 1 + (int)rnd(9), strategy) != Z_OK) abort();
  gz_header h{};
  std::string name = "file name.fq", comment = "a comment";
  unsigned char extra[5] = {1, 2, 3, 4, 5};
  if (flags) {
    if (flags & 1) h.name = (Bytef*)name.c_str();
    if (flags & 2) h.comment = (Bytef*)comment.c_str();
    if (flags & 4) { h.extra = extra; h.extra_len = 5; }
    if (flags & 8) h.hcrc = 1;
    deflateSetHeader(&z, &h);
  }
  std::string out(deflateBound(&z, (uLong)text.size()) + 256, '\0');
  z.next_in = (Bytef*)text.data();
  z.avail_in = (uInt)text.size();
  z.next_out = (Bytef*)&out[0];
  z.avail_out = (uInt)out.size();
  // now and then a flush in the middle: empty stored blocks, block boundaries at odd places
  if (rnd(3) == 0 && text.size() > 10) {
    z.avail_in = (uInt)rnd(text.size());
    deflate(&z, rnd(2) ? Z_FULL_FLUSH : Z_SYNC_FLUSH);
    z.avail_in = (uInt)(text.size() - (size_t)(z.next_in - (Bytef*)text.data()));
  }
  if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
  out.resize(z.total_out);
  deflateEnd(&z);
  return out;
}

// zlib's verdict on a gzip stream of concatenated members: the text, or an error
static bool zlib_gunzip(const std::string& in, std::string* out) {
  out->clear();
  size_t pos = 0;
  bool any = false;
  while (pos < in.size()) {
    if (any && !(in.size() - pos >= 2 && (unsigned char)in[pos] == 0x1f && (unsigned char)in[pos + 1] == 0x8b)) break;  // trailing bytes
    z_stream z{};
    if (inflateInit2(&z, 15 + 16) != Z_OK) abort();
    z.next_in = (Bytef*)in.data() + pos;
    z.avail_in = (uInt)(in.size() - pos);
    std::vector<char> buf(1 << 16);
    int rc;
    do {
      z.next_out = (Bytef*)buf.data();
      z.avail_out = (uInt)buf.size();
      rc = inflate(&z, Z_NO_FLUSH);
      if (rc != Z_OK && rc != Z_STREAM_END) {
        inflateEnd(&z);
        return false;
      }
      out->append(buf.data(), buf.size() - z.avail_out);
      if (rc == Z_OK && z.avail_in == 0 && z.avail_out != 0) {  // input ended inside the member
        inflateEnd(&z);
        return false;
      }
    } while (rc != Z_STREAM_END);
    pos = in.size() - z.avail_in;
    inflateEnd(&z);
    any = true;
  }
  return any;
}

static bool fast_gunzip(const std::string& in, std::string* out, std::string* err) {
  // the input exactly as long as it is (ASan sees any read past its end)
  std::vector<uint8_t> copy(in.begin(), in.end());
  FastGunzip f(copy.data(), copy.size());
  out->clear();
  std::vector<char> buf(1 + rnd(3) * 70000 + rnd(5000));
  for (;;) {
    const ssize_t n = f.read(buf.data(), buf.size());
    if (n < 0) {
      *err = f.error();
      return false;
    }
    if (n == 0) return true;
    out->append(buf.data(), (size_t)n);
  }
}

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 350;
  if (argc > 2) g.seed((uint64_t)atoll(argv[2]));
  unsigned long long good = 0, rejected = 0;
  for (int it = 0; it < cases; it++) {
    std::string stream, text_all;
    const int members = 1 + (rnd(4) == 0 ? (int)rnd(4) : 0);
    for (int m = 0; m < members; m++) {
      const size_t n = rnd(8) == 0 ? rnd(40) : (rnd(10) == 0 ? 1500000 + rnd(2500000) : rnd(300000));
      const std::string text = make_text((int)rnd(4), n);
      static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
      stream += gz_member(text, (int)rnd(10), strategies[rnd(6)], rnd(3) ? 0 : (int)rnd(16));
      text_all += text;
    }
    if (rnd(6) == 0) stream += std::string(1 + rnd(20), (char)rnd(200));  // trailing bytes (never "\x1f\x8b": rnd(200) < 0x1f8b's second byte)
    std::string a, b, err;
    const bool za = zlib_gunzip(stream, &a), fa = fast_gunzip(stream, &b, &err);
    if (!za || a != text_all) { printf("generator: zlib does not return the text (case %d)\n", it); return 1; }
    if (!fa || b != a) { printf("case %d: %s, %zu bytes instead of %zu\n", it, fa ? "different text" : err.c_str(), b.size(), a.size()); return 1; }
    good++;
    // damage
    for (int d = 0; d < 6; d++) {
      std::string bad = stream;
      const int how = (int)rnd(4);
      size_t where = 0;
      if (how == 0) bad.resize(rnd(bad.size()));  // truncated
      else if (how == 1) { where = rnd(bad.size()); bad[where] ^= (char)(1u << rnd(8)); }
      else if (how == 2) { const size_t p = rnd(bad.size()); bad.erase(p, 1 + rnd(4)); }
      else { const size_t p = rnd(bad.size()); bad.insert(p, std::string(1 + rnd(3), (char)rnd(256))); }
      const bool z2 = zlib_gunzip(bad, &a), f2 = fast_gunzip(bad, &b, &err);
      if (z2 != f2 || (z2 && a != b)) {
        printf("at byte %zu of %zu (members %d): ", where, bad.size(), members);
        printf("case %d damage %d (%d): zlib %s, fast %s%s%s\n", it, d, how, z2 ? "accepts" : "rejects", f2 ? "accepts" : "rejects: ", f2 ? "" : err.c_str(),
               z2 && f2 ? " (different text)" : "");
        return 1;
      }
      if (!z2) rejected++;
    }
  }
  printf("%llu streams identical to zlib, %llu damaged streams rejected like zlib\n", good, rejected);
  return 0;
}
