// fast_gunzip.hpp — gzip (RFC 1952 / DEFLATE, RFC 1951) decoder for kmcp-search's reader: the reference reads .gz input through
// its gzip package (kmcp/cmd/util-io.go:68-97, xopen); with the search on a GPU a plain zlib inflate of the reads (0.3-0.6 GB/s
// of text) is what bounds everything but the largest databases.  This decoder works on the whole compressed file in memory
// (mmap), keeps 64 bits of input in a register, resolves a literal/length code with one look-up in an 11-bit table (longer codes
// through a second-level table), copies matches eight bytes at a time, and hands out the text in blocks; CRC-32 and length of
// every member are checked (zlib's crc32), members may follow each other, bytes after the last member that do not start a
// member are ignored — as zlib's gzread does.  Truncated or corrupt input is an error, never a short read.
// Checked against zlib on generated and damaged streams by tests/gunzip_check.cpp (ASan/UBSan).
#pragma once
#include <stdint.h>
#include <string.h>
#include <sys/types.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

// CRC-32 (the gzip polynomial) by carry-less multiplication where the CPU has it.  The method is the folding step of Gopal et
// al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ" (Intel, 2009), written down from the paper's algebra:
// a 128-bit accumulator A that stands D bits ahead of the data it is about to meet is replaced by
//     lo64(A) * (x^(D+32) mod P)  xor  hi64(A) * (x^(D-32) mod P)  xor  data
// (bit-reflected arithmetic: the low half of the register holds the EARLIER bytes; the product of two reflected operands comes
// out one bit low, hence the constants are kept shifted left by one).  The constants are not copied from anywhere: xpow_mod()
// below derives x^n mod P for the reflected polynomial 0xEDB88320 at compile time.  Four accumulators fold 64 bytes per turn
// (D = 512), are then folded into one (D = 128), which swallows the remaining 16-byte pieces; the paper's Barrett reduction of
// the last 128 bits is not used — those 16 bytes are simply handed to zlib's crc32 with an all-zero register, which is the same
// number (the accumulator IS a message whose CRC equals that of everything folded into it).  zlib 1.2.11's table-driven crc32
// runs at 1 GB/s — a quarter of the time of the whole decoder; this runs at 10-20 GB/s.  Checked against zlib's crc32 once at
// start-up (and in tests/gunzip_check.cpp): on a mismatch, or without the instruction, zlib's is used.
#if defined(__x86_64__)
#include <immintrin.h>
#include <wmmintrin.h>
namespace fastgz {
// x^n mod P, reflected (bit 31 = x^0 ... bit 0 = x^31), shifted left by one for the carry-less multiplier
constexpr uint64_t xpow_mod(unsigned n) {
  uint32_t v = 0x80000000u;  // x^0
  for (unsigned i = 0; i < n; i++) v = (v >> 1) ^ ((v & 1u) ? 0xEDB88320u : 0u);
  return (uint64_t)v << 1;
}
template <unsigned D>
struct FoldBy {  // multipliers of an accumulator D bits ahead of its data
  static constexpr uint64_t lo = xpow_mod(D + 32), hi = xpow_mod(D - 32);
};
template <unsigned D>
__attribute__((target("pclmul,sse4.1"))) static inline __m128i fold_step(__m128i acc, __m128i data) {
  const __m128i k = _mm_set_epi64x((long long)FoldBy<D>::hi, (long long)FoldBy<D>::lo);
  const __m128i early = _mm_clmulepi64_si128(acc, k, 0x00);  // lo64(acc) * x^(D+32)
  const __m128i late = _mm_clmulepi64_si128(acc, k, 0x11);   // hi64(acc) * x^(D-32)
  return _mm_xor_si128(_mm_xor_si128(early, late), data);
}
// len >= 64 and a multiple of 16; crc is the raw register (the complement of the public value)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(const unsigned char* buf, size_t len, uint32_t crc) {
  const __m128i* p = (const __m128i*)buf;
  __m128i acc[4];
  for (int j = 0; j < 4; j++) acc[j] = _mm_loadu_si128(p + j);
  acc[0] = _mm_xor_si128(acc[0], _mm_cvtsi32_si128((int)crc));  // the register meets the first four bytes
  p += 4;
  size_t left = len - 64;
  for (; left >= 64; left -= 64, p += 4)
    for (int j = 0; j < 4; j++) acc[j] = fold_step<512>(acc[j], _mm_loadu_si128(p + j));
  __m128i a = acc[0];
  for (int j = 1; j < 4; j++) a = fold_step<128>(a, acc[j]);
  for (; left >= 16; left -= 16, p++) a = fold_step<128>(a, _mm_loadu_si128(p));
  alignas(16) unsigned char tail[16];
  _mm_store_si128((__m128i*)tail, a);
  // raw register 0 in = public value 0xFFFFFFFF in; raw register out = complement of the public value out
  return ~(uint32_t)crc32(0xFFFFFFFFu, tail, 16);
}
inline bool fold_usable() {
  static const bool ok = [] {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    const uint32_t seed = 0x12345678u;
    for (size_t n : {64u, 80u, 128u, 208u, 1040u}) {
      unsigned char t[1040];
      for (size_t i = 0; i < n; i++) t[i] = (unsigned char)(i * 151 + 7 + n);
      if ((uint32_t)crc32(seed, t, (uInt)n) != ~crc32_fold(t, n, ~seed)) return false;
    }
    return true;
  }();
  return ok;
}
}  // namespace fastgz
#endif
// crc32() of zlib, faster
inline uint32_t fast_crc32(uint32_t crc, const unsigned char* p, size_t n) {
#if defined(__x86_64__)
  if (n >= 64 && fastgz::fold_usable()) {
    const size_t m = n & ~(size_t)15;
    crc = ~fastgz::crc32_fold(p, m, ~crc);
    p += m;
    n -= m;
  }
#endif
  while (n) {  // zlib takes a 32-bit length
    const size_t m = std::min<size_t>(n, 1u << 30);
    crc = (uint32_t)crc32(crc, p, (uInt)m);
    p += m;
    n -= m;
  }
  return crc;
}

class FastGunzip {
 public:
  FastGunzip(const uint8_t* data, size_t size) : begin_(data), in_(data), in_end_(data + size) {
    buf_.resize(WIN + CAPB + SLACK);
    op_ = rd_ = crc_pos_ = mstart_ = WIN;
  }
  // up to `cap` bytes of text; 0 at the end of the stream; -1 on error (error())
  ssize_t read(char* dst, size_t cap) {
    size_t total = 0;
    while (total < cap) {
      if (rd_ < op_) {
        const size_t n = std::min(cap - total, op_ - rd_);
        memcpy(dst + total, &buf_[rd_], n);
        rd_ += n;
        total += n;
        continue;
      }
      if (done_) break;
      if (!err_.empty()) return -1;
      if (op_ >= WIN + CAPB) slide();
      if (!decode()) return -1;
    }
    return (ssize_t)total;
  }
  const std::string& error() const { return err_; }

 private:
  static constexpr size_t WIN = 32768, CAPB = 1u << 20, SLACK = 1024;
  static constexpr int LB = 11, DB = 8;  // bits of the first-level tables
  // table entry: bits 0-7 code length (bits to drop), 8-10 kind, 11-15 extra bits (or bits of the second-level table), 16-31 value
  enum : uint32_t { LIT = 0, LEN = 1, EOB = 2, SUB = 3, BAD = 4 };
  static uint32_t entry(uint32_t len, uint32_t kind, uint32_t extra, uint32_t val) { return len | (kind << 8) | (extra << 11) | (val << 16); }

  bool fail(const char* m) {
    if (err_.empty()) err_ = m;
    return false;
  }
  void slide() {
    crc_ = fast_crc32(crc_, &buf_[crc_pos_], op_ - crc_pos_);
    msize_ += (uint32_t)(op_ - msize_pos_);
    msize_pos_ = WIN;
    const size_t shift = op_ - WIN;
    memmove(&buf_[0], &buf_[shift], WIN);
    mstart_ = mstart_ > shift ? mstart_ - shift : 0;
    op_ = rd_ = crc_pos_ = WIN;
  }

  // ---- bits ----
  void refill() {
    if (in_end_ - in_ >= 8) {
      uint64_t w;
      memcpy(&w, in_, 8);
      bitbuf_ |= w << bitcnt_;
      in_ += (63 - bitcnt_) >> 3;
      bitcnt_ |= 56;
    } else {
      while (bitcnt_ <= 56) {
        if (in_ < in_end_) bitbuf_ |= (uint64_t)*in_++ << bitcnt_;
        else pad_++;  // zeros past the end: an error if they turn out to be needed
        bitcnt_ += 8;
      }
    }
  }
  uint32_t bits(int n) {
    const uint32_t v = (uint32_t)(bitbuf_ & ((1ull << n) - 1));
    bitbuf_ >>= n;
    bitcnt_ -= n;
    return v;
  }
  // back to bytes: drops the bits up to the next byte boundary and returns the whole bytes still in the register to the input
  bool to_bytes() {
    bits(bitcnt_ & 7);
    size_t whole = (size_t)bitcnt_ >> 3;
    const size_t from_pad = std::min(whole, pad_);
    pad_ -= from_pad;
    whole -= from_pad;
    if (pad_) return fail("unexpected end of file");  // consumed bits that were not there
    in_ -= whole;
    bitbuf_ = 0;
    bitcnt_ = 0;
    return true;
  }

  // ---- tables ----
  // canonical Huffman code of lens[0..n) -> two-level table (tb first-level bits); base/extra describe the symbols from `first`
  // on (length or distance codes), symbols below are literals, `eob` is the end-of-block symbol (-1: none)
  bool build(const uint8_t* lens, int n, int tb, std::vector<uint32_t>& tab, int first, int eob, const uint16_t* base, const uint8_t* extra,
             int n_coded, bool is_codes, bool pairs = false) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 15;
    while (maxlen > 0 && !count[maxlen]) maxlen--;
    tab.assign((size_t)1 << tb, entry(1, BAD, 0, 0));
    if (maxlen == 0) return is_codes ? fail("invalid code lengths set") : true;  // no symbols: an error only when one is used (zlib)
    int left = 1;
    for (int l = 1; l <= 15; l++) {
      left = (left << 1) - count[l];
      if (left < 0) return fail("over-subscribed code");
    }
    if (left > 0 && (is_codes || maxlen != 1)) return fail("incomplete code");  // a single one-bit code is accepted, as in zlib
    uint16_t next[16];
    next[1] = 0;
    for (int l = 1; l < 15; l++) next[l + 1] = (uint16_t)((next[l] + count[l]) << 1);
    // second-level tables: bits needed per first-level prefix
    std::vector<uint8_t>& sub_bits = sub_bits_;  // scratch kept between blocks (a block header should not cost heap calls)
    std::vector<uint16_t>& code = code_;
    code.assign((size_t)n, 0);
    sub_bits.clear();
    if (maxlen > tb) sub_bits.assign((size_t)1 << tb, 0);
    for (int s = 0; s < n; s++) {
      const int l = lens[s];
      if (!l) continue;
      uint32_t c = next[l]++, r = 0;
      for (int i = 0; i < l; i++) r |= ((c >> i) & 1u) << (l - 1 - i);  // codes are sent most significant bit first
      code[(size_t)s] = (uint16_t)r;
      if (l > tb) {
        uint8_t& b = sub_bits[r & ((1u << tb) - 1)];
        b = std::max<uint8_t>(b, (uint8_t)(l - tb));
      }
    }
    if (maxlen > tb)
      for (size_t p = 0; p < sub_bits.size(); p++)
        if (sub_bits[p]) {
          const size_t start = tab.size();
          if (start + ((size_t)1 << sub_bits[p]) > 65535) return fail("code table too large");
          tab[p] = entry((uint32_t)tb, SUB, sub_bits[p], (uint32_t)start);
          tab.resize(start + ((size_t)1 << sub_bits[p]), entry(1, BAD, 0, 0));
        }
    for (int s = 0; s < n; s++) {
      const int l = lens[s];
      if (!l) continue;
      uint32_t e;
      if (s == eob) e = entry((uint32_t)l, EOB, 0, 0);
      else if (s < first) e = entry((uint32_t)l, LIT, 0, (uint32_t)s);
      else if (s - first >= n_coded) e = entry((uint32_t)l, BAD, 0, 0);  // length codes 286/287, distance codes 30/31
      else e = entry((uint32_t)l, LEN, extra[s - first], base[s - first]);
      const uint32_t r = code[(size_t)s];
      if (l <= tb) {
        for (size_t i = r; i < ((size_t)1 << tb); i += (size_t)1 << l) tab[i] = e;
      } else {
        const uint32_t p = r & ((1u << tb) - 1);
        const uint32_t sb = (tab[p] >> 11) & 31, start = tab[p] >> 16;
        for (size_t i = r >> tb; i < ((size_t)1 << sb); i += (size_t)1 << (l - tb)) tab[start + i] = e;
      }
    }
    if (pairs) {
      // two literals in one look-up where both codes fit the first-level index: the entry of index i starts with literal a of
      // l1 bits; the tb - l1 bits after it select entry i >> l1 (entries repeat over the bits they do not use), and if that is a
      // literal b of at most tb - l1 bits, index i decodes "ab" (value a | b << 8, l1 + l2 bits, extra field 1 = one more byte)
      std::vector<uint32_t>& single = single_;
      single.assign(tab.begin(), tab.begin() + ((size_t)1 << tb));
      for (size_t i = 0; i < ((size_t)1 << tb); i++) {
        const uint32_t e1 = single[i];
        if (((e1 >> 8) & 7) != LIT) continue;
        const uint32_t l1 = e1 & 255;
        if (l1 >= (uint32_t)tb) continue;
        const uint32_t e2 = single[i >> l1];
        if (((e2 >> 8) & 7) != LIT || (e2 & 255) > (uint32_t)tb - l1) continue;
        tab[i] = entry(l1 + (e2 & 255), LIT, 1, (e1 >> 16) | ((e2 >> 16) << 8));
      }
    }
    return true;
  }
  bool fixed_tables() {
    uint8_t l[288], d[32];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    for (int i = 0; i < 32; i++) d[i] = 5;
    return build(l, 288, LB, lt_, 257, 256, kLenBase, kLenExtra, 29, false, true) && build(d, 32, DB, dt_, 0, -1, kDistBase, kDistExtra, 30, false);
  }
  bool dynamic_tables() {
    refill();
    const int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
    if (hlit > 286 || hdist > 30) return fail("too many length or distance symbols");
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) {
      if (bitcnt_ < 3) refill();
      cl[order[i]] = (uint8_t)bits(3);
    }
    // the code-length code: at most 7 bits, one level
    std::vector<uint32_t>& pt = pt_;
    static const uint16_t nob[19] = {0};
    static const uint8_t noe[19] = {0};
    {
      // as literals 0..18 (first = 19: no length symbols)
      if (!build(cl, 19, 7, pt, 19, -1, nob, noe, 0, true)) return false;
    }
    uint8_t lens[320];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
      refill();
      if (pad_ > 8) return fail("unexpected end of file");
      const uint32_t e = pt[bitbuf_ & 127];
      if (((e >> 8) & 7) != LIT) return fail("invalid code lengths set");
      bits((int)(e & 255));
      const int sym = (int)(e >> 16);
      if (sym < 16) lens[i++] = (uint8_t)sym;
      else {
        int rep, val = 0;
        if (sym == 16) {
          if (i == 0) return fail("invalid bit length repeat");
          val = lens[i - 1];
          rep = 3 + (int)bits(2);
        } else if (sym == 17) rep = 3 + (int)bits(3);
        else rep = 11 + (int)bits(7);
        if (i + rep > total) return fail("invalid bit length repeat");
        while (rep--) lens[i++] = (uint8_t)val;
      }
    }
    if (lens[256] == 0) return fail("invalid code -- missing end-of-block");
    return build(lens, hlit, LB, lt_, 257, 256, kLenBase, kLenExtra, 29, false, true) && build(lens + hlit, hdist, DB, dt_, 0, -1, kDistBase, kDistExtra, 30, false);
  }

  // ---- the stream ----
  enum State { HEADER, BLOCK, STORED, HUFF, TRAILER };
  bool header() {
    const uint8_t* p = in_;
    const size_t n = (size_t)(in_end_ - in_);
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b) return fail(first_member_ ? "not in gzip format" : "unexpected end of file");
    if (p[2] != 8) return fail("unknown compression method");
    const int flg = p[3];
    if (flg & 0xe0) return fail("unknown header flags set");
    size_t o = 10;
    if (flg & 4) {
      if (o + 2 > n) return fail("unexpected end of file");
      o += 2 + ((size_t)p[o] | ((size_t)p[o + 1] << 8));
    }
    for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
      if (flg & f) {
        while (o < n && p[o]) o++;
        o++;
      }
    if (flg & 2) {  // FHCRC: the low 16 bits of the CRC-32 of the header so far
      if (o + 2 > n) return fail("unexpected end of file");
      const uint32_t want = (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8);
      if (((uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)o) & 0xffffu) != want) return fail("header crc mismatch");
      o += 2;
    }
    if (o > n) return fail("unexpected end of file");
    in_ += o;
    bitbuf_ = 0;
    bitcnt_ = 0;
    pad_ = 0;
    crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
    crc_pos_ = op_;
    mstart_ = op_;
    msize_ = 0;
    msize_pos_ = op_;
    first_member_ = false;
    state_ = BLOCK;
    return true;
  }
  bool trailer() {
    if (!to_bytes()) return false;
    if (in_end_ - in_ < 8) return fail("unexpected end of file");
    crc_ = fast_crc32(crc_, &buf_[crc_pos_], op_ - crc_pos_);
    crc_pos_ = op_;
    msize_ += (uint32_t)(op_ - msize_pos_);
    msize_pos_ = op_;
    uint32_t c, s;
    memcpy(&c, in_, 4);
    memcpy(&s, in_ + 4, 4);
    in_ += 8;
    if (c != crc_) return fail("incorrect data check");
    if (s != msize_) return fail("incorrect length check");
    if (in_end_ - in_ >= 2 && in_[0] == 0x1f && in_[1] == 0x8b) state_ = HEADER;  // the next member
    else done_ = true;                                                             // the end, or bytes that are not a member: ignored
    return true;
  }
  // runs until the block buffer is full, the stream ends, or an error; false on error
  bool decode() {
    for (;;) {
      if (op_ >= WIN + CAPB) return true;
      switch (state_) {
        case HEADER:
          if (!header()) return false;
          break;
        case BLOCK: {
          refill();
          if (pad_ > 8) return fail("unexpected end of file");
          final_ = bits(1) != 0;
          const uint32_t type = bits(2);
          if (type == 0) {
            if (!to_bytes()) return false;
            if (in_end_ - in_ < 4) return fail("unexpected end of file");
            const uint32_t len = (uint32_t)in_[0] | ((uint32_t)in_[1] << 8), nlen = (uint32_t)in_[2] | ((uint32_t)in_[3] << 8);
            if ((len ^ 0xffffu) != nlen) return fail("invalid stored block lengths");
            in_ += 4;
            stored_ = len;
            state_ = STORED;
          } else if (type == 1) {
            if (!fixed_tables()) return false;
            state_ = HUFF;
          } else if (type == 2) {
            if (!dynamic_tables()) return false;
            state_ = HUFF;
          } else return fail("invalid block type");
          break;
        }
        case STORED: {
          const size_t n = std::min<size_t>(stored_, WIN + CAPB - op_);
          if ((size_t)(in_end_ - in_) < n) return fail("unexpected end of file");
          memcpy(&buf_[op_], in_, n);
          in_ += n;
          op_ += n;
          stored_ -= (uint32_t)n;
          if (stored_ == 0) state_ = final_ ? TRAILER : BLOCK;
          break;
        }
        case HUFF:
          if (!huff()) return false;
          break;
        case TRAILER:
          if (!trailer()) return false;
          if (done_) return true;
          break;
      }
    }
  }
  // literals and matches of the current block until the buffer is full or the block ends.  The bit register, the input and
  // output positions live in locals here (stores through the output pointer may alias members, which would force them through
  // memory on every symbol).  Far from the end of the input the register is topped up without looking (FAST); within its last
  // bytes every top-up checks, and zeros are supplied past the end (an error if they turn out to be consumed).
  bool huff() {
    uint8_t* const out0 = &buf_[0];
    size_t op = op_;
    const size_t limit = WIN + CAPB;
    const uint32_t* const lt = lt_.data();
    const uint32_t* const dt = dt_.data();
    uint64_t bb = bitbuf_;
    int bc = bitcnt_;
    const uint8_t* in = in_;
    const uint8_t* const in_fast_end = in_end_ - in_ >= 16 ? in_end_ - 16 : in_;
    const size_t mstart = mstart_;
    const char* why = nullptr;
    bool block_done = false;
    auto sync_out = [&] {
      bitbuf_ = bb;
      bitcnt_ = bc;
      in_ = in;
    };
    while (op < limit) {
      if (in < in_fast_end) {
        uint64_t w;
        memcpy(&w, in, 8);
        bb |= w << bc;
        in += (63 - bc) >> 3;
        bc |= 56;
      } else {
        sync_out();
        refill();
        bb = bitbuf_;
        bc = bitcnt_;
        in = in_;
        if (pad_ > 64) { why = "unexpected end of file"; break; }  // far past the end of the input and still no end of block
      }
      uint32_t e = lt[bb & ((1u << LB) - 1)];
      // a literal entry holds one byte or two (build: pairs): both are stored, the position moves by 1 + its extra field
#define FASTGZ_PUT_LITERALS(e)                      \
  do {                                              \
    out0[op] = (uint8_t)((e) >> 16);                \
    out0[op + 1] = (uint8_t)((e) >> 24);            \
    op += 1 + (((e) >> 11) & 1);                    \
  } while (0)
      if (((e >> 8) & 7) == LIT) {  // up to three literal look-ups per top-up: 3 x 15 bits fit the 56 that are there
        bb >>= (e & 255);
        bc -= (int)(e & 255);
        FASTGZ_PUT_LITERALS(e);
        e = lt[bb & ((1u << LB) - 1)];
        if (((e >> 8) & 7) == LIT) {
          bb >>= (e & 255);
          bc -= (int)(e & 255);
          FASTGZ_PUT_LITERALS(e);
          e = lt[bb & ((1u << LB) - 1)];
          if (((e >> 8) & 7) == LIT) {
            bb >>= (e & 255);
            bc -= (int)(e & 255);
            FASTGZ_PUT_LITERALS(e);
            continue;
          }
        }
        if (bc < 48) continue;  // not enough left for a length and a distance: top up first (the entry is looked up again)
      }
      if (((e >> 8) & 7) == SUB) e = lt[(e >> 16) + ((bb >> LB) & ((1u << ((e >> 11) & 31)) - 1))];
      bb >>= (e & 255);
      bc -= (int)(e & 255);
      const uint32_t kind = (e >> 8) & 7;
      if (kind == LIT) {
        FASTGZ_PUT_LITERALS(e);
        continue;
      }
#undef FASTGZ_PUT_LITERALS
      if (kind == LEN) {
        const uint32_t xl = (e >> 11) & 31;
        const size_t len = (e >> 16) + (uint32_t)(bb & ((1ull << xl) - 1));
        bb >>= xl;
        bc -= (int)xl;
        uint32_t d = dt[bb & ((1u << DB) - 1)];
        if (((d >> 8) & 7) == SUB) d = dt[(d >> 16) + ((bb >> DB) & ((1u << ((d >> 11) & 31)) - 1))];
        bb >>= (d & 255);
        bc -= (int)(d & 255);
        if (((d >> 8) & 7) != LEN) { why = "invalid distance code"; break; }
        const uint32_t xd = (d >> 11) & 31;
        const size_t dist = (d >> 16) + (uint32_t)(bb & ((1ull << xd) - 1));
        bb >>= xd;
        bc -= (int)xd;
        if (dist > op - mstart) { why = "invalid distance too far back"; break; }
        uint8_t* dst = out0 + op;
        const uint8_t* src = dst - dist;
        op += len;
        if (dist >= 8) {  // eight bytes at a time; the buffer has slack for the overshoot
          uint8_t* const end = dst + len;
          do {
            uint64_t w;
            memcpy(&w, src, 8);
            memcpy(dst, &w, 8);
            src += 8;
            dst += 8;
          } while (dst < end);
        } else {
          for (size_t i = 0; i < len; i++) dst[i] = src[i];
        }
        continue;
      }
      if (kind == EOB) {
        block_done = true;
        break;
      }
      why = "invalid literal/length code";
      break;
    }
    sync_out();
    op_ = op;
    if (why) return fail(why);
    if (bitcnt_ < 0) return fail("unexpected end of file");
    if (block_done) {
      if (pad_ > 8) return fail("unexpected end of file");
      state_ = final_ ? TRAILER : BLOCK;
    }
    return true;
  }

  static constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static constexpr uint16_t kDistBase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                             193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

  const uint8_t *begin_, *in_, *in_end_;
  uint64_t bitbuf_ = 0;
  int bitcnt_ = 0;
  size_t pad_ = 0;
  std::vector<uint8_t> buf_;  // [0, WIN): the last 32 KB handed out; [WIN, op_): text of this block
  size_t op_, rd_, crc_pos_, mstart_, msize_pos_ = 0;
  uint32_t crc_ = 0, msize_ = 0, stored_ = 0;
  bool final_ = false, done_ = false, first_member_ = true;
  State state_ = HEADER;
  std::vector<uint32_t> lt_, dt_;
  std::vector<uint8_t> sub_bits_;   // scratch of build()
  std::vector<uint16_t> code_;
  std::vector<uint32_t> single_, pt_;
  std::string err_;
};
