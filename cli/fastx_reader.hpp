// fastx_reader.hpp — the FASTA/FASTQ reader of kmcp-search (header only).  The including program provides die().
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

[[noreturn]] void die(const char* fmt, ...);  // log the message and exit(255), like the reference's checkError

// ------------------------------------------------------------------------------------------------
// FASTA/Q reader (what bio/seqio/fastx delivers to search.go: ID = header up to the first blank, sequence
// with line breaks removed; gzip transparently)
// ------------------------------------------------------------------------------------------------
// One record of a FASTA/FASTQ stream.  The pointers stay valid until the next call of FastxReader::next.
struct FastxRec {
  const char* id = nullptr;
  size_t id_len = 0;
  const char* seq = nullptr;
  size_t seq_len = 0;
};

// Block-buffered FASTA/FASTQ reader (plain files through read(2), gzip through zlib).  A record whose sequence sits on one
// line — every FASTQ in practice — is handed out as pointers into the block buffer: no per-record allocation or copy; wrapped
// sequences (FASTA) are joined in a scratch string.  ID = header up to the first blank (fastx: `record.ID`).
class FastxReader {
 public:
  // KMCP_READER_BUF (bytes) shrinks the initial block buffer so that tests cross refill boundaries with small files
  static size_t initial_buffer() {
    const char* e = getenv("KMCP_READER_BUF");
    const long v = e ? atol(e) : 0;
    return v >= 16 ? (size_t)v : (size_t)(8u << 20);
  }
  explicit FastxReader(const std::string& path) : buf_(initial_buffer()) {
    if (path == "-") {
      gz_ = gzdopen(0, "rb");
      if (!gz_) die("stdin: %s", strerror(errno));
    } else {
      fd_ = open(path.c_str(), O_RDONLY);
      if (fd_ < 0) die("%s: %s", path.c_str(), strerror(errno));
      unsigned char magic[6] = {0, 0, 0, 0, 0, 0};
      const ssize_t got = pread(fd_, magic, 6, 0);
      // the reference's xopen also reads xz, zstd and bzip2; here only gzip is built in: refuse the others instead of parsing noise
      if (got >= 6 && memcmp(magic, "\xfd" "7zXZ\0", 6) == 0) die("%s: xz input is not supported (decompress it: xz -dc file | kmcp-search ... -)", path.c_str());
      if (got >= 4 && memcmp(magic, "\x28\xb5\x2f\xfd", 4) == 0) die("%s: zstd input is not supported (zstd -dc file | kmcp-search ... -)", path.c_str());
      if (got >= 3 && memcmp(magic, "BZh", 3) == 0) die("%s: bzip2 input is not supported (bzip2 -dc file | kmcp-search ... -)", path.c_str());
      if (got < 0 || (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {  // gzip, or not seekable (a pipe): let zlib look
        gz_ = gzdopen(fd_, "rb");
        if (!gz_) die("%s: %s", path.c_str(), strerror(errno));
        fd_ = -1;  // owned by zlib now
      }
    }
    if (gz_) {
      gzbuffer(gz_, 1 << 20);
      // inflate runs ahead on its own thread (4 chunks of 4 MB in flight) while this thread parses
      for (int i = 0; i < 4; i++) spare_.emplace_back(new Chunk());
      inflater_ = std::thread([this] {
        for (;;) {
          std::unique_ptr<Chunk> c;
          {
            std::unique_lock<std::mutex> l(im_);
            icv_.wait(l, [&] { return !spare_.empty() || istop_; });
            if (istop_) return;
            c = std::move(spare_.front());
            spare_.pop_front();
          }
          const int n = gzread(gz_, c->data.data(), (unsigned)c->data.size());
          std::lock_guard<std::mutex> l(im_);
          if (n <= 0) {
            idone_ = true;
            icv_.notify_all();
            return;
          }
          c->n = (size_t)n;
          ready_.push_back(std::move(c));
          icv_.notify_all();
        }
      });
    }
  }
  ~FastxReader() {
    if (inflater_.joinable()) {
      {
        std::lock_guard<std::mutex> l(im_);
        istop_ = true;
        icv_.notify_all();
      }
      inflater_.join();
    }
    if (gz_) gzclose(gz_);
    if (fd_ >= 0) close(fd_);
  }
  FastxReader(const FastxReader&) = delete;
  FastxReader& operator=(const FastxReader&) = delete;

  // returns false at EOF
  bool next(FastxRec* r) {
    size_t lo, ln;
    if (have_next_) {  // a FASTA record ended on this header line; it is still in the buffer
      have_next_ = false;
      keep_ = next_keep_;
      hdr_off_ = next_hdr_off_;
      hdr_len_ = next_hdr_len_;
    } else {
      for (;;) {
        keep_ = pos_;  // nothing before this line is needed any more
        if (!getline(&lo, &ln)) return false;
        if (ln && (buf_[lo] == '>' || buf_[lo] == '@')) break;
      }
      hdr_off_ = lo;
      hdr_len_ = ln;
    }
    const bool fastq = buf_[hdr_off_] == '@';
    size_t e = 1;
    while (e < hdr_len_ && buf_[hdr_off_ + e] != ' ' && buf_[hdr_off_ + e] != '\t') e++;
    const size_t id_len = e - 1;
    bool have_seq = false, joined = false;  // joined: sequence on more than one line, collected in tmp_
    seq_off_ = 0;
    seq_len_ = 0;
    if (!fastq) {
      while (getline(&lo, &ln)) {
        if (ln && buf_[lo] == '>') {  // the next record's header: stays in the buffer for the next call
          have_next_ = true;
          next_keep_ = next_hdr_off_ = lo;
          next_hdr_len_ = ln;
          break;
        }
        append_line(lo, ln, &have_seq, &joined);
      }
    } else {
      // FASTQ: sequence lines up to '+', then as many quality characters as bases
      while (getline(&lo, &ln)) {
        if (ln && buf_[lo] == '+') break;
        append_line(lo, ln, &have_seq, &joined);
      }
      const size_t need = joined ? tmp_.size() : seq_len_;
      size_t q = 0;
      while (q < need && getline(&lo, &ln)) q += ln;
    }
    r->id = buf_.data() + hdr_off_ + 1;
    r->id_len = id_len;
    r->seq = joined ? tmp_.data() : buf_.data() + seq_off_;
    r->seq_len = joined ? tmp_.size() : seq_len_;
    return true;
  }
  // std::string flavour for callers that keep the record
  bool next(std::string* id, std::string* seq) {
    FastxRec r;
    if (!next(&r)) return false;
    id->assign(r.id, r.id_len);
    seq->assign(r.seq, r.seq_len);
    return true;
  }

 private:
  void append_line(size_t lo, size_t ln, bool* have_seq, bool* joined) {
    if (!*have_seq) {
      seq_off_ = lo;
      seq_len_ = ln;
      *have_seq = true;
    } else {
      if (!*joined) {
        tmp_.assign(buf_.data() + seq_off_, seq_len_);
        *joined = true;
      }
      tmp_.append(buf_.data() + lo, ln);
    }
  }
  // next line without its terminator ("\n" or "\r\n") as offset + length into buf_.  Refills keep everything from keep_ on
  // (the current record) and shift the remembered offsets.
  bool getline(size_t* off, size_t* n) {
    for (;;) {
      const char* nl = pos_ < end_ ? (const char*)memchr(buf_.data() + pos_, '\n', end_ - pos_) : nullptr;
      if (nl || (eof_ && pos_ < end_)) {
        size_t len = nl ? (size_t)(nl - (buf_.data() + pos_)) : end_ - pos_;
        *off = pos_;
        pos_ += len + (nl ? 1 : 0);
        while (len && buf_[*off + len - 1] == '\r') len--;
        *n = len;
        return true;
      }
      if (eof_) return false;
      const size_t shift = keep_;
      if (shift) {
        memmove(&buf_[0], buf_.data() + shift, end_ - shift);
        pos_ -= shift;
        end_ -= shift;
        hdr_off_ = hdr_off_ >= shift ? hdr_off_ - shift : 0;
        seq_off_ = seq_off_ >= shift ? seq_off_ - shift : 0;
        keep_ = 0;
      }
      if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);  // a record longer than the buffer (a genome on one line)
      const size_t room = std::min<size_t>(buf_.size() - end_, 1u << 30);
      ssize_t got;
      if (gz_) got = (ssize_t)take_inflated(&buf_[end_], room);
      else
        do got = read(fd_, &buf_[end_], room);
        while (got < 0 && errno == EINTR);
      if (got <= 0) eof_ = true;
      else end_ += (size_t)got;
    }
  }
  // up to `room` inflated bytes from the helper thread's chunks; 0 at the end of the stream
  size_t take_inflated(char* dst, size_t room) {
    if (!cur_ || cur_pos_ == cur_->n) {
      std::unique_lock<std::mutex> l(im_);
      if (cur_) {
        spare_.push_back(std::move(cur_));
        icv_.notify_all();
      }
      icv_.wait(l, [&] { return !ready_.empty() || idone_; });
      if (ready_.empty()) return 0;
      cur_ = std::move(ready_.front());
      ready_.pop_front();
      cur_pos_ = 0;
    }
    const size_t n = std::min(room, cur_->n - cur_pos_);
    memcpy(dst, cur_->data.data() + cur_pos_, n);
    cur_pos_ += n;
    return n;
  }
  struct Chunk {
    std::vector<char> data = std::vector<char>(4u << 20);
    size_t n = 0;
  };
  gzFile gz_ = nullptr;
  int fd_ = -1;
  std::thread inflater_;
  std::mutex im_;
  std::condition_variable icv_;
  std::deque<std::unique_ptr<Chunk>> ready_, spare_;
  std::unique_ptr<Chunk> cur_;
  size_t cur_pos_ = 0;
  bool idone_ = false, istop_ = false;
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0, keep_ = 0;
  bool eof_ = false;
  size_t hdr_off_ = 0, hdr_len_ = 0, seq_off_ = 0, seq_len_ = 0;
  bool have_next_ = false;
  size_t next_hdr_off_ = 0, next_hdr_len_ = 0, next_keep_ = 0;
  std::string tmp_;
};

