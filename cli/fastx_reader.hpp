// fastx_reader.hpp — the FASTA/FASTQ reader of kmcp-search (header only).  The including program provides die().
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
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

#include "fast_gunzip.hpp"

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

// BGZF (block gzip: bgzip / htslib output — gzip members of at most 64 KB that carry their compressed size in a "BC" extra
// field) inflated by several threads: one thread walks the block headers and cuts the file into tasks of ~4 MB, workers
// inflate the blocks of a task independently (every block's output size is in its trailer, so the offsets are known up
// front) and check CRC32/ISIZE, the consumer takes the tasks back in file order.  Plain gzip has no such structure and stays
// on the single inflate thread of FastxReader.
class BgzfInflater {
 public:
  static bool detect(int fd) {
    unsigned char h[18];
    if (pread(fd, h, 18, 0) != 18) return false;
    return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
  }
  BgzfInflater(int fd, const std::string& path, int workers) : path_(path) {
    f_ = fdopen(fd, "rb");
    if (!f_) die("%s: %s", path.c_str(), strerror(errno));
    setvbuf(f_, nullptr, _IOFBF, 1 << 20);
    walker_ = std::thread([this] { walk(); });
    for (int i = 0; i < std::max(1, workers); i++) pool_.emplace_back([this] { work(); });
  }
  ~BgzfInflater() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
      cv_.notify_all();
    }
    walker_.join();
    for (auto& t : pool_) t.join();
    fclose(f_);
  }
  // up to `room` inflated bytes in file order; 0 at the end of the file
  size_t read(char* dst, size_t room) {
    if (!cur_ || cur_pos_ == cur_->out.size()) {
      std::unique_lock<std::mutex> l(m_);
      for (;;) {
        cur_.reset();
        cv_.wait(l, [&] { return !err_.empty() || (!order_.empty() && order_.front()->done) || (order_.empty() && walked_); });
        if (!err_.empty()) die("%s: %s", path_.c_str(), err_.c_str());
        if (order_.empty()) return 0;
        cur_ = std::move(order_.front());
        order_.pop_front();
        cv_.notify_all();  // room for the walker
        cur_pos_ = 0;
        if (!cur_->out.empty()) break;  // (a task of empty blocks, e.g. the EOF marker alone)
      }
    }
    const size_t n = std::min(room, cur_->out.size() - cur_pos_);
    memcpy(dst, cur_->out.data() + cur_pos_, n);
    cur_pos_ += n;
    return n;
  }

 private:
  struct Block {
    size_t coff, clen, ooff;  // deflate bytes in Task::comp, output offset
    uint32_t crc, isize;
  };
  struct Task {
    std::vector<unsigned char> comp;
    std::vector<Block> blocks;
    std::vector<char> out;
    bool done = false;
  };
  void fail(const std::string& e) {
    std::lock_guard<std::mutex> l(m_);
    if (err_.empty()) err_ = e;
    walked_ = true;
    cv_.notify_all();
  }
  void walk() {
    for (;;) {
      std::shared_ptr<Task> t(new Task());
      size_t out_total = 0;
      while (t->comp.size() < (4u << 20) && t->blocks.size() < 512) {
        unsigned char h[12];
        const size_t got = fread(h, 1, 12, f_);
        if (got == 0) break;  // end of file at a block boundary
        if (got != 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail("corrupt BGZF block header");
        const size_t xlen = h[10] | ((size_t)h[11] << 8);
        unsigned char extra[65536];
        if (fread(extra, 1, xlen, f_) != xlen) return fail("truncated BGZF block");
        long bsize = -1;
        for (size_t p = 0; p + 4 <= xlen;) {
          const size_t sl = extra[p + 2] | ((size_t)extra[p + 3] << 8);
          if (extra[p] == 'B' && extra[p + 1] == 'C' && sl == 2 && p + 6 <= xlen) bsize = (long)(extra[p + 4] | ((size_t)extra[p + 5] << 8));
          p += 4 + sl;
        }
        const long rest = bsize + 1 - 12 - (long)xlen;  // deflate data + CRC32 + ISIZE
        if (bsize < 0 || rest < 8) return fail("corrupt BGZF block header");
        const size_t at = t->comp.size();
        t->comp.resize(at + (size_t)rest);
        if (fread(t->comp.data() + at, 1, (size_t)rest, f_) != (size_t)rest) return fail("truncated BGZF block");
        const unsigned char* tr = t->comp.data() + at + rest - 8;
        Block b;
        b.coff = at;
        b.clen = (size_t)rest - 8;
        b.crc = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24);
        b.isize = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (b.isize > 65536) return fail("corrupt BGZF block (ISIZE > 64 KB)");
        b.ooff = out_total;
        out_total += b.isize;
        t->blocks.push_back(b);
      }
      if (t->blocks.empty()) break;
      t->out.resize(out_total);
      std::unique_lock<std::mutex> l(m_);
      cv_.wait(l, [&] { return stop_ || order_.size() < 12; });
      if (stop_) return;
      order_.push_back(t);
      todo_.push_back(t);
      cv_.notify_all();
    }
    std::lock_guard<std::mutex> l(m_);
    walked_ = true;
    cv_.notify_all();
  }
  void work() {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) return fail("zlib initialisation failed");
    for (;;) {
      std::shared_ptr<Task> t;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || !todo_.empty() || walked_; });
        if (stop_ || (todo_.empty() && walked_)) break;
        if (todo_.empty()) continue;
        t = std::move(todo_.front());
        todo_.pop_front();
      }
      for (const Block& b : t->blocks) {
        inflateReset(&z);
        z.next_in = t->comp.data() + b.coff;
        z.avail_in = (uInt)b.clen;
        z.next_out = (Bytef*)t->out.data() + b.ooff;
        z.avail_out = b.isize;
        const int rc = inflate(&z, Z_FINISH);
        if (rc != Z_STREAM_END || z.avail_out != 0 || crc32(crc32(0L, Z_NULL, 0), (const Bytef*)t->out.data() + b.ooff, b.isize) != b.crc) {
          inflateEnd(&z);
          return fail("corrupt BGZF block (inflate / CRC32 mismatch)");
        }
      }
      std::vector<unsigned char>().swap(t->comp);
      std::lock_guard<std::mutex> l(m_);
      t->done = true;
      cv_.notify_all();
    }
    inflateEnd(&z);
  }
  std::string path_;
  FILE* f_ = nullptr;
  std::thread walker_;
  std::vector<std::thread> pool_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Task>> order_, todo_;
  std::shared_ptr<Task> cur_;
  size_t cur_pos_ = 0;
  bool walked_ = false, stop_ = false;
  std::string err_;
};

// Block-buffered FASTA/FASTQ reader (plain files through read(2), gzip through zlib, BGZF through BgzfInflater).  A record whose sequence sits on one
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
  // the records of a memory range (it must outlive the reader): what a worker of ParallelFastq falls back to
  FastxReader(const char* data, size_t n) : path_("<memory>"), buf_(data, data + n) {
    end_ = n;
    eof_ = true;
  }
  // `offset`: start reading a plain (uncompressed, seekable) file there
  explicit FastxReader(const std::string& path, uint64_t offset = 0) : path_(path), buf_(initial_buffer()) {
    if (path == "-") {
      gz_ = gzdopen(0, "rb");
      if (!gz_) die("stdin: %s", strerror(errno));
    } else {
      fd_ = open(path.c_str(), O_RDONLY);
      if (fd_ < 0) die("%s: %s", path.c_str(), strerror(errno));
      unsigned char magic[6] = {0, 0, 0, 0, 0, 0};
      const ssize_t got = pread(fd_, magic, 6, 0);
      // the reference's xopen also reads xz, zstd and bzip2 (util-io.go:68-97).  gzip is built in; for the others the system's
      // own decompressor is run and its output parsed like a plain stream (it must be on PATH; its exit status is checked)
      const char* tool = nullptr;
      if (got >= 6 && memcmp(magic, "\xfd" "7zXZ\0", 6) == 0) tool = "xz";
      else if (got >= 4 && memcmp(magic, "\x28\xb5\x2f\xfd", 4) == 0) tool = "zstd";
      else if (got >= 3 && memcmp(magic, "BZh", 3) == 0) tool = "bzip2";
      if (tool) {
        close(fd_);
        fd_ = spawn_decompressor(tool, path);
      } else if (got >= 6 && BgzfInflater::detect(fd_)) {  // block gzip: several inflate threads
        int w = (int)std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 4));
        if (const char* e = getenv("KMCP_BGZF_THREADS")) w = std::max(1, atoi(e));
        bgzf_.reset(new BgzfInflater(fd_, path, w));
        fd_ = -1;  // owned by the inflater now
      } else if (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b && map_gzip()) {  // a gzip file: the in-memory decoder (fast_gunzip.hpp)
      } else if (got < 0 || (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {  // gzip that cannot be mapped, or not seekable (a pipe): let zlib look
        gz_ = gzdopen(fd_, "rb");
        if (!gz_) die("%s: %s", path.c_str(), strerror(errno));
        fd_ = -1;  // owned by zlib now
      } else if (offset && lseek(fd_, (off_t)offset, SEEK_SET) < 0) {
        die("%s: %s", path.c_str(), strerror(errno));
      }
    }
    if (gz_ || fgz_) {
      if (gz_) gzbuffer(gz_, 1 << 20);
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
          const ssize_t n = fgz_ ? fgz_->read(c->data.data(), c->data.size()) : (ssize_t)gzread(gz_, c->data.data(), (unsigned)c->data.size());
          std::lock_guard<std::mutex> l(im_);
          if (n <= 0) {
            // a truncated or corrupt stream must not look like a short input (the reference's gzip reader aborts with
            // "unexpected EOF"): zlib reports Z_BUF_ERROR for a stream that ends inside a member, Z_DATA_ERROR for a bad CRC
            if (fgz_) {
              if (n < 0) ierr_ = fgz_->error().empty() ? "corrupt gzip stream" : fgz_->error();
            } else {
              int errnum = Z_OK;
              const char* msg = gzerror(gz_, &errnum);
              if (n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) ierr_ = errnum == Z_ERRNO ? strerror(errno) : (msg && *msg ? msg : "corrupt gzip stream");
            }
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
    fgz_.reset();
    if (gzmap_) munmap((void*)gzmap_, gzmap_size_);
    if (fd_ >= 0) close(fd_);
    if (child_ > 0) {  // reader dropped before the stream ended
      kill(child_, SIGTERM);
      waitpid(child_, nullptr, 0);
    }
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
      if (bgzf_) got = (ssize_t)bgzf_->read(&buf_[end_], room);
      else if (gz_ || fgz_) got = (ssize_t)take_inflated(&buf_[end_], room);
      else
        do got = read(fd_, &buf_[end_], room);
        while (got < 0 && errno == EINTR);
      if (got < 0 && !bgzf_ && !gz_ && !fgz_) die("%s: %s", path_.c_str(), strerror(errno));
      if (got <= 0 && child_ > 0) {  // the decompressor's verdict on the file
        int st = 0;
        waitpid(child_, &st, 0);
        child_ = -1;
        if (WIFEXITED(st) && WEXITSTATUS(st) == 127) die("%s: this is %s input and `%s` is not on PATH (decompress it: %s -dc file | kmcp-search ... -)", path_.c_str(), tool_.c_str(), tool_.c_str(), tool_.c_str());
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) die("%s: %s failed (corrupt or truncated input?)", path_.c_str(), tool_.c_str());
      }
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
      if (ready_.empty()) {
        if (!ierr_.empty()) die("%s: %s", path_.c_str(), ierr_.c_str());
        return 0;
      }
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
  // maps a regular gzip file for FastGunzip; false (nothing changed) if it cannot be mapped or KMCP_ZLIB_GUNZIP asks for zlib
  bool map_gzip() {
    if (getenv("KMCP_ZLIB_GUNZIP")) return false;
    struct stat st;
    if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) return false;
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) return false;
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    gzmap_ = (const uint8_t*)m;
    gzmap_size_ = (size_t)st.st_size;
    fgz_.reset(new FastGunzip(gzmap_, gzmap_size_));
    return true;
  }
  // `tool -dc path` with its stdout on a pipe; returns the read end
  int spawn_decompressor(const char* tool, const std::string& path) {
    int pfd[2];
    if (pipe(pfd) != 0) die("%s: pipe: %s", path.c_str(), strerror(errno));
    const pid_t pid = fork();
    if (pid < 0) die("%s: fork: %s", path.c_str(), strerror(errno));
    if (pid == 0) {
      dup2(pfd[1], 1);
      close(pfd[0]);
      close(pfd[1]);
      execlp(tool, tool, "-dc", "--", path.c_str(), (char*)nullptr);
      _exit(127);  // not on PATH
    }
    close(pfd[1]);
    child_ = pid;
    tool_ = tool;
    // a missing tool shows as exit status 127 with no output: tell it apart from an empty file right away
    return pfd[0];
  }
  std::string path_;
  std::string tool_;
  pid_t child_ = -1;
  std::string ierr_;  // set by the inflate thread
  gzFile gz_ = nullptr;
  std::unique_ptr<FastGunzip> fgz_;  // ... or the in-memory decoder over the mapped file
  const uint8_t* gzmap_ = nullptr;
  size_t gzmap_size_ = 0;
  int fd_ = -1;
  std::unique_ptr<BgzfInflater> bgzf_;
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


// ------------------------------------------------------------------------------------------------
// Plain (uncompressed) four-line FASTQ parsed by several threads.
//
// One thread tops out near 8 M reads/s of 150-bp FASTQ (memchr per line + copying IDs and bases into the batch); the GPU
// takes 13–45 M reads/s on small databases.  A regular file is mapped, cut into chunks of about `chunk_reads` records at record
// boundaries, and every chunk is parsed by a worker straight into the flat buffers a GPU batch consists of (IDs back to back
// + offsets, bases back to back + offsets) — the consumer moves them into a batch without touching the bytes again.
//
// Where a record starts cannot be told from one line ('@' also starts quality lines), so the boundary rule is: a line that
// starts with '@', whose second next line starts with '+', and whose next and third next lines have equal lengths.  In a
// strict four-line file (header, bases, '+', qualities — what every short-read FASTQ is) a quality line can never pass for a
// header, because two lines after it comes a line of bases.  The workers insist on that structure for every record; a file
// that wraps its sequences (or breaks the rule anywhere) makes next() report the offset of the chunk it happened in, and the
// caller goes on from there with the general single-threaded FastxReader — same records, just slower.
// ------------------------------------------------------------------------------------------------
struct FastqChunk {
  std::vector<char> id_buf;            // IDs back to back (header up to the first blank)
  std::vector<uint64_t> id_offs{0};
  std::vector<uint8_t> seqs;           // bases back to back
  std::vector<uint64_t> offs{0};
  uint64_t file_off = 0;               // where the chunk starts in the file
  bool strict = true;                  // false: the chunk is not strict four-line FASTQ: resume serially at file_off
  bool done = false;
  size_t size() const { return offs.size() - 1; }
};

// The vectors of chunks that have been searched and printed go round: a batch is ~20 MB of bases, IDs and offsets, and a fresh set per
// batch is mapped, page-faulted in by the parser and unmapped by the writer (TLB shootdowns across every thread of the process) — at
// 80 batches per second that was a fifth of the search phase (profiles/r06_cli_e2e.txt).  take() hands out a cleared set that keeps its
// capacity, give() takes one back; a handful are kept.
class ChunkPool {
 public:
  static ChunkPool& get() {
    static ChunkPool* p = new ChunkPool();
    return *p;
  }
  std::unique_ptr<FastqChunk> take() {
    {
      std::lock_guard<std::mutex> l(m_);
      if (!free_.empty()) {
        std::unique_ptr<FastqChunk> c = std::move(free_.back());
        free_.pop_back();
        return c;
      }
    }
    return std::unique_ptr<FastqChunk>(new FastqChunk());
  }
  void give(std::vector<char>& id_buf, std::vector<uint64_t>& id_offs, std::vector<uint8_t>& seqs, std::vector<uint64_t>& offs) {
    if (seqs.capacity() > ((size_t)256 << 20)) return;  // an unusually large batch (long reads): let it go
    std::unique_ptr<FastqChunk> c(new FastqChunk());
    c->id_buf.swap(id_buf);
    c->id_offs.swap(id_offs);
    c->seqs.swap(seqs);
    c->offs.swap(offs);
    c->id_buf.clear();
    c->id_offs.assign(1, 0);
    c->seqs.clear();
    c->offs.assign(1, 0);
    std::lock_guard<std::mutex> l(m_);
    if (free_.size() < 48) free_.push_back(std::move(c));
  }

 private:
  std::mutex m_;
  std::vector<std::unique_ptr<FastqChunk>> free_;
};

class ParallelFastq {
 public:
  // a regular, uncompressed file that starts with '@' and whose first records are strict four-line FASTQ
  static bool eligible(const std::string& path, uint64_t min_bytes = 8u << 20) {
    if (getenv("KMCP_SERIAL_READER")) return false;
    if (const char* e = getenv("KMCP_PARALLEL_MIN_BYTES")) min_bytes = (uint64_t)atoll(e);  // tests: small files through the parallel path
    if (path == "-") return false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (uint64_t)st.st_size >= min_bytes;
    if (ok) {
      std::vector<char> head((size_t)std::min<uint64_t>((uint64_t)st.st_size, 1u << 20));
      const ssize_t got = pread(fd, head.data(), head.size(), 0);
      ok = got > 0 && head[0] == '@';
      if (ok) {
        FastqChunk c;
        size_t used = 0;
        ok = parse_strict(head.data(), (size_t)got, (uint64_t)got == (uint64_t)st.st_size, &c, &used) && c.size() >= 1;
      }
    }
    close(fd);
    return ok;
  }

  // chunks of about `chunk_reads` records, and of at most `max_chunk_bytes` of file (0: no limit) — long reads
  ParallelFastq(const std::string& path, size_t chunk_reads, int workers, size_t max_chunk_bytes = 0) : path_(path) {
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ < 0) die("%s: %s", path.c_str(), strerror(errno));
    struct stat st;
    if (fstat(fd_, &st) != 0) die("%s: %s", path.c_str(), strerror(errno));
    size_ = (size_t)st.st_size;
    if (size_) {
      void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
      if (m == MAP_FAILED) die("%s: mmap: %s", path.c_str(), strerror(errno));
      map_ = (const char*)m;
      madvise((void*)map_, size_, MADV_SEQUENTIAL);
    }
    // bytes per record from the first few
    size_t recs = 0, p = 0;
    while (recs < 64 && p < size_) {
      int lines = 0;
      while (lines < 4 && p < size_) {
        const char* nl = (const char*)memchr(map_ + p, '\n', size_ - p);
        p = nl ? (size_t)(nl - map_) + 1 : size_;
        lines++;
      }
      recs++;
    }
    const size_t per_rec = recs ? std::max<size_t>(1, p / recs) : 320;
    chunk_bytes_ = std::max<size_t>(1u << 16, per_rec * std::max<size_t>(1, chunk_reads));
    if (max_chunk_bytes) chunk_bytes_ = std::min(chunk_bytes_, std::max<size_t>(1u << 16, max_chunk_bytes));
    if (const char* e = getenv("KMCP_READER_CHUNK")) chunk_bytes_ = std::max<size_t>(64, (size_t)atol(e));
    for (int i = 0; i < std::max(1, workers); i++) pool_.emplace_back([this] { work(); });
    cutter_ = std::thread([this] { cut(); });
  }
  ~ParallelFastq() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
      cv_.notify_all();
    }
    cutter_.join();
    for (auto& t : pool_) t.join();
    if (map_) munmap((void*)map_, size_);
    close(fd_);
  }
  // chunks in file order; nullptr at the end.  A chunk with strict == false carries no records: resume at its file_off with
  // FastxReader(path, file_off) (later chunks are discarded).
  std::unique_ptr<FastqChunk> next() {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return (!order_.empty() && order_.front()->c->done) || (order_.empty() && cut_done_); });
    if (order_.empty()) return nullptr;
    std::unique_ptr<FastqChunk> c = std::move(order_.front()->c);
    order_.pop_front();
    cv_.notify_all();
    return c;
  }

  // strict four-line parse of [p, p+n).  Returns false on the first line that breaks the four-line structure; *used = bytes
  // consumed by complete records (a last record without a final newline is complete when at_eof).
  static bool parse_strict(const char* p, size_t n, bool at_eof, FastqChunk* c, size_t* used) {
    size_t pos = 0;
    *used = 0;
    auto line = [&](size_t* lo, size_t* ln) -> bool {  // next line without "\n" / "\r\n"; false if no complete line is left
      if (pos >= n) return false;
      const char* nl = (const char*)memchr(p + pos, '\n', n - pos);
      if (!nl && !at_eof) return false;
      size_t len = nl ? (size_t)(nl - (p + pos)) : n - pos;
      *lo = pos;
      pos += len + (nl ? 1 : 0);
      while (len && p[*lo + len - 1] == '\r') len--;
      *ln = len;
      return true;
    };
    for (;;) {
      size_t h, hl, s, sl, pl, pll, q, ql;
      const size_t rec0 = pos;
      if (!line(&h, &hl)) break;
      if (hl == 0) {  // blank lines between records are tolerated by the general reader too
        *used = pos;
        continue;
      }
      if (p[h] != '@') return false;
      if (!line(&s, &sl) || !line(&pl, &pll)) {
        pos = rec0;
        break;
      }
      if (pll == 0 || p[pl] != '+') return false;
      if (!line(&q, &ql)) {
        if (sl == 0 && at_eof && pos >= n) ql = 0;  // "@id\n\n+" at the very end: an empty record without its quality line
        else {
          pos = rec0;
          break;
        }
      }
      if (ql != sl) return false;
      if (sl && (p[s] == '@' || p[s] == '+' || p[s] == '>')) return false;  // bases do not start like that: a wrapped or shifted file
      size_t e = 1;
      while (e < hl && p[h + e] != ' ' && p[h + e] != '\t') e++;
      c->id_buf.insert(c->id_buf.end(), p + h + 1, p + h + e);
      c->id_offs.push_back(c->id_buf.size());
      c->seqs.insert(c->seqs.end(), (const uint8_t*)p + s, (const uint8_t*)p + s + sl);
      c->offs.push_back(c->seqs.size());
      *used = pos;
    }
    return true;  // no violation seen; *used tells how far the complete records reach
  }

 private:
  struct Task {
    std::unique_ptr<FastqChunk> c;
    size_t lo = 0, hi = 0;
  };
  // first record start at or after `from` (from itself if it is one), size_ if there is none
  size_t record_start(size_t from) const {
    size_t p = from;
    if (p > 0 && p < size_ && map_[p - 1] != '\n') {  // move to the next line start
      const char* nl = (const char*)memchr(map_ + p, '\n', size_ - p);
      if (!nl) return size_;
      p = (size_t)(nl - map_) + 1;
    }
    for (int tries = 0; p < size_ && tries < 64; tries++) {
      size_t lo[4], ln[4], q = p;
      int k = 0;
      for (; k < 4 && q < size_; k++) {
        const char* nl = (const char*)memchr(map_ + q, '\n', size_ - q);
        size_t len = nl ? (size_t)(nl - (map_ + q)) : size_ - q;
        lo[k] = q;
        q += len + (nl ? 1 : 0);
        while (len && map_[lo[k] + len - 1] == '\r') len--;
        ln[k] = len;
      }
      if (k == 4 && ln[0] && map_[lo[0]] == '@' && ln[2] && map_[lo[2]] == '+' && ln[1] == ln[3] &&
          !(ln[1] && (map_[lo[1]] == '@' || map_[lo[1]] == '+')))
        return p;
      if (k < 2) return size_;
      p = lo[1];  // try the next line
    }
    return size_;  // no boundary within 64 lines: not a four-line file here; the tail goes to the previous chunk, which will notice
  }
  void cut() {
    size_t lo = 0;
    while (lo < size_) {
      size_t hi = lo + chunk_bytes_ >= size_ ? size_ : record_start(lo + chunk_bytes_);
      std::shared_ptr<Task> t(new Task());
      t->c = ChunkPool::get().take();
      t->c->file_off = lo;
      t->c->strict = true;
      t->c->done = false;
      t->lo = lo;
      t->hi = hi;
      std::unique_lock<std::mutex> l(m_);
      cv_.wait(l, [&] { return stop_ || order_.size() < 2 * pool_.size() + 2; });
      if (stop_) return;
      order_.push_back(t);
      todo_.push_back(t);
      cv_.notify_all();
      lo = hi;
    }
    std::lock_guard<std::mutex> l(m_);
    cut_done_ = true;
    cv_.notify_all();
  }
  void work() {
    for (;;) {
      std::shared_ptr<Task> t;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || !todo_.empty() || cut_done_; });
        if (stop_ || (todo_.empty() && cut_done_)) return;
        if (todo_.empty()) continue;
        t = std::move(todo_.front());
        todo_.pop_front();
      }
      FastqChunk* c = t->c.get();
      const size_t n = t->hi - t->lo;
      c->id_buf.reserve(n / 12);
      c->seqs.reserve(n / 2 + 64);
      size_t used = 0;
      const bool ok = parse_strict(map_ + t->lo, n, true, c, &used) && used == n;  // a chunk ends where the next record starts
      if (!ok) {
        c->strict = false;
        c->id_buf.clear();
        c->id_offs.assign(1, 0);
        c->seqs.clear();
        c->offs.assign(1, 0);
      }
      std::lock_guard<std::mutex> l(m_);
      c->done = true;
      cv_.notify_all();
    }
  }
  std::string path_;
  int fd_ = -1;
  const char* map_ = nullptr;
  size_t size_ = 0, chunk_bytes_ = 0;
  std::thread cutter_;
  std::vector<std::thread> pool_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Task>> order_, todo_;
  bool cut_done_ = false, stop_ = false;
};
