// common.hpp — structures shared by the host engine and the HIP kernels of libkmcpgpu.so.
#pragma once
#include <stdint.h>

#include "../../include/kmcp_gpu.h"

namespace kmcpg {

// Device-side description of resident index rows (index/serialization.go:66-82 Header, re-laid-out: rows padded to `stride`
// bytes, one all-zero row appended at index num_sigs).  Two arrays of these live on the device:
//  - one entry per resident .uniki block (used by the planting / read-back helpers): `rows` points at the block's first byte
//    inside its group's rows, ncols/col_base are the block's own;
//  - one entry per GROUP (used by K2): resident blocks with the same NumSigs are laid side by side in one row — a k-mer's row
//    index h % NumSigs is the same in all of them, so one wide gather serves them all (the on-disk format is untouched).
//    row_bytes is the sum of the members' NumRowBytes; segs[seg0 .. seg0+nsegs) map a byte of the row back to columns.
struct BlockDev {
  const uint8_t* rows;  // device pointer, (num_sigs + 1) * stride bytes
  uint64_t num_sigs;    // Header.NumSigs: modulus of the row address (util-db-search.go:6811)
  uint64_t magic_hi;    // fastmod_magic(num_sigs): exact h % num_sigs with one 64-bit high multiply (fastmod.hpp; replaces fastdiv, :6611)
  uint32_t stride;      // bytes per row in HBM (multiple of 16)
  uint32_t row_bytes;   // Header.NumRowBytes = (ncols+7)/8 (group: sum over the members)
  uint32_t ncols;
  uint32_t col_base;    // global column id of column 0 (group: of the first member)
  uint32_t seg0, nsegs; // group entries: its members in the segment table
};

// One member block of a group: bytes [byte_start, byte_end) of the group's row hold its columns, MSB of a byte first
// (index.go:1157); the last byte may carry padding bits (always zero).
struct Seg {
  uint32_t byte_start, byte_end;
  uint32_t col_base;  // global column id of the member's column 0
  uint32_t ncols;
};

// 2-bit packed upload of a batch's bases (host.cpp stage / support.hip k_unpack2): a run of bytes that are not A/C/G/T/U in either
// case, restored after unpacking (positions in bases from the start of the batch)
struct ExcRun {
  uint64_t pos;
  uint32_t len;
  uint32_t byte;
};

// One unit of COBS work per read: (local block, tile of LPR*16 bytes of the row).
struct Slot {
  uint32_t block;  // index into the group array
  uint32_t tile;
};

struct K1Args {
  const uint8_t* seqs;
  const uint64_t* offs;
  const uint8_t* seqs2;  // mates or nullptr
  const uint64_t* offs2;
  uint32_t n_reads;
  int32_t k;
  int32_t min_qlen;
  int32_t scaled;
  uint64_t max_hash;
  int32_t mode;          // 0 plain (+scaled), 1 minimizer, 2 syncmer
  uint32_t w_or_s;       // minimizer-w or syncmer-s
  uint64_t* hashes;      // read i writes at hashes[offs[i] + offs2[i] ...]
  uint64_t* scratch;     // same size as hashes: k-mer hashes of window sketches / dedup output
  uint64_t* scratch2;    // same size: s-mer hashes (syncmer mode)
  int32_t* nk_raw;       // k-mers emitted for read i (both mates)
  int32_t* nk1;          // k-mers emitted for mate 1 (for --try-se)
  int32_t* qlen;
  int32_t* seg_cnt;      // whole-genome path: kept hashes per (read, segment), n_reads * segs_max entries; else nullptr
  uint32_t segs_max;
  // window sketches of long reads (k1_kmers_wg<1|2>): emissions without adjacent repeats go to scratch[], their number here
  // (queries above max(dedup_threshold, 512) emissions; the others keep their raw emissions in hashes[]); nullptr = raw only
  int32_t* nk_adj;
  int32_t dedup_threshold;
  int32_t flags;         // experiments (KMCPG_K1_FLAGS): bit 0 = two-level window arg-min, bit 1 = fused adjacent-repeat filter
  uint32_t* seg_list;    // ... the segments k1_seg_roll2 left to the byte kernel, and how many (launch_k1 places both behind seg_cnt[])
  uint32_t* seg_nflag;
  int32_t seg_only_flagged;  // k1_seg_roll as the fallback of k1_seg_roll2: only the segments that kernel marked (seg_cnt == -1)
  // whole genomes that arrived as 2-bit codes (kmcpg_submit_packed / a batch stage() packed): k1_seg_roll2 takes its codes from the packed
  // stream as it is — base j of the batch in bits 2 (j % 4) of byte j / 4 — and only the segments a run of foreign bytes reaches are expanded
  // to text (seqs_w = the buffer `seqs` points at) for the byte kernel.  nullptr: `seqs` holds the text already.
  const uint8_t* codes;
  const ExcRun* exc;
  uint32_t n_exc;
  uint8_t* seqs_w;
  uint32_t* seg_exc;     // per (read, segment): != 0 when a foreign byte lies among the bases the segment's k-mers cover (k_mark_exc)
};

// the packed source of the batch the calling thread is about to enqueue (host.cpp -> run_kmers): codes + runs on the device, and the
// text buffer the expansion goes to when the k-mer kernels of this batch read text
struct PackedSrc {
  const uint8_t* codes = nullptr;
  const ExcRun* exc = nullptr;
  uint32_t n_exc = 0;
  uint8_t* text = nullptr;
  uint64_t n_bases = 0;
};
extern thread_local PackedSrc tl_packed_src;

struct DedupArgs {
  const uint64_t* offs;
  const uint64_t* offs2;
  uint32_t n_reads;
  int32_t dedup_threshold;
  int32_t min_matched;
  int32_t lo, hi;        // this workgroup-class launch sorts queries with lo < m <= hi elements (m: after k_adj_unique)
  int32_t n_lo, n_hi;    // ... among those whose raw count n is in (n_lo, n_hi]
  int32_t pre;           // window-sketch database: k_adj_unique runs first (input of the sort = scratch, m in nk_search)
  int32_t pre_done;      // ... and the k-mer kernel has already done it (k1_kmers_wg<1|2>: fused)
  int32_t key_shift;     // leading zero bits of the largest possible hash (0, or clz(maxHash) for FracMinHash databases): k_dedup_bucket
  uint64_t* hashes;
  uint64_t* scratch;
  const int32_t* nk_raw;
  int32_t* nk_search;    // NumKmers after dedup, 0 if the read is not searched
};

constexpr int K2_GATHER_SLOTS = 256;

struct K2Args {
  const BlockDev* blocks;  // groups
  const Seg* segs;
  const Slot* slots;
  uint32_t nslots;
  uint32_t n_reads;
  const uint64_t* hashes;
  const uint64_t* offs;
  const uint64_t* offs2;
  const int32_t* nk;     // nk_search
  double min_qcov;
  int32_t min_matched;
  int32_t num_hashes;
  int32_t nt_loads;      // non-temporal row loads
  int32_t prune;         // stop loading sectors whose columns can no longer reach the threshold
  int32_t group_rows;    // rows gathered between two pruning tests: 4 or 8
  int32_t prune_every;   // the pruning test runs after every prune_every-th row group (1, 2, 4 or 8) and at the end of a chunk of 64 rows
  int32_t split_min;     // >0: queries with more k-mers are handled by the SPLIT launch
  int32_t slot_major;    // unit order: 1 = slot-major (unit u -> slot u / n_reads, read u % n_reads), 0 = read-major
  int32_t tail_sectors;  // long queries on 1-KiB tiles: with at most this many live sectors (<= 4; 0 = never) and
  int32_t tail_min;      // ... at least this many k-mers to go a wave finishes in tail mode (k2_cobs.hip)
  // long-query (SPLIT) form
  const uint32_t* long_list;  // indices of the long queries
  uint32_t n_long;
  uint32_t split_chk;         // k-mers per chunk (<= 8192: 16 counter planes)
  uint32_t split_chunks;      // chunks per long query (from the largest one)
  uint32_t* long_counts;      // [n_long][ncols_total] match counts
  uint32_t ncols_total;
  uint64_t unit_base;     // first work unit of this launch (a batch may take several launches)
  kmcpg_hit* hits;
  uint64_t hit_cap;
  unsigned long long* counter;
  unsigned long long* gathered;  // optional (profiling level 2): 16-byte row loads issued, K2_GATHER_SLOTS counters 128 B apart
  const uint16_t* cmin_fpr;      // optional: [n] = smallest count whose FPR(n, count) passes -f, n <= cmin_fpr_n (query.cpp fpr_bound)
  int32_t cmin_fpr_n;
};

// K3 (k3_finalize.hip): hit list -> per-read segments of (column, count), filtered by -T, ordered as the reference orders a
// query's matches.  Segments of more than K3_WG_CAP matches are grouped but left unordered (the host sorts those).
constexpr int K3_WAVE_CAP = 512, K3_WG_CAP = 4096;
struct K3Args {
  const kmcpg_hit* hits;
  const unsigned long long* n_hits;  // device word: hits produced (may exceed hit_cap: then only hit_cap are there)
  uint64_t hit_cap;
  const int32_t* nk;        // NumKmers per read
  uint32_t n_reads;
  uint32_t n_cols;
  const uint64_t* col_size; // Header.Sizes per global column
  double min_tcov;          // -T
  int32_t sort_mode;        // 0 qcov, 1 tcov, 2 jacc (-s), 3 column order (-S)
  uint32_t* cnt;            // [n_reads + 1], zero on entry and on exit
  uint64_t* offs;           // [n_reads + 1] out: CSR offsets of the reads' segments
  uint64_t* sums;           // scan scratch, one word per 4096 reads
  kmcpg_pair* pairs;        // out: offs[n_reads] pairs
  uint32_t* bad;            // hits naming a read / column that does not exist (an internal error: the caller reports it)
};

}  // namespace kmcpg
