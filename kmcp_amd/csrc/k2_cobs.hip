// k2_cobs.hip — K2: the COBS query: row = hash % NumSigs, gather rows, AND the h rows, per-column match counts in bit-sliced
// counters, integer threshold, hit emission (kmcp/cmd/util-db-search.go:6611-7742); SPLIT form + k_threshold_long for long
// queries.  The path is bitwise/integer and HBM-bound; there is no MFMA here by design.  wave = 64 lanes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "common.hpp"
#include "csa.hpp"
#include "device_utils.hpp"
#include "kernels.hpp"

namespace kmcpg {

// ------------------------------------------------------------------------------------------------
// K2: the COBS query.
//
// Work unit = (read, slot) with slot = (group of resident blocks that share NumSigs, tile of LPR*16 bytes of its rows).  LPR
// lanes serve one unit, so a wave carries G = 64/LPR units: LPR = 64 for every whole KiB of a row (GTDB-scale: 1872 B), 32, 16, 8 or 4
// for what is left of it or for narrow rows (a lone 312-column block has 39-byte rows, `kmcp index -b 1024` gives 128-byte ones).  Units are numbered slot-major: all
// waves in flight gather from one (group, tile) slice of the index.  Each lane owns 16 bytes = 128 columns of its unit's rows
// and keeps their match counts as NPL bit-sliced planes (vertical counters): the rows of a group of GR = 8 (or 4) are reduced
// with a carry-save adder tree and the carry word rippled into the upper planes, ~4 VALU ops per loaded dword, which keeps the
// kernel memory-bound; after every group the sectors whose columns cannot reach the threshold any more stop
// loading (exact branch and bound).
// Row indices of a chunk of CH k-mers are computed cooperatively (one exact fastmod per (k-mer, block), fastmod.hpp) into a
// per-wave LDS table — hash loads of the whole chunk first, block constants from LDS: no dependent global loads in that loop;
// k-mers past the end of a read map to the all-zero row appended to each block, so the inner loop has no tail code.
// ------------------------------------------------------------------------------------------------
// 16 bytes of a row.  Index rows are read once per launch and L2 cannot hold a slice (cache note): non-temporal
// loads keep them from displacing the hash/offset lines in L2 (no measurable difference either way in the kernel).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_row16(const uint8_t* p, int nt) {
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
  const u32x4 v = nt ? __builtin_nontemporal_load(q) : *q;
  return make_uint4(v.x, v.y, v.z, v.w);
}

// the same through a pointer the compiler knows to be global memory (a row pointer read from a BlockDev record is a generic one to
// it: flat_load, whose completions cannot be counted in order — every wait becomes "all loads").  With global_load the adders of
// one row group wait for that group's loads only while the next group's are in flight.
typedef const u32x4 __attribute__((address_space(1))) * global_row_ptr;
__device__ __forceinline__ uint4 load_row16_global(const uint8_t* p, int nt) {
  global_row_ptr q = (global_row_ptr)(uintptr_t)p;
  const u32x4 v = nt ? __builtin_nontemporal_load(q) : *q;
  return make_uint4(v.x, v.y, v.z, v.w);
}

// byte `byte` of a group's row, bit `bit` (7 = first column of the byte, index.go:1157) -> global column; false for padding
__device__ __forceinline__ bool group_col(const Seg* __restrict__ segs, const BlockDev* __restrict__ bd, uint32_t byte, uint32_t bit, uint32_t* col) {
  const Seg* s = segs + bd->seg0;
  for (uint32_t i = 0; i < bd->nsegs; i++, s++) {
    if (byte < s->byte_end) {
      const uint32_t c = (byte - s->byte_start) * 8u + (7u - bit);
      *col = s->col_base + c;
      return byte >= s->byte_start && c < s->ncols;
    }
  }
  return false;
}

// SPLIT = true is the long-query form: a unit is (long query, slot, chunk of a.split_chk <= 8192 k-mers); its counts are added to a
// per-query u32 array with atomics and thresholded by k_threshold_long, so a whole genome spreads over the chip instead
// of one wave per (query, slot).

// GR = rows gathered between two pruning tests: 8, or 4 where the launch waits for HBM (the host decides, query.cpp).  The
// short form is its own instantiation (4 rows in flight; 97 VGPRs, the 8-row form 91) built for at most 4 waves per SIMD: the kernel
// is bound by the L2->fabric path, not by latency (the 8-row form runs as fast at 2 waves per SIMD as at 5), and of the
// occupancy targets tried for the 4-row form this one is the fastest (GTDB scale: 488 ms; 506-510 ms at 5-6 waves, 510 ms at 3,
// starved at 2: profiles/r02_group_rows.txt).  That cap is for the 1-KB tiles only: with several units per wave (narrow rows, where
// a unit's lanes idle once its sector is dead) a fifth wave is worth 1-2 % (128-byte rows at 55 GB: 63.2 -> 62.3 ms).
struct alignas(16) UnitConst {  // one (read, slot) unit as the index phase sees it
  uint64_t ns, mh;  // NumSigs of the slot's block and its fastmod constant
  uint64_t koff;    // first hash of the read (of this chunk of it: SPLIT)
  uint32_t s16;     // row pitch in 16-byte units
  int32_t n;        // k-mers of the read handled by this unit
};

// The body lives in k2_cobs_body.inc and is included twice: as the kernel k2_cobs (one launch = one lane form; the token stream the kernel
// has always had, so the tuned instantiations compile to the ISA they had) and as the device function k2_body, which k2_cobs_pair calls
// for each of its two lane forms.
template <int LPR, int NPL, bool MULTI, bool SPLIT, int GR = 8>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NPL == 16 && !SPLIT && !MULTI && LPR >= 16) ? 3 : 1, (GR == 4 && LPR == 64) ? 4 : 10))) k2_cobs(const K2Args a) {
#define K2_BID blockIdx.x
#include "k2_cobs_body.inc"
#undef K2_BID
}

template <int LPR, int NPL, bool MULTI, bool SPLIT, int GR>
__device__ __forceinline__ void k2_body(const K2Args& a, const unsigned bid) {
#define K2_BID bid
#include "k2_cobs_body.inc"
#undef K2_BID
}


// Two lane forms of one database in ONE launch (long queries, round 6): the first `nba` workgroups are form A's (the 1-KiB tiles), the rest
// form B's (the remainder of the same rows).  A batch of 16 384 HiFi reads is five rounds of the chip on the 64-lane form and a round and a
// third on the 16-lane one: as two launches, the second waits for the last wave of the first and both end on a part-filled round; in one
// grid B's workgroups take the slots A's last waves free (same-box: K2 3.17 -> 3.04 ms).  The same on two streams worked in a fresh process
// and LOST 0.6 ms per batch in one that had created a dozen streams before (streams share hardware queues): profiles/r06_tail_mode.txt.
template <int LPRA, int LPRB, int NPL, bool MULTI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NPL == 16 && !MULTI) ? 3 : 1, 10))) k2_cobs_pair(const K2Args a, const K2Args b, const unsigned nba) {
  if (blockIdx.x < nba) k2_body<LPRA, NPL, MULTI, false, 8>(a, blockIdx.x);
  else k2_body<LPRB, NPL, MULTI, false, 8>(b, blockIdx.x - nba);
}

constexpr uint64_t K2_MAX_BLOCKS = 1ull << 23;  // x 256 threads = 2^31

template <int LPR, int NPL>
static void launch_k2_t(const K2Args& a, bool multi, hipStream_t st) {
  constexpr int G = 64 / LPR;
  const uint64_t units = (G > 1 && a.slot_major == 2) ? (uint64_t)a.n_reads * (((uint64_t)a.nslots + G - 1) / G) * G : (uint64_t)a.n_reads * a.nslots;
  const uint64_t waves = (units + G - 1) / G;
  const uint64_t blocks = (waves + 3) / 4;
  K2Args b = a;
  for (uint64_t b0 = 0; b0 < blocks; b0 += K2_MAX_BLOCKS) {  // a launch holds fewer than 2^32 threads
    const unsigned nb = (unsigned)std::min<uint64_t>(K2_MAX_BLOCKS, blocks - b0);
    b.unit_base = b0 * 4 * G;
    if (NPL <= 10 && a.group_rows == 4) {
      if (multi)
        hipLaunchKernelGGL((k2_cobs<LPR, NPL <= 10 ? NPL : 8, true, false, 4>), dim3(nb), dim3(256), 0, st, b);
      else
        hipLaunchKernelGGL((k2_cobs<LPR, NPL <= 10 ? NPL : 8, false, false, 4>), dim3(nb), dim3(256), 0, st, b);
    } else if (multi)
      hipLaunchKernelGGL((k2_cobs<LPR, NPL, true, false>), dim3(nb), dim3(256), 0, st, b);
    else
      hipLaunchKernelGGL((k2_cobs<LPR, NPL, false, false>), dim3(nb), dim3(256), 0, st, b);
  }
}

template <int LPR>
static int launch_k2_l(const K2Args& a, int npl, bool multi, hipStream_t st) {
  switch (npl) {
    case 8: launch_k2_t<LPR, 8>(a, multi, st); return 0;
    case 10: launch_k2_t<LPR, 10>(a, multi, st); return 0;
    case 16: launch_k2_t<LPR, 16>(a, multi, st); return 0;
    case 24: launch_k2_t<LPR, 24>(a, multi, st); return 0;
    default: return -1;
  }
}

int launch_k2(const K2Args& a, int lpr, int npl, hipStream_t st) {
  const bool multi = a.num_hashes > 1;
  switch (lpr) {
    case 4: return launch_k2_l<4>(a, npl, multi, st);
    case 8: return launch_k2_l<8>(a, npl, multi, st);
    case 16: return launch_k2_l<16>(a, npl, multi, st);
    case 32: return launch_k2_l<32>(a, npl, multi, st);
    case 64: return launch_k2_l<64>(a, npl, multi, st);
    default: return -1;
  }
}

// blocks of one lane form's launch (as launch_k2_t counts them)
static uint64_t k2_blocks(const K2Args& a, int lpr) {
  const uint64_t G = 64 / (uint64_t)lpr;
  const uint64_t units = (G > 1 && a.slot_major == 2) ? (uint64_t)a.n_reads * (((uint64_t)a.nslots + G - 1) / G) * G : (uint64_t)a.n_reads * a.nslots;
  return ((units + G - 1) / G + 3) / 4;
}

template <int LPRB, int NPL>
static int launch_k2_pair_t(const K2Args& a, const K2Args& b, bool multi, hipStream_t st) {
  const uint64_t nba = k2_blocks(a, 64), nbb = k2_blocks(b, LPRB);
  if (nba == 0 || nbb == 0 || nba + nbb > K2_MAX_BLOCKS) return -1;
  K2Args a0 = a, b0 = b;
  a0.unit_base = b0.unit_base = 0;
  if (multi)
    hipLaunchKernelGGL((k2_cobs_pair<64, LPRB, NPL, true>), dim3((unsigned)(nba + nbb)), dim3(256), 0, st, a0, b0, (unsigned)nba);
  else
    hipLaunchKernelGGL((k2_cobs_pair<64, LPRB, NPL, false>), dim3((unsigned)(nba + nbb)), dim3(256), 0, st, a0, b0, (unsigned)nba);
  return 0;
}

// the 64-lane form (args a) and a narrower one (args b, lane form lprb) of the same batch in one grid; -1: not a pair this file has a kernel for
int launch_k2_pair(const K2Args& a, const K2Args& b, int lprb, int npl, hipStream_t st) {
  const bool multi = a.num_hashes > 1;
  if (npl != 16 || (a.group_rows == 4)) return -1;
  switch (lprb) {
    case 32: return launch_k2_pair_t<32, 16>(a, b, multi, st);
    case 16: return launch_k2_pair_t<16, 16>(a, b, multi, st);
    case 8: return launch_k2_pair_t<8, 16>(a, b, multi, st);
    case 4: return launch_k2_pair_t<4, 16>(a, b, multi, st);
    default: return -1;
  }
}

template <int LPR>
static void launch_k2_split_t(const K2Args& a, bool multi, hipStream_t st) {
  constexpr int G = 64 / LPR;
  const uint64_t units = (uint64_t)a.n_long * a.nslots * a.split_chunks;
  const uint64_t blocks = ((units + G - 1) / G + 3) / 4;
  K2Args b = a;
  for (uint64_t b0 = 0; b0 < blocks; b0 += K2_MAX_BLOCKS) {
    const unsigned nb = (unsigned)std::min<uint64_t>(K2_MAX_BLOCKS, blocks - b0);
    b.unit_base = b0 * 4 * G;
    if (multi)
      hipLaunchKernelGGL((k2_cobs<LPR, 16, true, true>), dim3(nb), dim3(256), 0, st, b);
    else
      hipLaunchKernelGGL((k2_cobs<LPR, 16, false, true>), dim3(nb), dim3(256), 0, st, b);
  }
}

int launch_k2_split(const K2Args& a, int lpr, hipStream_t st) {
  const bool multi = a.num_hashes > 1;
  switch (lpr) {
    case 4: launch_k2_split_t<4>(a, multi, st); return 0;
    case 8: launch_k2_split_t<8>(a, multi, st); return 0;
    case 16: launch_k2_split_t<16>(a, multi, st); return 0;
    case 32: launch_k2_split_t<32>(a, multi, st); return 0;
    case 64: launch_k2_split_t<64>(a, multi, st); return 0;
    default: return -1;
  }
}

// queries with more than split_min k-mers: meta[0] = how many, meta[1] = their largest NumKmers
__global__ void k_list_long(const int32_t* __restrict__ nk, uint32_t n_reads, int32_t split_min, uint32_t* __restrict__ list, uint32_t* __restrict__ meta) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_reads && nk[r] > split_min) {
    list[atomicAdd(&meta[0], 1u)] = r;
    atomicMax(&meta[1], (uint32_t)nk[r]);
  }
}

void launch_list_long(const int32_t* nk, uint32_t n_reads, int32_t split_min, uint32_t* list, uint32_t* meta, hipStream_t st) {
  if (n_reads == 0) return;
  hipLaunchKernelGGL(k_list_long, dim3((n_reads + 255) / 256), dim3(256), 0, st, nk, n_reads, split_min, list, meta);
}

// largest NumKmers of the batch (callers compare it with what the stated read length allows)
__global__ void k_max_nk(const int32_t* __restrict__ nk, uint32_t n_reads, unsigned long long* __restrict__ out) {
  int m = 0;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) m = max(m, nk[r]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, (unsigned long long)m);
}

void launch_max_nk(const int32_t* nk, uint32_t n_reads, unsigned long long* out, hipStream_t st) {
  if (n_reads == 0) return;
  const unsigned blocks = std::min<unsigned>(1024, (n_reads + 255) / 256);
  hipLaunchKernelGGL(k_max_nk, dim3(blocks), dim3(256), 0, st, nk, n_reads, out);
}

// threshold over the accumulated counts of the long queries (same integer rule as the k2_cobs epilogue); one atomic per wave
// and pass: the lanes that hold a hit get consecutive places (ballot + popcount of the lower lanes)
__global__ void k_threshold_long(const K2Args a) {
  const uint64_t total = (uint64_t)a.n_long * a.ncols_total;
  const int lane = threadIdx.x & 63;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < total; i0 += step) {  // wave-uniform trip count
    const uint64_t i = i0 + (uint64_t)lane;
    uint32_t c = 0, r = 0, col = 0;
    bool pass = false;
    if (i < total && (c = a.long_counts[i]) != 0) {
      const uint32_t li = (uint32_t)(i / a.ncols_total);
      col = (uint32_t)(i % a.ncols_total);
      r = a.long_list[li];
      const int n = a.nk[r];
      uint32_t cmin = count_threshold(n, a.min_qcov, a.min_matched);
      // the -f bound as in the plain kernel's epilogue: a query that took the chunked form because KMCPG_SPLIT_MIN is small is still
      // covered by the table (the host's trusted path for compact results relies on "n <= cmin_fpr_n => bound applied", whichever form ran)
      if (a.cmin_fpr && n <= a.cmin_fpr_n) cmin = max(cmin, (uint32_t)a.cmin_fpr[n]);
      pass = c >= cmin;
    }
    const uint64_t m = __ballot(pass);
    if (m == 0) continue;
    const int leader = __ffsll((unsigned long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(a.counter, (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    if (pass) {
      const unsigned long long idx = base + (unsigned long long)__popcll(m & ((1ULL << lane) - 1ULL));
      if (idx < a.hit_cap) {
        kmcpg_hit hit;
        hit.read = r;
        hit.col = col;
        hit.count = c;
        a.hits[idx] = hit;
      }
    }
  }
}

void launch_threshold_long(const K2Args& a, hipStream_t st) {
  const uint64_t total = (uint64_t)a.n_long * a.ncols_total;
  if (total == 0) return;
  unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(k_threshold_long, dim3(blocks), dim3(256), 0, st, a);
}

}  // namespace kmcpg
