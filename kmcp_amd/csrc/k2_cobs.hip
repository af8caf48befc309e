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

template <int LPR, int NPL, bool MULTI, bool SPLIT, int GR = 8>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NPL == 16 && !SPLIT && !MULTI && LPR >= 16) ? 3 : 1, (GR == 4 && LPR == 64) ? 4 : 10))) k2_cobs(const K2Args a) {
  constexpr int G = 64 / LPR;
  constexpr int PAIRS = MULTI ? 256 : 1024;
  constexpr int CH = (PAIRS / G) > 64 ? 64 : (PAIRS / G);
  constexpr int NHMAX = MULTI ? 4 : 1;
  static_assert(CH % 8 == 0, "chunk must be a multiple of the CSA group");
  __shared__ uint32_t s_rows[4][NHMAX][G * CH];
  __shared__ UnitConst s_unit[4][G];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, li = lane % LPR;
  const uint64_t per_read = SPLIT ? (uint64_t)a.nslots * a.split_chunks : (uint64_t)a.nslots;
  // slot_major 2 (several units per wave only): the G units of a wave are ONE read on G consecutive slots — the read's hashes are
  // fetched once per wave instead of once per unit (the other units find them in the vector cache) — and the waves are numbered
  // slot-group-major.  For small indexes, where the slices the waves in flight gather from fit the TLB reach anyway.
  const bool by_read = !SPLIT && G > 1 && a.slot_major == 2;
  const uint64_t slot_groups = ((uint64_t)a.nslots + G - 1) / G;
  const uint64_t total_units = by_read ? (uint64_t)a.n_reads * slot_groups * G : (SPLIT ? (uint64_t)a.n_long : (uint64_t)a.n_reads) * per_read;
  const uint64_t u = a.unit_base + ((uint64_t)blockIdx.x * 4 + wave) * G + g;
  bool valid = u < total_units;
  uint32_t r = 0, sidx = 0, li_long = 0;
  int k0 = 0;
  if (valid && by_read) {
    const uint64_t wid = u / G;
    sidx = (uint32_t)((wid / a.n_reads) * G + (uint64_t)g);
    r = (uint32_t)(wid % a.n_reads);
    if (sidx >= a.nslots) {
      valid = false;
      sidx = 0;
    }
  } else if (valid) {
    if (SPLIT) {
      li_long = (uint32_t)(u / per_read);
      const uint32_t rem = (uint32_t)(u % per_read);
      sidx = rem / a.split_chunks;
      k0 = (int)(rem % a.split_chunks) * (int)a.split_chk;
      r = a.long_list[li_long];
    } else {
      if (a.slot_major) {  // all waves in flight work on one (block, tile) slice of the index at a time
        sidx = (uint32_t)(u / a.n_reads);
        r = (uint32_t)(u % a.n_reads);
      } else {
        r = (uint32_t)(u / a.nslots);
        sidx = (uint32_t)(u % a.nslots);
      }
    }
  }
  // (one unit per wave and a tail mode to make room for: what is the same in every lane is told to the compiler, which keeps it in
  // scalar registers — the tail-mode forms have no vector register to spare)
  constexpr bool UNI = LPR == 64 && NPL >= 16 && !SPLIT;
  if constexpr (UNI) {
    r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    sidx = (uint32_t)__builtin_amdgcn_readfirstlane((int)sidx);
  }
  const Slot slot = a.slots[sidx];
  const BlockDev* __restrict__ bd = a.blocks + slot.block;
  int n = valid ? a.nk[r] : 0;
  if constexpr (UNI) n = __builtin_amdgcn_readfirstlane(n);
  if (SPLIT) n = max(0, min(n - k0, (int)a.split_chk));
  else if (a.split_min > 0 && n > a.split_min) n = 0;  // long queries are left to the SPLIT launch
  int nmax = n;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
  if (nmax == 0) return;

  const uint32_t stride = bd->stride;
  const uint32_t boff = (slot.tile * LPR + li) * 16u;
  const bool active = n > 0 && boff < stride;
  // integer threshold (:7468-7470): count >= minMatched && float64(count) > nHashes*queryCov
  uint32_t cmin = count_threshold(n, a.min_qcov, a.min_matched);
  // counts below this fail the host's FPR(n, count) <= max_fpr test (:7474-7478), see fpr_bound in query.cpp
  if (!SPLIT && a.cmin_fpr && n <= a.cmin_fpr_n) cmin = max(cmin, (uint32_t)a.cmin_fpr[n]);
  // Branch and bound: once count + (k-mers still to come) < cmin for every column of a 128-byte sector of the row, nothing
  // in it can become a hit any more and its lanes stop loading.  Unrelated references are dead after ~80 % of a read's
  // k-mers (Bloom density <= fpr), so the tail of the row traffic is never fetched; results are unchanged.
  constexpr int GRP = LPR < 8 ? LPR : 8;  // lanes that share a sector
  bool live = active;
  const uint8_t* __restrict__ base = bd->rows + boff;
  const int nh = MULTI ? a.num_hashes : 1;

  uint32_t pl[4][NPL];
#pragma unroll
  for (int d = 0; d < 4; d++)
#pragma unroll
    for (int p = 0; p < NPL; p++) pl[d][p] = 0;
  uint32_t g_acc = 0;  // 16-byte row loads this wave issued (profiling level 2)
  uint32_t h_acc = 0;  // ... and its 8-byte hash loads (a read's hashes are fetched once per slot: 8 B per 16*LPR B of row)

  // ---- what the index phase needs to know about a unit — its block's modulus, fastmod constant (fastmod.hpp) and row pitch,
  //      the read's k-mer count and where its hashes start — goes to LDS once (32 B per unit).  With the block constants
  //      loaded from global memory inside the loop, every (unit, chunk) paid two dependent round trips through a memory
  //      system that the row gathers keep saturated: on 128-byte rows (8 units per wave) the waves spent as long there as
  //      on their rows.  (LDS rather than registers: five more live VGPRs cost the 8-row forms a wave per SIMD.)
  if (li == 0) {
    UnitConst uc;
    uc.ns = bd->num_sigs;
    uc.mh = bd->magic_hi;
    uc.koff = a.offs[r] + (a.offs2 ? a.offs2[r] : 0) + (uint64_t)k0;
    uc.s16 = stride >> 4;  // rows are addressed in 16-byte units: 32 bits reach 64 GB per block
    uc.n = n;
    s_unit[wave][g] = uc;
  }
  wave_lds_fence();
  constexpr int IT = G * CH / 64;  // (unit, k-mer) pairs of a chunk per lane
  // Tail mode (round 6, below the chunk loop): a long query on a 1-KiB tile whose sectors have died down to a few.
  constexpr bool TAIL = LPR == 64 && NPL >= 16 && !SPLIT;
  [[maybe_unused]] int c_tail = -1;  // >= 0: the first k-mer the tail mode takes

  for (int c0 = 0; c0 < nmax; c0 += CH) {
    // ---- row indices of this chunk: loc = h % NumSigs (:6811), multi-hash h_i = uint32(a + b*i) (util-hash.go:125-142).
    //      First every hash load of the chunk (IT per lane, all in flight together), then the arithmetic.
    uint64_t hv[IT];
    uint32_t has_mask = 0;
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int p = lane + 64 * it;
      const int q = p / CH, j = p % CH;
      const bool has = c0 + j < s_unit[wave][q].n;
      if (a.gathered) h_acc += (uint32_t)__popcll(__ballot(has));  // measurement runs only
      hv[it] = has ? a.hashes[s_unit[wave][q].koff + (uint64_t)(c0 + j)] : 0;
      has_mask |= (has ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int p = lane + 64 * it;
      const uint64_t ns = s_unit[wave][p / CH].ns, mh = s_unit[wave][p / CH].mh;
      const uint32_t s16 = s_unit[wave][p / CH].s16;
      if ((has_mask >> it) & 1u) {
        const uint64_t h = hv[it];
        if (!MULTI) {
          s_rows[wave][0][p] = (uint32_t)fastmod_u64(h, ns, mh) * s16;
        } else {
          const uint32_t ha = (uint32_t)(h >> 32), hb = (uint32_t)h;
          for (int i = 0; i < nh; i++) s_rows[wave][i][p] = (uint32_t)fastmod_u64((uint64_t)(uint32_t)(ha + hb * (uint32_t)i), ns, mh) * s16;
        }
      } else {
        for (int i = 0; i < nh; i++) s_rows[wave][i][p] = (uint32_t)ns * s16;  // the appended all-zero row
      }
    }
    wave_lds_fence();

    const int cnt = min(CH, nmax - c0);
    // NR rows: gather and AND (multi-hash).  The row indices come from LDS first (dead lanes read them too: no harm), then one
    // branch around all loads of the group.
    auto gather = [&](auto nr_tag, int j, uint4* __restrict__ x) {
      constexpr int NR = decltype(nr_tag)::value;
      uint32_t ri[NR];
#pragma unroll
      for (int i = 0; i < NR; i++) ri[i] = s_rows[wave][0][g * CH + j + i];
#pragma unroll
      for (int i = 0; i < NR; i++) x[i] = make_uint4(0, 0, 0, 0);
      if (live) {
#pragma unroll
        for (int i = 0; i < NR; i++) x[i] = load_row16_global(base + ((uint64_t)ri[i] << 4), a.nt_loads);
      }
      if (MULTI) {
        // AND of the h rows (pand.AndUnsafe, :6639-6646), one hash function at a time over all NR k-mers: the NR loads of a hash
        // function are in flight together.  (With the hash loop inside the k-mer loop — a run-time trip count — the compiler waited for
        // every row before it asked for the next: the 3-hash genome search ran at 3.8 TB/s with 24 dependent loads per group, now 5.0.
        // Keeping the loads of TWO hash functions in flight changes nothing more: 4.12-4.17 vs 4.08 ms, same-box A/B.)
        for (int hh = 1; hh < nh; hh++) {
          uint4 w[NR];
#pragma unroll
          for (int i = 0; i < NR; i++) ri[i] = s_rows[wave][hh][g * CH + j + i];
#pragma unroll
          for (int i = 0; i < NR; i++) w[i] = make_uint4(0, 0, 0, 0);
          if (live) {
#pragma unroll
            for (int i = 0; i < NR; i++) w[i] = load_row16_global(base + ((uint64_t)ri[i] << 4), a.nt_loads);
          }
#pragma unroll
          for (int i = 0; i < NR; i++) {
            x[i].x &= w[i].x; x[i].y &= w[i].y; x[i].z &= w[i].z; x[i].w &= w[i].w;
          }
        }
      }
      if (a.gathered) g_acc += (uint32_t)__popcll(__ballot(live)) * NR * (MULTI ? nh : 1);  // measurement runs only
    };
    // the branch-and-bound test after `done` k-mers of the unit's read: true when the whole wave is done with these rows
    auto prune_test = [&](int done) -> bool {
      const int need = (int)cmin - (n - done);  // a column must already hold this many to stay in the race
      bool lane_alive = live;
      if (live && need > 0) {
        uint32_t any = 0;
        if (NPL >= 32 || ((uint32_t)need >> NPL) == 0) {
#pragma unroll
          for (int d = 0; d < 4; d++) {
            uint32_t ge = 0xffffffffu;
#pragma unroll
            for (int p = 0; p < NPL; p++) ge = (((uint32_t)need >> p) & 1u) ? (ge & pl[d][p]) : (ge | pl[d][p]);
            any |= ge;
          }
        }
        lane_alive = any != 0;
      }
      const uint64_t alive = __ballot(lane_alive);
      live = live && ((alive >> (lane & ~(GRP - 1))) & ((1ULL << GRP) - 1ULL)) != 0;
      return alive == 0;
    };
    if constexpr (NPL >= 16) {
      // Long queries, 16 / 24 planes: these kernels run near their issue limits, and most of their VALU work was the carry out of
      // every 8-row group rippling through 13 (21) upper planes.  Here the weight-8 carries of FOUR groups are first reduced among
      // themselves (two of weight 16, then one of weight 32) and a single carry ripples from plane 5 on: 32 rows cost
      // 4 x 7 + 3 carry-save adders and one ripple instead of 4 x 7 and four ripples.  The planes are in canonical form after every
      // 32 rows, which is where the pruning test runs (a sector dies <= 24 rows later than with a test per group: < 2 % of a HiFi
      // sketch); groups past the end of the chunk contribute zero carries.
      static_assert(GR == 8, "deferred carries are written for 8-row groups");
      constexpr int GI_UNROLL = 4;  // (multi-hash: 174 VGPRs = 2 waves per SIMD unrolled, 166 = 3 waves rolled — and the unrolled form is 2 % faster, tools/ab/r04_call17.sh)
      for (int j = 0; j < cnt; j += 32) {
        uint32_t e8[4] = {0, 0, 0, 0}, s16[4] = {0, 0, 0, 0};
        // (single hash function: the loads of group gi + 1 are issued before the adders of group gi run — 16 rows in flight per lane)
        uint4 xq[2][8];
        if (!MULTI) gather(std::integral_constant<int, 8>{}, j, xq[0]);
#pragma unroll GI_UNROLL
        for (int gi = 0; gi < 4; gi++) {
          uint32_t en[4] = {0, 0, 0, 0};
          if (!MULTI && gi < 3 && j + 8 * (gi + 1) < cnt) gather(std::integral_constant<int, 8>{}, j + 8 * (gi + 1), xq[(gi + 1) & 1]);
          if (j + 8 * gi < cnt) {  // wave-uniform
            uint4(&x)[8] = xq[gi & 1];
            if (MULTI) gather(std::integral_constant<int, 8>{}, j + 8 * gi, x);
            en[0] = csa8_low(pl[0], x[0].x, x[1].x, x[2].x, x[3].x, x[4].x, x[5].x, x[6].x, x[7].x);
            en[1] = csa8_low(pl[1], x[0].y, x[1].y, x[2].y, x[3].y, x[4].y, x[5].y, x[6].y, x[7].y);
            en[2] = csa8_low(pl[2], x[0].z, x[1].z, x[2].z, x[3].z, x[4].z, x[5].z, x[6].z, x[7].z);
            en[3] = csa8_low(pl[3], x[0].w, x[1].w, x[2].w, x[3].w, x[4].w, x[5].w, x[6].w, x[7].w);
          }
#pragma unroll
          for (int d = 0; d < 4; d++) carry_step<NPL>(pl[d], gi, en[d], e8[d], s16[d]);
        }
        if (!SPLIT && a.prune && prune_test(min(n, c0 + min(cnt, j + 32)))) break;  // (a chunk can be shorter than 32 rows: multi-hash on 64-byte tiles)
      }
    } else {
      // short queries (8 / 10 planes): the loop these kernels were tuned with, untouched
      // NR rows at a time: gather, AND (multi-hash), add to the counters, then the branch-and-bound test
      auto group = [&](auto nr_tag, int j) -> bool {
        constexpr int NR = decltype(nr_tag)::value;
        uint4 x[NR];
  #pragma unroll
        for (int i = 0; i < NR; i++) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (live) {
            const uint32_t row = s_rows[wave][0][g * CH + j + i];
            v = load_row16(base + ((uint64_t)row << 4), a.nt_loads);
          }
          x[i] = v;
        }
        if (MULTI) {
          // AND of the h rows (pand.AndUnsafe, :6639-6646), one hash function at a time over all NR k-mers: the NR loads of a hash
          // function are in flight together.  (With the hash loop inside the k-mer loop — a run-time trip count — the compiler waited for
          // every row before it asked for the next: the 3-hash genome search ran at 3.8 TB/s with 24 dependent loads per group, now 5.0.
          // Keeping the loads of TWO hash functions in flight changes nothing more: 4.12-4.17 vs 4.08 ms, same-box A/B.)
          for (int hh = 1; hh < nh; hh++) {
            uint4 w[NR];
  #pragma unroll
            for (int i = 0; i < NR; i++) {
              w[i] = make_uint4(0, 0, 0, 0);
              if (live) {
                const uint32_t row2 = s_rows[wave][hh][g * CH + j + i];
                w[i] = load_row16(base + ((uint64_t)row2 << 4), a.nt_loads);
              }
            }
  #pragma unroll
            for (int i = 0; i < NR; i++) {
              x[i].x &= w[i].x; x[i].y &= w[i].y; x[i].z &= w[i].z; x[i].w &= w[i].w;
            }
          }
        }
        if (a.gathered) g_acc += (uint32_t)__popcll(__ballot(live)) * NR * (MULTI ? nh : 1);  // measurement runs only
#ifndef KMCPG_CSA3_LPR8
#define KMCPG_CSA3_LPR8 0
#endif
        constexpr bool B3 = KMCPG_CSA3_LPR8 && LPR == 8;  // A/B knob: v_bitop3 adders on the 128-byte-row form only (profiles/r04_rows128.txt)
        if constexpr (NR == 8) {
          csa8<NPL, B3>(pl[0], x[0].x, x[1].x, x[2].x, x[3].x, x[4].x, x[5].x, x[6].x, x[7].x);
          csa8<NPL, B3>(pl[1], x[0].y, x[1].y, x[2].y, x[3].y, x[4].y, x[5].y, x[6].y, x[7].y);
          csa8<NPL, B3>(pl[2], x[0].z, x[1].z, x[2].z, x[3].z, x[4].z, x[5].z, x[6].z, x[7].z);
          csa8<NPL, B3>(pl[3], x[0].w, x[1].w, x[2].w, x[3].w, x[4].w, x[5].w, x[6].w, x[7].w);
        } else {
          csa4<NPL, B3>(pl[0], x[0].x, x[1].x, x[2].x, x[3].x);
          csa4<NPL, B3>(pl[1], x[0].y, x[1].y, x[2].y, x[3].y);
          csa4<NPL, B3>(pl[2], x[0].z, x[1].z, x[2].z, x[3].z);
          csa4<NPL, B3>(pl[3], x[0].w, x[1].w, x[2].w, x[3].w);
        }
        if (!SPLIT && a.prune && ((((j / NR) + 1) & (a.prune_every - 1)) == 0 || j + NR >= cnt)) {
          const int done = min(n, c0 + j + NR);
          const int need = (int)cmin - (n - done);  // a column must already hold this many to stay in the race
          bool lane_alive = live;
          if (live && need > 0) {
            uint32_t any = 0;
            if (NPL >= 32 || ((uint32_t)need >> NPL) == 0) {
  #pragma unroll
              for (int d = 0; d < 4; d++) {
                uint32_t ge = 0xffffffffu;
#pragma unroll
            for (int p = 0; p < NPL; p++) ge = (((uint32_t)need >> p) & 1u) ? (ge & pl[d][p]) : (ge | pl[d][p]);
            any |= ge;
              }
            }
            lane_alive = any != 0;
          }
          const uint64_t alive = __ballot(lane_alive);
          live = live && ((alive >> (lane & ~(GRP - 1))) & ((1ULL << GRP) - 1ULL)) != 0;
          if (alive == 0) return true;  // the whole wave is done with these rows
        }
        return false;
      };
      // Sectors die at the first test after their last column has fallen behind: with 8 rows between tests that point is
      // overshot by ~3.5 rows on average, with 4 rows by ~1.5 — 2.3 % of a 130-k-mer read's row traffic, for ~15 % more VALU work
      // (the host picks the group size by regime, query.cpp; profiles/r02_group_rows.txt).
      for (int j = 0; j < cnt; j += GR)
        if (group(std::integral_constant<int, GR>{}, j)) break;
    }
    wave_lds_fence();
    if (!SPLIT && a.prune && __ballot(live) == 0) break;
    if constexpr (TAIL) {
      if (a.tail_sectors > 0 && nmax - (c0 + CH) >= a.tail_min) {  // wave-uniform
        const uint64_t lv = __ballot(live);
        int nsec = 0;
#pragma unroll
        for (int sc = 0; sc < 8; sc++) nsec += ((lv >> (8 * sc)) & 0xffull) ? 1 : 0;
        if (nsec <= a.tail_sectors) {
          c_tail = c0 + CH;
          break;
        }
      }
    }
  }
  // ---- Tail mode.  Once all but a few sectors of the tile are dead (unrelated references after ~60-70 % of a sketch; the sectors
  //      that hold the query's relatives never die), the loop above keeps 8 rows x 8-32 lanes in flight per wave — every chunk of 64
  //      k-mers is still eight dependent round trips to memory, for an eighth of the bytes: the genome search spent a fifth of its K2
  //      time moving a twentieth of its traffic, with every wave of the launch in that state at once.  Here the idle lane octets help:
  //      with nsec <= 4 live sectors every live octet gets ngr - 1 = 7 / 3 / 1 helpers, the 64 rows of a chunk are dealt out over the
  //      ngr octets of a sector (8 / 16 / 32 rows each, all in flight together), helpers count into planes of their own, and at the
  //      end the helpers' planes are added to the owner's (bit-sliced ripple add over __shfl).  No pruning test in here: what is
  //      still alive this late holds a hit or a near-hit.  Counts, and so the hits, are the same (integer sums in another order).
  if constexpr (TAIL) {
    if (c_tail >= 0) {
      const uint64_t lv = __ballot(live);
      uint32_t secmask = 0;
#pragma unroll
      for (int sc = 0; sc < 8; sc++) secmask |= ((lv >> (8 * sc)) & 0xffull) ? (1u << sc) : 0u;
      const int nsec = __popc(secmask);
      const int ngr = nsec == 1 ? 8 : (nsec == 2 ? 4 : 2);  // octets per live sector
      const int rpg = CH / ngr;                               // rows of a chunk per octet
      // (what the chunk loop needs of a lane's new role is j0, t_live and t_base; the rest is worked out again for the reduction
      // rather than kept in registers over the loop: the single-hash form has three registers to spare for its third wave)
      int j0;
      bool t_live, own;
      const uint8_t* __restrict__ t_base;
      {
        const int oct = lane >> 3;
        own = (secmask >> oct) & 1u;
        const int below = __popc(secmask & ((1u << oct) - 1u));  // live octets below this one (own: the ordinal of its sector)
        const int dead_rank = oct - below;                       // (helpers) dead octets below this one
        const int ord = own ? below : dead_rank % nsec;          // the live sector this octet works for
        const int grp = own ? 0 : 1 + dead_rank / nsec;          // ... as which of its octets
        uint32_t m = secmask;
#pragma unroll
        for (int i = 0; i < 3; i++) m = i < ord ? (m & (m - 1u)) : m;
        const int src_lane = (__ffs(m) - 1) * 8 + (lane & 7);    // the lane that owns these 16 bytes of the rows
        t_live = grp < ngr && ((lv >> src_lane) & 1ull);
        t_base = bd->rows + (int64_t)boff + (int64_t)(src_lane - lane) * 16;  // (boff = this lane's own 16 bytes)
        j0 = grp * rpg;
      }
      if (!own) {
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
          for (int p = 0; p < NPL; p++) pl[d][p] = 0;
      }
      // (the hashes of the NEXT chunk are asked for before this chunk's rows: a chunk is three round trips to memory here, not four)
      uint64_t h_next = c_tail + lane < s_unit[wave][0].n ? a.hashes[s_unit[wave][0].koff + (uint64_t)(c_tail + lane)] : 0;
      for (int c0 = c_tail; c0 < nmax; c0 += CH) {
        {  // row indices of the chunk (one unit per wave: k-mer = lane), as above
          const bool has = c0 + lane < s_unit[wave][0].n;
          if (a.gathered) h_acc += (uint32_t)__popcll(__ballot(has));
          const uint64_t h = h_next;
          h_next = c0 + CH + lane < s_unit[wave][0].n ? a.hashes[s_unit[wave][0].koff + (uint64_t)(c0 + CH + lane)] : 0;
          const uint64_t ns = s_unit[wave][0].ns, mh = s_unit[wave][0].mh;
          const uint32_t s16 = s_unit[wave][0].s16;
          if (has) {
            if (!MULTI) {
              s_rows[wave][0][lane] = (uint32_t)fastmod_u64(h, ns, mh) * s16;
            } else {
              const uint32_t ha = (uint32_t)(h >> 32), hb = (uint32_t)h;
              for (int i = 0; i < nh; i++) s_rows[wave][i][lane] = (uint32_t)fastmod_u64((uint64_t)(uint32_t)(ha + hb * (uint32_t)i), ns, mh) * s16;
            }
          } else {
            for (int i = 0; i < nh; i++) s_rows[wave][i][lane] = (uint32_t)ns * s16;
          }
        }
        wave_lds_fence();
        const int cnt = min(CH, nmax - c0);
#pragma unroll 1
        for (int gi = 0; 8 * gi < rpg; gi++) {  // (rolled: unrolled, the loads of several groups were hoisted together and cost the kernel a wave per SIMD)
          {
            const int j = j0 + 8 * gi;
            const bool ld = t_live && j < cnt;
            uint32_t ri[8];
            uint4 x[8];
#pragma unroll
            for (int i = 0; i < 8; i++) ri[i] = s_rows[wave][0][(j + i) & (CH - 1)];
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = make_uint4(0, 0, 0, 0);
            if (ld) {
#pragma unroll
              for (int i = 0; i < 8; i++) x[i] = load_row16_global(t_base + ((uint64_t)ri[i] << 4), a.nt_loads);
            }
            if (MULTI) {
              for (int hh = 1; hh < nh; hh++) {
                uint4 w[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ri[i] = s_rows[wave][hh][(j + i) & (CH - 1)];
#pragma unroll
                for (int i = 0; i < 8; i++) w[i] = make_uint4(0, 0, 0, 0);
                if (ld) {
#pragma unroll
                  for (int i = 0; i < 8; i++) w[i] = load_row16_global(t_base + ((uint64_t)ri[i] << 4), a.nt_loads);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                  x[i].x &= w[i].x; x[i].y &= w[i].y; x[i].z &= w[i].z; x[i].w &= w[i].w;
                }
              }
            }
            if (a.gathered) g_acc += (uint32_t)__popcll(__ballot(ld)) * 8u * (uint32_t)(MULTI ? nh : 1);
            // (the weight-8 carry ripples at once: this loop waits for memory, the deferred carries of the loop above would buy nothing)
            ripple<NPL, 3>(pl[0], csa8_low(pl[0], x[0].x, x[1].x, x[2].x, x[3].x, x[4].x, x[5].x, x[6].x, x[7].x));
            ripple<NPL, 3>(pl[1], csa8_low(pl[1], x[0].y, x[1].y, x[2].y, x[3].y, x[4].y, x[5].y, x[6].y, x[7].y));
            ripple<NPL, 3>(pl[2], csa8_low(pl[2], x[0].z, x[1].z, x[2].z, x[3].z, x[4].z, x[5].z, x[6].z, x[7].z));
            ripple<NPL, 3>(pl[3], csa8_low(pl[3], x[0].w, x[1].w, x[2].w, x[3].w, x[4].w, x[5].w, x[6].w, x[7].w));
          }
        }
        wave_lds_fence();
      }
      // the helpers' counts join the owner's: one helper octet per step, plane by plane with a rippling carry
      const int below = __popc(secmask & ((1u << (lane >> 3)) - 1u));
      for (int g2 = 1; g2 < ngr; g2++) {
        const int rank = (g2 - 1) * nsec + below;  // (owners) the dead octet that was their helper g2
        uint32_t dm = ~secmask & 0xffu;
#pragma unroll
        for (int i = 0; i < 6; i++) dm = i < rank ? (dm & (dm - 1u)) : dm;
        const int from = own ? (__ffs(dm) - 1) * 8 + (lane & 7) : lane;
        uint32_t cy[4] = {0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < NPL; p++)
#pragma unroll
          for (int d = 0; d < 4; d++) {
            uint32_t v = (uint32_t)__shfl((int)pl[d][p], from);
            v = own ? v : 0u;
            CSA3(cy[d], pl[d][p], pl[d][p], v, cy[d]);
          }
      }
    }
  }
  // what this wave asked the memory system for: one atomic per wave, spread over K2_GATHER_SLOTS counters a cache line apart
  // (one atomic per row group on a single counter made a GTDB-scale launch take 11 s instead of 0.49 s)
  if (a.gathered && lane == 0 && (g_acc | h_acc)) {
    atomicAdd(a.gathered + (size_t)(blockIdx.x % K2_GATHER_SLOTS) * 16, (unsigned long long)g_acc);
    atomicAdd(a.gathered + (size_t)(blockIdx.x % K2_GATHER_SLOTS) * 16 + 1, (unsigned long long)h_acc);  // same cache line
    if (TAIL && c_tail >= 0) atomicAdd(a.gathered + (size_t)(blockIdx.x % K2_GATHER_SLOTS) * 16 + 2, 1ull);  // waves that finished in tail mode
  }

  if (SPLIT) {
    if (!live) return;
    // partial counts of this chunk -> the query's count array (consecutive lanes hit consecutive words)
    uint32_t* __restrict__ acc = a.long_counts + (uint64_t)li_long * a.ncols_total;
#pragma unroll
    for (int d = 0; d < 4; d++) {
      for (int q = 0; q < 32; q++) {
        uint32_t count = 0;
#pragma unroll
        for (int p = 0; p < NPL; p++) count |= ((pl[d][p] >> q) & 1u) << p;
        uint32_t col;
        if (count && group_col(a.segs, bd, boff + (uint32_t)d * 4u + (uint32_t)(q >> 3), (uint32_t)(q & 7), &col)) atomicAdd(acc + col, count);
      }
    }
    return;
  }
  // ---- hit emission (:7462-7731 scan of the counters + Match append), wave-aggregated: every lane finds its columns with
  //      count >= cmin by a bit-sliced compare, the wave reserves room for all of them with ONE atomic on the global counter
  //      (ballot + prefix sum over the lanes' hit counts) and the lanes write their tuples side by side.  Hits are rare for
  //      distinct references (~1 per read) but come in runs for a family of close relatives (neighbouring columns = one lane).
  //      Padding bits of a row are zero in the resident index (k_repack masks them at load time, the synthetic fill and the
  //      planting helpers never set them), so a count >= cmin >= 1 always belongs to a real column and nothing needs to be
  //      validated before room is reserved.  The byte -> column mapping of a lane's hits keeps the segment of the previous hit
  //      in registers: the dependent global loads of a table walk per hit (58 hits in one lane = 58 round trips to memory in a
  //      row, under a saturated memory system) were what made hit-heavy batches 5 % slower, not the atomics.
  const bool emit = live && !(NPL < 32 && (cmin >> NPL) != 0);  // else: nothing left alive (the host picks NPL with n + 1 < 2^NPL, so cmin always fits: query.cpp)
  uint32_t ge[4];
  uint32_t mine = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t v = emit ? 0xffffffffu : 0u;  // bit-sliced (count >= cmin), LSB to MSB
#pragma unroll
    for (int p = 0; p < NPL; p++) v = ((cmin >> p) & 1u) ? (v & pl[d][p]) : (v | pl[d][p]);
    ge[d] = v;
    mine += (uint32_t)__popc(v);
  }
  if (__ballot(mine != 0) == 0) return;  // wave-uniform: the common case
  uint32_t incl = mine;  // inclusive prefix sum over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  const uint32_t total = __shfl(incl, 63);
  unsigned long long base_idx = 0;
  if (lane == 63) base_idx = atomicAdd(a.counter, (unsigned long long)total);
  base_idx = __shfl(base_idx, 63);
  unsigned long long idx = base_idx + (incl - mine);
  uint32_t sg_lo = 1, sg_hi = 0, sg_col = 0;  // the segment of the previous hit: bytes [sg_lo, sg_hi), column of its first bit
  const Seg* __restrict__ segs = a.segs + bd->seg0;
  const uint32_t nsegs = bd->nsegs;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t w = ge[d];
    while (w) {
      const int q = __ffs(w) - 1;
      w &= w - 1;
      uint32_t count = 0;
#pragma unroll
      for (int p = 0; p < NPL; p++) count |= ((pl[d][p] >> q) & 1u) << p;
      // byte (q>>3) of this dword, bit (q&7): bit 7 = first column of the byte (index.go:1157)
      const uint32_t byte = boff + (uint32_t)d * 4u + (uint32_t)(q >> 3);
      if (byte < sg_lo || byte >= sg_hi) {
        for (uint32_t i = 0; i < nsegs; i++) {
          const Seg sg = segs[i];
          if (byte < sg.byte_end) {
            sg_lo = sg.byte_start;
            sg_hi = sg.byte_end;
            sg_col = sg.col_base;
            break;
          }
        }
      }
      if (idx < a.hit_cap) {
        kmcpg_hit hit;
        hit.read = r;
        hit.col = sg_col + (byte - sg_lo) * 8u + (7u - (uint32_t)(q & 7));
        hit.count = count;
        // A set bit outside every member of the group cannot exist in a resident index (k_repack masks the padding bits of every
        // member's last byte, rows are zero-filled beyond the members).  Should one ever appear, its reserved slot becomes a
        // tombstone that K3 and kmcpg_finalize skip, instead of a hit with a made-up column.
        if (byte < sg_lo || byte >= sg_hi) hit = kmcpg_hit{0xffffffffu, 0xffffffffu, 0u};
        a.hits[idx] = hit;
      }
      idx++;
    }
  }
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
