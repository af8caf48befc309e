// kernels.hpp — host-callable launchers of the HIP kernels in k1_kmers.hip, k1_dedup.hip, k2_cobs.hip, support.hip and sort_huge.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace kmcpg {

bool launch_k1(const K1Args& a, uint32_t max_read_len, hipStream_t st);  // true: adjacent repeats already dropped (scratch[], nk_adj[])
int k1_segment_len();  // positions per workgroup on the whole-genome path
void launch_nk_simple(const int32_t* nk_raw, int32_t* nk_search, uint32_t n, int32_t min_matched, hipStream_t st);
void launch_dedup(DedupArgs a, uint64_t max_n, hipStream_t st);  // queries above HUGE_MIN are left to huge_dedup
// lpr in {4,8,16,32,64}: lanes per row tile; npl in {8,10,16,24}: counter planes.  <0 on bad arguments.
int launch_k2(const K2Args& a, int lpr, int npl, hipStream_t st);
// two lane forms in one grid (long queries: the 64-lane tiles + the remainder's form); -1 when there is no such kernel: launch them one by one
int launch_k2_pair(const K2Args& a64, const K2Args& b, int lprb, int npl, hipStream_t st);
// long queries: chunked counting into a.long_counts, then one thresholding pass
int launch_k2_split(const K2Args& a, int lpr, hipStream_t st);
void launch_list_long(const int32_t* nk, uint32_t n_reads, int32_t split_min, uint32_t* list, uint32_t* meta, hipStream_t st);
void launch_threshold_long(const K2Args& a, hipStream_t st);
// K3: group + filter + order the hit list on the device (k3_finalize.hip); hits_hint = expected number of hits (grid sizing), 0 = hit_cap
void launch_k3(const K3Args& a, uint64_t hits_hint, hipStream_t st);
uint32_t k3_scan_tiles_for(uint32_t n);
void launch_max_nk(const int32_t* nk, uint32_t n_reads, unsigned long long* out, hipStream_t st);
void launch_repack(const uint8_t* src, uint8_t* dst, uint64_t n_rows, uint32_t row_bytes, uint32_t stride, uint32_t byte_off, uint32_t ncols, hipStream_t st);
void launch_gather_rows(const uint8_t* rows, uint32_t stride, uint32_t row_bytes, const uint64_t* idx, uint64_t first, uint64_t n, uint8_t* out,
                        hipStream_t st);
void launch_synth_fill(uint8_t* rows, uint64_t n_rows, uint32_t stride, uint32_t own_stride, uint32_t ncols, uint64_t key, uint32_t p8, hipStream_t st);
void launch_plant(const BlockDev& bd, uint32_t col, int num_hashes, const uint64_t* hashes, uint64_t n, hipStream_t st);

void launch_plant_reads(const BlockDev* blocks, uint32_t nblocks, int num_hashes, const uint64_t* hashes, const uint64_t* offs,
                        const int32_t* nk, const uint32_t* cols, uint32_t n_reads, hipStream_t st);

void launch_build_scatter(uint8_t* sigs, uint64_t num_sigs, uint64_t mh, uint32_t row_bytes, int num_hashes, const uint64_t* hashes,
                          const uint64_t* col_off, uint32_t col0, uint32_t n_cols, uint64_t n, hipStream_t st);

// 2-bit packed bases (4 per byte, base j in bits 2*(j%4) of byte j/4; A=0 C=1 T=2 G=3 = (ascii >> 1) & 3) -> ASCII, then the
// exception runs (every byte that is not A/C/G/T/U in either case) written over them
void launch_unpack2(const uint8_t* packed, uint8_t* out, uint64_t n_bases, const ExcRun* runs, uint32_t n_runs, hipStream_t st);
void launch_apply_exc(const ExcRun* runs, uint32_t n_runs, uint8_t* out, hipStream_t st);  // the runs alone, over text that is there

// experiment only (KMCPG_DEBUG_ROWSORT): every query's hashes re-ordered by h % num_sigs of one block; mode 2 = rotated
void launch_debug_rowsort(uint64_t* hashes, const uint64_t* offs, const int32_t* nk, uint32_t n_reads, uint64_t num_sigs, uint64_t mh, int mode, hipStream_t st);

// queries of up to this many emissions above -u are sorted by one wave (k_dedup_wave), which reads them from hashes[]; the
// long-read sketch kernels keep the raw emissions of such queries for it (k1_kmers.hip)
constexpr int DEDUP_WAVE_CAP = 512;
// queries with more than HUGE_MIN k-mers: device-wide sort + unique (sort_huge.hip)
constexpr uint32_t HUGE_MIN = 65536;
size_t huge_dedup_temp_bytes(uint32_t max_n);
void launch_gather_huge(const uint32_t* list, uint32_t n, const int32_t* nk_raw, const uint64_t* offs, const uint64_t* offs2, uint64_t* out,
                        hipStream_t st);
int huge_dedup(uint64_t* keys, uint64_t* tmp, uint32_t n, int* d_num, void* d_temp, size_t temp_bytes, int32_t* nk_search, uint32_t r, int min_matched,
               hipStream_t st);

}  // namespace kmcpg
