#pragma once
#include "common.h"
namespace tsamd {
// Stable sort of COO entries by (row, col) -- the order of row * N + col -- see sort.hip.
//   row_out / col_out (nullable): the sorted ids; perm_out: position of every sorted entry in the input.
//   todo (nullable, device): number of descents of the input known from an earlier probe; 0 at run time = nothing to
//     sort: every kernel returns at once and the finish kernel writes (copy, identity);
//   probe: the build kernel counts descents / adjacent duplicates itself (counts_out[0..1], device), and the passes
//     are decided by that count -- a sort decided on the device without a host sync.  check4: the probe also takes
//     the maxima of the ids (as unsigned numbers) into counts_out[2..3] -- the range check of the constructor.
//   gather_src / gather_dst (nullable): gather_dst[o] = gather_src[perm_out[o]] for arrays of 4- or 8-byte elements,
//     written by the last pass (the values of the entries ride along instead of a gather through perm_out later).
// Inputs are not modified; outputs must not alias them.  E < 2^32, bits(M) + bits(N) <= 64.
// Ranking used by the radix kernels: 0 = one returning LDS atomic per entry (stable when the LDS unit serves the lanes
// of one instruction in ascending order -- checked on the device by a self-test the first time this is called),
// 1 = ballot matching (independent of that order).  sort_set_rank_mode(-1) forgets the decision (the next sort runs the
// self-test again), 0 / 1 force a mode (tests).
int sort_rank_mode(hipStream_t stream);
void sort_set_rank_mode(int mode);
size_t sort_coo_workspace_bytes(int64_t E);
bool sort_coo_supported(int64_t E, int64_t M, int64_t N);
// co (nullable): a COMPACTING sort.  When the bucket path sorts the input, its last kernel writes the distinct pairs
//   to co->row_u / col_u, the start of every run of equal pairs in the sorted order to co->seg_ptr (seg_ptr[nnz] = E)
//   and their number to co->nnz_out, and row_out / col_out / perm_out stay untouched; otherwise the one-sweep passes
//   write row_out / col_out (/ perm_out, nullable) as usual and the CALLER compacts them (it can tell on the device:
//   *sort_fast_flag(workspace, E) != 0 means the compacted outputs are already there).  co->status: nb words of scratch.
//   Fused reduction (round 6): with a riding 4-byte value (gather_bytes == 4) and co->reduce >= 0 the bucket path also
//   REDUCES the values of every run -- sequentially in sorted order, in the accumulator type of
//   segment_reduce_kernel, i.e. the same bits -- into co->value_u (capacity E, entry p = the p-th distinct pair),
//   writes neither seg_ptr nor the sorted values, and sets *co->fused_out = 1 (the caller zeroes it beforehand).
struct SortCoalesce {
  int64_t *row_u, *col_u, *seg_ptr, *nnz_out;
  unsigned long long *status;  // [kSortCoalesceStatusWords]
  void *value_u = nullptr;     // [E] 4-byte elements, or null
  int64_t *fused_out = nullptr;
  int reduce = -1;             // -1: no fused reduction; 0 sum, 1 mean, 2 min, 3 max (TSAMD_SUM .. TSAMD_MAX)
  int is_float = 1;            // 4-byte value type: 1 float32, 0 int32
  bool no_seg = false;         // index only (no value to reduce afterwards): the bucket route does not write seg_ptr
  // the caller's zero-initialised state (the status words above, the state of its compaction kernel) lies in the
  // pre_zero_bytes bytes directly IN FRONT of `workspace`: the sort's first fill covers them too (one fill kernel
  // instead of three, ~4.5 us each); 0 = the sort zeroes co->status itself
  size_t pre_zero_bytes = 0;
};
constexpr int kSortCoalesceStatusWords = 1 << 14;
const unsigned long long *sort_fast_flag(void *workspace, int64_t E);
int sort_coo_onesweep(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N, int64_t *row_out,
                      int64_t *col_out, int64_t *perm_out, const int64_t *todo, bool probe, int64_t *counts_out,
                      void *workspace, hipStream_t stream, const void *gather_src = nullptr,
                      void *gather_dst = nullptr, int gather_bytes = 0, bool check4 = false,
                      const SortCoalesce *co = nullptr);
}  // namespace tsamd
