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
int sort_coo_onesweep(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N, int64_t *row_out,
                      int64_t *col_out, int64_t *perm_out, const int64_t *todo, bool probe, int64_t *counts_out,
                      void *workspace, hipStream_t stream, const void *gather_src = nullptr,
                      void *gather_dst = nullptr, int gather_bytes = 0, bool check4 = false);
}  // namespace tsamd
