// Entry points shared between translation units of libtsamd (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace tsamd {

// Winner records of the pull-formulated min/max backward (csrc/spmm_bw.hip writes them, the masked
// merge kernel of csrc/spmm.hip and the masked SDDMM read them).  One record per (batch, CSR entry e),
// `win_record_stride(K)` 32-bit words, 16-byte aligned:
//   words [0, W)   W = ceil(K / 32): bit (k % 32) of word k / 32 = (arg_out[b, row(e), k] == e)
//   word  W        row(e)                       (the "column" of e in the transposed product)
//   words W+1, W+2 value[e] as accumulator bits (fp32; fp64 uses both), 1.0 when the matrix has no values
//   word  W+3      when the padding leaves it ((W + 3) % 4 != 0): bit s = (word s != 0), s < 32 -- which 32-feature
//                  segments of the gathered row have a winner in this entry at all
// padded to a multiple of 4 words, so that a 32-byte record (K <= 160) never straddles a 64-byte line and
// ONE line serves the three random accesses an entry needs.
static inline uint32_t win_record_stride(int64_t K) {
  const uint32_t w = (uint32_t)((K + 31) / 32) + 3u;
  return (w + 3u) & ~3u;
}

// out[b, m, k] = sum over the entries i of row m of the pattern `rowptr` (entry i is record perm[i], perm
// may be NULL) whose bit k is set of round_T(value * mat[b, id, k]) -- the merge-path SpMM of csrc/spmm.hip
// with a per-(entry, feature) predicate; E < 2^32.  Workspace as for tsamd_spmm (SUM).
size_t spmm_masked_sum_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E);
int spmm_masked_sum(int dtype, const int64_t *rowptr, bool has_value, const int64_t *perm,
                    const uint32_t *records, const void *mat, void *out, int64_t B, int64_t M, int64_t N,
                    int64_t K, int64_t E, void *workspace, size_t workspace_bytes, hipStream_t stream);

// Winner records from int32 winner ids (csrc/spmm_bw.hip: minmax_winrec_kernel); row = COO row ids [E].
int minmax_winrec_from_ids(int dtype, const int64_t *row, const void *value, const int32_t *arg32, uint32_t *records,
                           int64_t B, int64_t M, int64_t K, int64_t E, hipStream_t stream);

// Verification mode (csrc/spmm_ref_order.hip): the reference CPU kernel's order of operations, bit-identical results.
bool spmm_reference_order_on();
int spmm_reference_order_run(int dtype, int reduce, const int64_t *rowptr, const int64_t *col, const void *value,
                             const void *mat, void *out, void *arg_out, bool arg32, int64_t B, int64_t M, int64_t N,
                             int64_t K, int64_t E, hipStream_t stream);

// grad_mat of the min / max backward by winner lists (csrc/spmm_bw_list.hip): compacted (feature, product) pairs per
// entry instead of one grad_out row per entry.  K <= 1024, ids < 2^32.
bool minmax_bw_lists_supported(int dtype, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E);
size_t minmax_bw_lists_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E);
int minmax_bw_lists(int dtype, const int64_t *row, const int64_t *col, const void *value, const void *grad_out,
                    const int64_t *arg_out, const int64_t *colptr, const int64_t *csr2csc, void *grad_mat, int64_t B,
                    int64_t M, int64_t N, int64_t K, int64_t E, void *workspace, hipStream_t stream);

}  // namespace tsamd
