/*
 * tsamd.h -- C-ABI of the MI355X (gfx950) sparse-matmul hot path.
 *
 * This is the drop-in boundary: plain pointers + sizes, no torch types.  Every
 * entry point replaces one L0 function (or one ATen composition) of
 * rusty1s/pytorch_sparse; the reference interface it stands in for is cited
 * next to it as `path:line` relative to the reference tree.
 *
 * Conventions (all entry points):
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - index arrays are int64 (reference contract: storage.py:52,85,92 and
 *     csrc/cpu/spmm_cpu.cpp:39-40);
 *   - dense operands are row-major and contiguous ([B, rows, K]);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     nothing synchronises, nothing allocates, inputs are never written;
 *   - scratch memory is caller-provided: ask `*_workspace_bytes`, pass a
 *     device buffer of at least that size (256-byte aligned);
 *   - the return value is a tsamd_status; TSAMD_ERR_HIP means a HIP runtime
 *     call failed (tsamd_last_hip_error() returns the hipError_t).
 */
#ifndef TSAMD_H_
#define TSAMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  TSAMD_OK = 0,
  TSAMD_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, ...) */
  TSAMD_ERR_UNSUPPORTED = 2, /* dtype / reduce / size not implemented           */
  TSAMD_ERR_HIP = 3,         /* HIP runtime error, see tsamd_last_hip_error()   */
  TSAMD_ERR_WORKSPACE = 4    /* workspace missing or too small                  */
} tsamd_status;

/* Element types of `value` / `mat` / `out`.  Mirrors the dtypes the
 * reference tests sweep (torch_sparse/testing.py:7-14). */
typedef enum {
  TSAMD_F32 = 0,
  TSAMD_F64 = 1,
  TSAMD_F16 = 2,
  TSAMD_BF16 = 3,
  TSAMD_I32 = 4,
  TSAMD_I64 = 5
} tsamd_dtype;

/* Reductions reachable from the registered ops (csrc/spmm.cpp:82,145,195,255;
 * csrc/cpu/reducer.h:6-11 also lists mul/div, which no op can reach). */
typedef enum { TSAMD_SUM = 0, TSAMD_MEAN = 1, TSAMD_MIN = 2, TSAMD_MAX = 3 } tsamd_reduce;

/* Library / runtime identification.  Replaces torch_sparse::cuda_version
 * (csrc/version.cpp:26-41): returns HIP_VERSION the library was built with. */
int64_t tsamd_hip_version(void);
/* Last hipError_t observed by this library on the calling thread. */
int tsamd_last_hip_error(void);
/* Human-readable string for a tsamd_status. */
const char *tsamd_status_string(int status);

/* ------------------------------------------------------------------------ *
 * CSR SpMM forward.   Replaces spmm_cuda / spmm_cpu
 * (csrc/cuda/spmm_cuda.cu:92-155, csrc/cpu/spmm_cpu.cpp:8-101).
 *
 *   out[b, m, :] = REDUCE_{e in [rowptr[m], rowptr[m+1])} value[e] * mat[b, col[e], :]
 *
 *   rowptr  [M+1] int64, col [E] int64 (0 <= col < N, not validated),
 *   value   [E] dtype or NULL (treated as all-ones),
 *   mat     [B, N, K] dtype,   out [B, M, K] dtype,
 *   arg_out [B, M, K] int64, required for MIN/MAX, ignored otherwise:
 *           edge id of the winning entry, first occurrence on ties
 *           (csrc/cpu/reducer.h:63-67); rows without entries get out = 0 and
 *           arg_out = E (csrc/cpu/spmm_cpu.cpp:35, reducer.h:76-82).
 *   MEAN divides by max(row length, 1) (reducer.h:73-74).
 *
 * f16/bf16 SUM/MEAN accumulate in fp32 and round once (the reference CPU
 * path accumulates in the narrow type); MIN/MAX compare the products rounded
 * to the narrow type exactly as the reference does.
 * ------------------------------------------------------------------------ */
size_t tsamd_spmm_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M,
                                  int64_t K, int64_t E);
int tsamd_spmm(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
               const void *value, const void *mat, void *out, int64_t *arg_out,
               int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
               void *workspace, size_t workspace_bytes, void *stream);

/* Measurement aid (bench.py): same as tsamd_spmm, but brackets the three
 * kernels of the launch sequence (merge-path partition, merge, carry fix-up)
 * with hipEvents on `stream`, waits for the last one and writes their
 * durations in milliseconds to kernel_ms_host[0..2] (host memory). */
int tsamd_spmm_profiled(int dtype, int reduce, const int64_t *rowptr,
                        const int64_t *col, const void *value, const void *mat,
                        void *out, int64_t *arg_out, int64_t B, int64_t M, int64_t N,
                        int64_t K, int64_t E, void *workspace, size_t workspace_bytes,
                        void *stream, float *kernel_ms_host);

/* ------------------------------------------------------------------------ *
 * Gradient of SUM/MEAN SpMM w.r.t. the sparse values (an SDDMM over the
 * pattern).  Replaces spmm_value_bw_cuda / spmm_value_bw_cpu
 * (csrc/cuda/spmm_cuda.cu:196-237, csrc/cpu/spmm_cpu.cpp:103-152).
 *
 *   out[e] = sum_b sum_k mat[b, col[e], k] * grad[b, row(e), k]
 *            ( / max(deg(row(e)), 1) for MEAN )
 *
 * The reference takes both the COO `row` and `rowptr`; only `rowptr` is
 * needed here (row(e) is implied by the CSR segment), `row` may be NULL.
 * reduce must be TSAMD_SUM or TSAMD_MEAN.  out is [E] dtype, fully written.
 * ------------------------------------------------------------------------ */
int tsamd_spmm_value_bw(int dtype, int reduce, const int64_t *row,
                        const int64_t *rowptr, const int64_t *col,
                        const void *mat, const void *grad, void *out, int64_t B,
                        int64_t M, int64_t N, int64_t K, int64_t E, void *stream);

/* ------------------------------------------------------------------------ *
 * Backward of MIN/MAX SpMM.  Replaces the ATen composition in
 * SPMMMin/SPMMMax::backward (csrc/spmm.cpp:204-242, 264-302):
 *
 *   for every (b, m, k) with a = arg_out[b,m,k] != E:
 *     grad_value[a]            += mat[b, col[a], k] * grad_out[b,m,k]
 *     grad_mat[b, col[a], k]   += value[a] * grad_out[b,m,k]   (value==NULL: 1)
 *
 * grad_value ([E] dtype) and grad_mat ([B,N,K] dtype) may each be NULL (not
 * requested).  Both are fully defined on return (zero where nothing lands).
 * f16/bf16 accumulate through an fp32 workspace.
 * ------------------------------------------------------------------------ */
size_t tsamd_spmm_minmax_bw_workspace_bytes(int dtype, int64_t B, int64_t N,
                                            int64_t K, int64_t E);
int tsamd_spmm_minmax_bw(int dtype, const int64_t *col, const void *value,
                         const void *mat, const void *grad_out,
                         const int64_t *arg_out, void *grad_value, void *grad_mat,
                         int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ *
 * COO row ids <-> CSR row pointer.  Replace ind2ptr_cuda / ptr2ind_cuda
 * (csrc/cuda/convert_cuda.cu:26-67, csrc/cpu/convert_cpu.cpp:7-57).
 *   ind2ptr: ind [E] sorted ascending, values in [0, M) -> out [M+1];
 *            E == 0 gives all zeros (convert_cpu.cpp:15-16).
 *   ptr2ind: ptr [M+1] -> out [E].
 * ------------------------------------------------------------------------ */
int tsamd_ind2ptr(const int64_t *ind, int64_t M, int64_t E, int64_t *out,
                  void *stream);
int tsamd_ptr2ind(const int64_t *ptr, int64_t M, int64_t E, int64_t *out,
                  void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TSAMD_H_ */
