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
  TSAMD_I64 = 5,
  /* SpMM forward only (the reference dispatches AT_DISPATCH_ALL_TYPES_AND2, csrc/cpu/spmm_cpu.cpp:47):
   * sums wrap like the C++ type, mean divides the wrapped sum by the count cast to the type */
  TSAMD_U8 = 6,
  TSAMD_I8 = 7,
  TSAMD_I16 = 8
} tsamd_dtype;

/* Reductions reachable from the registered ops (csrc/spmm.cpp:82,145,195,255;
 * csrc/cpu/reducer.h:6-11 also lists mul/div, which no op can reach). */
typedef enum { TSAMD_SUM = 0, TSAMD_MEAN = 1, TSAMD_MIN = 2, TSAMD_MAX = 3 } tsamd_reduce;

/* Library / runtime identification.  Replaces torch_sparse::cuda_version
 * (csrc/version.cpp:26-41): returns HIP_VERSION the library was built with. */
int64_t tsamd_hip_version(void);
/* Bit 0: the library was built with -DTSAMD_EXPERIMENTS=1 (scripts/variants.py): alternative / rejected kernel
 * variants are compiled in and TSAMD_* environment switches select them.  The shipped build returns 0: it reads no
 * environment variable on any call path and contains none of those variants.  (No reference counterpart:
 * csrc/version.cpp:26-41 only reports the toolkit version.) */
int tsamd_build_flags(void);
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
                                  int64_t N, int64_t K, int64_t E);
int tsamd_spmm(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
               const void *value, const void *mat, void *out, int64_t *arg_out,
               int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
               void *workspace, size_t workspace_bytes, void *stream);
/* Verification mode: set = 1 makes tsamd_spmm / tsamd_spmm_cached / tsamd_spmm_minmax_arg32 (and every torch op on
 * top of them) compute in the ORDER OF OPERATIONS of the reference CPU kernel (csrc/cpu/spmm_cpu.cpp:61-87,
 * csrc/cpu/reducer.h:43-84: a row's entries one after the other, product and sum rounded separately in the element
 * type, mean = sum / (scalar_t)count), one thread per output element: results BIT-IDENTICAL to the reference for every
 * dtype and reduction, at a fraction of the speed.  set = 0 back to the product kernels, anything else: query.
 * Returns the mode in force.  Process-wide; the backward kernels are not affected. */
int tsamd_spmm_reference_order(int set);

/* The same product with the entries of the CSR taken through a permutation: entry e is
 * (col[perm[e]], value[perm[e]]), perm [E] int64.  With (rowptr, col, perm) = (colptr, row, csr2csc)
 * this multiplies by the TRANSPOSE of a matrix straight from its COO/CSR arrays -- the
 * grad_mat = A^T * grad_out of SPMMSum::backward (csrc/spmm.cpp:100-108), where the reference first
 * materialises row.index_select(0, csr2csc) and value.index_select(0, csr2csc).  arg_out (MIN/MAX)
 * holds positions in the permuted order (e, not perm[e]).  Workspace: tsamd_spmm_workspace_bytes. */
int tsamd_spmm_permuted(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                        const void *value, const int64_t *perm, const void *mat, void *out,
                        int64_t *arg_out, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Partial product of ONE COLUMN BLOCK of a matrix, combined into the result of the blocks before it -- the
 * building block of the overlapped all-gather of the row-sharded SpMM (pytorch_sparse_amd/parallel.py, SURVEY.md
 * section 8e: "pre-split each rank's CSR into column blocks by owner, run block p's partial SpMM as soon as
 * shard p lands"; the reference has no distributed code, its slicing primitives are torch_sparse/narrow.py:15-42
 * and cat.py:60-114).  (rowptr, col, value) is the CSR of the block's entries only (col = positions in `mat`,
 * the buffer the block's rows of X landed in).  Never copies `mat` (no relabelling), otherwise the kernels of
 * tsamd_spmm.
 *   accumulate = 0   first block: out / arg_out are overwritten;
 *   accumulate = 1   out / arg_out hold the result of the earlier blocks:
 *       SUM       out += block product (accumulator precision, one rounding per block for f16 / bf16);
 *       MEAN      out = (out + block product) / max(deg_rowptr[m + 1] - deg_rowptr[m], 1): pass SUM for every block
 *                 but the last and MEAN with the WHOLE matrix's rowptr (deg_rowptr, required) for the last one;
 *       MIN/MAX   (out, arg_out) = better of the two, ties to the smaller entry id -- the first occurrence in
 *                 the whole row, as csrc/cpu/reducer.h:63-67 -- whatever order the blocks come in.
 *   arg_map  [E] block entry -> entry id of the whole matrix (NULL: identity), reported in arg_out;
 *   arg_none the whole matrix's "no winner" id (its number of entries): rows without entries so far hold
 *            (0, arg_none), rows whose entries so far all lost against the init value (NaN only) hold
 *            (init, arg_none) -- the same final states as tsamd_spmm.
 * After the last block out / arg_out equal tsamd_spmm on the whole matrix: exactly for MIN / MAX, up to the
 * association of the partial sums for SUM / MEAN.  Workspace: tsamd_spmm_partial_workspace_bytes. */
size_t tsamd_spmm_partial_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N, int64_t K,
                                          int64_t E);
int tsamd_spmm_partial(int dtype, int reduce, const int64_t *rowptr, const int64_t *col, const void *value,
                       const void *mat, void *out, int64_t *arg_out, int64_t B, int64_t M, int64_t N, int64_t K,
                       int64_t E, const int64_t *arg_map, int64_t arg_none, int accumulate,
                       const int64_t *deg_rowptr, void *workspace, size_t workspace_bytes, void *stream);

/* Operand cache.  On Kronecker-like graphs tsamd_spmm copies `mat` to hashed row positions before the
 * gather (channel camping, see below) -- 15 % of a north-star call.  A caller that multiplies by the SAME
 * dense operand again (inference with fixed features, the first layer of every training epoch) can keep
 * that copy: tsamd_spmm_cached takes a caller-owned device buffer of tsamd_spmm_operand_cache_bytes()
 * bytes (0 = this product never copies its operand; 256-byte aligned) that holds the copy, the verdict of
 * the camping probe and a fingerprint of `mat` (64 x 256 sampled 16-byte packets).
 *   cache_valid = 0: fill the buffer (the same work as tsamd_spmm);
 *   cache_valid = 1: the buffer was filled by a call with the same (col, E, N, K, dtype, reduce class): the
 *     fingerprint of `mat` is recomputed (microseconds) and the copy kernel returns at once when it still
 *     matches; when it does not, the copy is redone -- the decision is taken on the device, no sync.
 * The CALLER decides whether `mat` can still be the same matrix (the torch glue keys on storage identity,
 * data pointer, version counter, shape and stream); the fingerprint is the second line of defence for writes
 * that bypass such bookkeeping.  Results are bit-identical to tsamd_spmm.  The workspace
 * (tsamd_spmm_cached_workspace_bytes) no longer contains the copy. */
size_t tsamd_spmm_operand_cache_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N, int64_t K,
                                      int64_t E);
size_t tsamd_spmm_cached_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N, int64_t K,
                                         int64_t E);
int tsamd_spmm_cached(int dtype, int reduce, const int64_t *rowptr, const int64_t *col, const void *value,
                      const void *mat, void *out, int64_t *arg_out, int64_t B, int64_t M, int64_t N,
                      int64_t K, int64_t E, void *workspace, size_t workspace_bytes, void *cache,
                      size_t cache_bytes, int cache_valid, void *stream);

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
 * Relabelled ("channel-camping free") layout, end to end.  tsamd_spmm fixes the camping of
 * Kronecker-like graphs by copying `mat` to hashed row positions on EVERY call (15 % of a
 * north-star step).  A caller that runs many products with the same matrix (every layer / epoch
 * of a GNN) can instead keep its dense matrices in that layout for good:
 *
 *   position of row i of a [n, K] matrix = tsamd_relabel_ids(i, n)   (a bijection on [0, n))
 *
 *   tsamd_relabel_ids       out[j] = position(ids[j]) (ids == NULL: ids[j] = j, i.e. the whole map);
 *                           ids outside [0, n) are passed through (the reference's "E = no winner").
 *   tsamd_spmm_relabelled   same arithmetic as tsamd_spmm -- bit-identical values -- with
 *                           col_h = position(col) (computed once per matrix), mat_h in relabelled
 *                           row order (positions over N) and out_h / arg_out_h written in
 *                           relabelled row order (positions over M): no probe, no copy, and the
 *                           result feeds the next product as it is.  arg_out_h holds the ORIGINAL
 *                           entry ids (E for "no winner").
 * ------------------------------------------------------------------------ */
int tsamd_relabel_ids(const int64_t *ids, int64_t count, int64_t n, int64_t *out, void *stream);
size_t tsamd_spmm_relabelled_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M,
                                             int64_t N, int64_t K, int64_t E);
int tsamd_spmm_relabelled(int dtype, int reduce, const int64_t *rowptr, const int64_t *col_h,
                          const void *value, const void *mat_h, void *out_h, int64_t *arg_out_h,
                          int64_t B, int64_t M, int64_t N, int64_t K, int64_t E, void *workspace,
                          size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ *
 * Row gather of a dense row-major matrix: dst[i, :] = src[idx[i], :], rows of `row_bytes` bytes
 * (any element type).  The pack step in front of the row exchange of the sharded SpMM
 * (pytorch_sparse_amd/parallel.py; no reference counterpart: the reference has no distributed code).
 * idx in [-n_src, n_src); negative ids wrap like torch indexing.
 * ------------------------------------------------------------------------ */
int tsamd_gather_rows(const void *src, const int64_t *idx, void *dst, int64_t n, int64_t n_src,
                      int64_t row_bytes, void *stream);

/* ------------------------------------------------------------------------ *
 * Legacy functional SpMM on a SMALL, UNSORTED COO in one launch.  Replaces the
 * ATen composition of torch_sparse/spmm.py:25-31
 *   (matrix.index_select(-2, col) * value.unsqueeze(-1)  ->  scatter_add over row)
 * where launch latency is all that counts (BASELINE.json configs[0]).
 *
 *   out[r, :] = sum over entries e with row[e] == r of value[e] * mat[col[e], :]
 *
 *   row, col [E] int64 in any order, duplicates add up; value [E] dtype (required);
 *   mat [N, K] dtype; out [M, K] dtype, fully written (rows without entries = 0).
 * No workspace, no host sync, no pre-zeroed output.  fp sums are formed in
 * arrival order (like a device scatter_add); integer sums are exact (wrapping).
 * tsamd_spmm_coo_small_supported(...) == 0: use the sorted route (tsamd_sort_coo*
 * + tsamd_ind2ptr + tsamd_spmm); the entry point itself then returns
 * TSAMD_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------ */
int tsamd_spmm_coo_small_supported(int dtype, int64_t E, int64_t M, int64_t K);
int tsamd_spmm_coo_small(int dtype, const int64_t *row, const int64_t *col,
                         const void *value, const void *mat, void *out,
                         int64_t E, int64_t M, int64_t N, int64_t K, void *stream);

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
 *
 * rowptr [M+1] is the CSR pointer of the forward call (the reference's backward does
 * not need it; here it lets grad_value be accumulated per row in LDS and written with
 * plain stores -- no atomics, no memset).  It may be NULL when grad_value is NULL.
 * grad_mat is accumulated with hardware atomics (it is a scatter into the transposed
 * pattern, which this op does not receive): fp32 / fp64 global_atomic_add; f16 / bf16
 * packed global_atomic_pk_add on the final buffer, i.e. one rounding per addition like the
 * reference's own narrow-type scatter_add_, in a non-deterministic order.  Odd K (or
 * TSAMD_MINMAX_BW_SHADOW=1) accumulates narrow types in an fp32 workspace instead and
 * rounds once; tsamd_spmm_minmax_bw_workspace_bytes() says how much that takes (0
 * otherwise).
 * ------------------------------------------------------------------------ */
size_t tsamd_spmm_minmax_bw_workspace_bytes(int dtype, int64_t B, int64_t N,
                                            int64_t K, int64_t E);
int tsamd_spmm_minmax_bw(int dtype, const int64_t *rowptr, const int64_t *col,
                         const void *value, const void *mat, const void *grad_out,
                         const int64_t *arg_out, void *grad_value, void *grad_mat,
                         int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                         void *workspace, size_t workspace_bytes, void *stream);

/* The same backward as a PULL over the transposed pattern, for callers that hold the CSC arrays
 * (SparseTensor.matmul does: the sum backward uses the same three, csrc/spmm.cpp:84,100-108):
 *   colptr [N+1], csr2csc [E] (CSC position -> CSR entry), row [E] (COO rows of the CSR order).
 * grad_mat is produced column-parallel without atomics:
 *   1. one pass over arg_out writes a winner mask per entry (ceil(K/32) words, bit k = "entry e is the
 *      arg of feature k of its row");
 *   2. the merge-path SpMM kernel runs on (colptr, row, value) through csr2csc with that mask as a
 *      per-(entry, feature) predicate:  grad_mat[b, n, k] = sum_{e in column n, bit k set}
 *      round_T(value[e] * grad_out[b, row[e], k]), accumulated in fp32 (fp64 for f64), rounded once.
 * Deterministic (fixed summation order), every element of grad_mat written exactly once (no memset of
 * the result), the device-scope atomics of tsamd_spmm_minmax_bw are gone.  grad_value is computed as
 * in tsamd_spmm_minmax_bw.  Either output may be NULL.  E must be < 2^32.  Replaces the same ATen
 * composition (csrc/spmm.cpp:204-242, 264-302). */
size_t tsamd_spmm_minmax_bw_csc_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N, int64_t K,
                                                int64_t E);
int tsamd_spmm_minmax_bw_csc(int dtype, const int64_t *rowptr, const int64_t *col, const void *value,
                             const void *mat, const void *grad_out, const int64_t *arg_out,
                             const int64_t *colptr, const int64_t *csr2csc, const int64_t *row,
                             void *grad_value, void *grad_mat, int64_t B, int64_t M, int64_t N, int64_t K,
                             int64_t E, void *workspace, size_t workspace_bytes, void *stream);

/* min / max whose winners stay inside the caller (SparseTensor.matmul(x, 'min' | 'max') returns `out` only and
 * keeps arg_out for its own backward, torch_sparse/matmul.py:60-77, csrc/spmm.cpp:183-203): the same entry ids as
 * arg_out in 32-bit words -- half the bytes in the forward's store and in the backward's read of them (configs[2]:
 * 1.07 -> 0.54 GB each way).  E < 2^31 ("no winner" = E as in tsamd_spmm).
 *   tsamd_spmm_minmax_arg32: tsamd_spmm / tsamd_spmm_cached (cache == NULL: stateless) for reduce = MIN | MAX with
 *     arg_out32 [B, M, K] int32; workspace = tsamd_spmm_workspace_bytes / tsamd_spmm_cached_workspace_bytes.
 *   tsamd_spmm_minmax_bw_csc_arg32: tsamd_spmm_minmax_bw_csc on those ids; TSAMD_ERR_UNSUPPORTED when grad_value is
 *     wanted and the rows are not 16-byte packets (or grad_mat is NULL): widen the ids and take
 *     tsamd_spmm_minmax_bw / _csc instead. */
int tsamd_spmm_minmax_arg32(int dtype, int reduce, const int64_t *rowptr, const int64_t *col, const void *value,
                            const void *mat, void *out, int32_t *arg_out32, int64_t B, int64_t M, int64_t N,
                            int64_t K, int64_t E, void *workspace, size_t workspace_bytes, void *cache,
                            size_t cache_bytes, int cache_valid, void *stream);
int tsamd_spmm_minmax_bw_csc_arg32(int dtype, const int64_t *rowptr, const int64_t *col, const void *value,
                                   const void *mat, const void *grad_out, const int32_t *arg_out32,
                                   const int64_t *colptr, const int64_t *csr2csc, const int64_t *row,
                                   void *grad_value, void *grad_mat, int64_t B, int64_t M, int64_t N, int64_t K,
                                   int64_t E, void *workspace, size_t workspace_bytes, void *stream);

/* min / max whose forward leaves the WINNER RECORDS of the pull backward instead of the winner ids (round 6): for a
 * caller that keeps the winners only for its own backward (SparseTensor.matmul(x, 'max') with x.requires_grad:
 * torch_sparse/matmul.py:60-77 -> csrc/spmm.cpp:183-242) the backward's first step -- re-reading the ids and writing one
 * 32-byte record per entry, 0.30 of the 1.36 ms at configs[2] -- is done by the forward where the winners still sit in
 * registers: at the end of every row that one wave finishes by itself it writes the row's records and does not store
 * the ids at all; rows cut between waves get theirs from the ids right behind the forward (a pass that skips every
 * 64-entry chunk without such a row).  Same records bit for bit as tsamd_spmm_minmax_winrec makes from the ids, hence
 * the same gradients bit for bit.  The forward writes records itself for f16 / bf16 rows of 33..256 features and f32 rows
 * of 65..256 (multiples of 4; 97..128 features -- 32-byte records -- have their own, faster kernel instantiation); any other
 * float shape works too (ids to the workspace, then every record from them).
 *   tsamd_spmm_minmax_records_bytes        size of `records` ([B][E] records of 8..  32-bit words, see csrc/spmm_internal.h)
 *   tsamd_spmm_minmax_records              out [B, M, K] and records; row [E] = COO row ids; E < 2^31
 *   tsamd_spmm_minmax_winrec               records from int32 ids (what tsamd_spmm_minmax_bw_csc_arg32 does first)
 *   tsamd_spmm_minmax_bw_csc_records       tsamd_spmm_minmax_bw_csc on such records: grad_mat (required) and grad_value
 *                                          (optional; TSAMD_ERR_UNSUPPORTED unless the rows are 16-byte packets);
 *                                          has_value: the matrix had values (they are in the records) */
int tsamd_spmm_minmax_records_in_forward(int dtype, int64_t B, int64_t M, int64_t K, int64_t E); /* 1: the merge kernel writes them */
size_t tsamd_spmm_minmax_records_bytes(int64_t B, int64_t K, int64_t E);
size_t tsamd_spmm_minmax_records_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N, int64_t K,
                                                 int64_t E);
int tsamd_spmm_minmax_records(int dtype, int reduce, const int64_t *rowptr, const int64_t *col, const void *value,
                              const void *mat, void *out, const int64_t *row, uint32_t *records, int64_t B, int64_t M,
                              int64_t N, int64_t K, int64_t E, void *workspace, size_t workspace_bytes, void *stream);
int tsamd_spmm_minmax_winrec(int dtype, const int64_t *row, const void *value, const int32_t *arg_out32,
                             uint32_t *records, int64_t B, int64_t M, int64_t K, int64_t E, void *stream);
size_t tsamd_spmm_minmax_bw_csc_records_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N, int64_t K,
                                                        int64_t E);
int tsamd_spmm_minmax_bw_csc_records(int dtype, const int64_t *rowptr, const int64_t *col, int has_value,
                                     const void *mat, const void *grad_out, const uint32_t *records,
                                     const int64_t *colptr, const int64_t *csr2csc, const int64_t *row,
                                     void *grad_value, void *grad_mat, int64_t B, int64_t M, int64_t N, int64_t K,
                                     int64_t E, void *workspace, size_t workspace_bytes, void *stream);

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

/* ------------------------------------------------------------------------
 * Entry-balanced segmented reduction (csrc/segreduce.hip): same result as tsamd_segment_reduce,
 * for segments that may be very long (rows / columns of a power-law matrix:
 * SparseTensor.sum/mean/min/max(dim), torch_sparse/reduce.py:8-67 -> torch_scatter.segment_csr).
 * Work is split by entries, not by segments; deterministic, no atomics; fp32 partial sums of a
 * segment that spans several chunks fold in fp64.  Needs seg_ptr[0] = 0 and seg_ptr[nseg] = E.
 * ------------------------------------------------------------------------ */
size_t tsamd_segment_reduce_balanced_workspace_bytes(int dtype, int64_t E, int64_t D);
int tsamd_segment_reduce_balanced(int dtype, int reduce, const void *value, const int64_t *perm,
                                  const int64_t *seg_ptr, int64_t nseg, int64_t E, int64_t D,
                                  void *out, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ *
 * COO ordering / sorting / coalescing.  Replace the Python + ATen +
 * torch_scatter compositions of SparseStorage (torch_sparse/storage.py):
 *
 *   tsamd_coo_order      the `(idx[1:] < idx[:-1]).any()` / `mask.all()` probes
 *                        (storage.py:150-154, 431-441): counts_out[0] = number of
 *                        positions where key decreases, counts_out[1] = number of
 *                        adjacent duplicates, key = row * N + col.  counts_out is
 *                        a DEVICE int64[2]; reading it is the caller's one sync.
 *   tsamd_sort_coo       sort-on-construct (storage.py:149-162, utils.py:14-21):
 *                        stable radix sort by (row, col) -- the order of row * N + col (one most-significant-digit
 *                        scatter + one in-LDS sort per bucket, or one-sweep LSD passes when a bucket overflows);
 *                        E < 2^32 and bits(M) + bits(N) <= 64, else TSAMD_ERR_UNSUPPORTED; writes the sorted
 *                        row / col (either may be NULL) and the permutation.
 *                        Called with (col, row, E, N, M, NULL, NULL, perm) it
 *                        yields csr2csc (storage.py:407-416).
 *   tsamd_coalesce_index adjacent-duplicate compaction of SORTED (row, col)
 *                        (storage.py:436-447): unique pairs to row_out/col_out
 *                        (capacity E), seg_ptr[j] = first input position of
 *                        unique pair j, seg_ptr[nnz] = E (capacity E + 1),
 *                        *nnz_out (device) = number of unique pairs.
 *   tsamd_segment_reduce torch_scatter.segment_csr (storage.py:448-451):
 *                        out[j, :] = REDUCE_{i in [seg_ptr[j], seg_ptr[j+1])}
 *                        value[perm ? perm[i] : i, :], value is [*, D] row-major.
 *                        MEAN on integer types floors, like torch_scatter.
 * ------------------------------------------------------------------------ */
int tsamd_coo_order(const int64_t *row, const int64_t *col, int64_t E, int64_t N,
                    int64_t *counts_out, void *stream);
/* The same probe plus the range check of the constructor (storage.py:113-129: `row.max() < M`, `col.max() < N`)
 * in one pass: counts_out[0..3] = (#descents, #adjacent duplicates, max row id, max col id) -- one transfer
 * instead of three.  The order is taken lexicographically on (row, col), so the number of columns need not be
 * known yet (the constructor may still have to infer it from max col id). */
int tsamd_coo_check(const int64_t *row, const int64_t *col, int64_t E, int64_t *counts_out, void *stream);
/* tsamd_sort_coo decided ON THE DEVICE (no host sync): probes the order into counts_out[0..1] as
 * tsamd_coo_order does; when the input has no descent the radix passes return at once and the outputs are
 * (row, col, identity), otherwise they are those of tsamd_sort_coo.  Same workspace. */
int tsamd_sort_coo_auto(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                        int64_t *row_out, int64_t *col_out, int64_t *perm_out, int64_t *counts_out,
                        void *workspace, size_t workspace_bytes, void *stream);
/* The same with the order already probed: descents[0] (DEVICE) = #descents of the input, e.g. counts_out[0] of
 * tsamd_coo_check.  Lets a caller enqueue check, sort and the gathers through the permutation back to back
 * and read the check's result only once everything is in flight. */
int tsamd_sort_coo_probed(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                          int64_t *row_out, int64_t *col_out, int64_t *perm_out, const int64_t *descents,
                          void *workspace, size_t workspace_bytes, void *stream);
/* The three sorts above behind one entry, with the entries' values riding along: mode 0 = tsamd_sort_coo, 1 =
 * tsamd_sort_coo_auto (counts [2] written), 2 = tsamd_sort_coo_probed (counts[0] read), 3 = tsamd_coo_check +
 * tsamd_sort_coo_probed in one go: counts [4] = (#descents, #adjacent duplicates, max row id, max col id) written, the
 * check riding in the sort's first pass over (row, col).  value / value_out
 * (both or neither): arrays of E elements of value_bytes = 4 or 8 bytes; value_out[i] = value[perm_out[i]] is
 * written by the last radix pass (the `value.index_select(0, perm)` of torch_sparse/storage.py:160-161 without a
 * second pass over the permutation).  E < 2^32, bits(M) + bits(N) <= 64. */
int tsamd_sort_coo_values(int mode, const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                          int64_t *row_out, int64_t *col_out, int64_t *perm_out, int64_t *counts,
                          const void *value, void *value_out, int64_t value_bytes, void *workspace,
                          size_t workspace_bytes, void *stream);
/* How the radix kernels rank equal digits (what makes the sorts STABLE, i.e. what fixes the permutation the
 * reference gets from its sort in torch_sparse/storage.py:152-162, 407-416 and the order in which coalesce,
 * storage.py:431-466, reduces duplicates): 0 = one returning LDS atomic per entry -- a stable rank as long as the
 * LDS unit serves the lanes of one instruction in ascending order, which a device-side self-test checks the first
 * time a sort runs in the process -- 1 = ballot matching, independent of that order (slower).  set: -1 = query (runs
 * the self-test if no sort has run yet: one synchronising round trip of 4 bytes; call it once before capturing a
 * stream), 0 / 1 = force a mode, 2 = forget the decision and run the self-test again.  Returns the mode in force. */
int tsamd_sort_rank_mode(int set);
size_t tsamd_sort_coo_workspace_bytes(int64_t E);
int tsamd_sort_coo(const int64_t *row, const int64_t *col, int64_t E, int64_t M,
                   int64_t N, int64_t *row_out, int64_t *col_out, int64_t *perm_out,
                   void *workspace, size_t workspace_bytes, void *stream);
/* tsamd_sort_coo_auto + tsamd_coalesce_index in one call -- the functional coalesce / transpose of
 * torch_sparse/coalesce.py:5-25, transpose.py:39-62 (storage.py:149-162 + 431-447 behind them): the distinct
 * (row, col) pairs in row-major order to row_u / col_u (capacity E), the start of every run of equal pairs in the
 * SORTED order to seg_ptr (capacity E + 1, seg_ptr[nnz] = E), counts[0..2] (DEVICE) = (#descents of the input,
 * #adjacent duplicates of the input, #distinct pairs); value_out (nullable, with value: E elements of value_bytes = 4
 * or 8) = the values in sorted order, ready for tsamd_segment_reduce(..., perm = NULL, seg_ptr, ...).  When the bucket
 * path sorts the input the compaction happens while the last kernel writes its output (the sorted ids are never
 * written); otherwise row_tmp / col_tmp (capacity E each) receive the sorted ids and a compaction kernel follows.
 * No host sync; reading counts is the caller's one. */
size_t tsamd_sort_coalesce_workspace_bytes(int64_t E);
int tsamd_sort_coalesce(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N, int64_t *row_tmp,
                        int64_t *col_tmp, int64_t *row_u, int64_t *col_u, int64_t *seg_ptr, int64_t *counts,
                        const void *value, void *value_out, int64_t value_bytes, void *workspace,
                        size_t workspace_bytes, void *stream);
/* tsamd_sort_coalesce with the reduction of the duplicates' values fused into the bucket sort (round 6) -- the whole
 * of torch_sparse/coalesce.py:5-25 (storage.py:431-466: `segment_csr(value, ptr, reduce)`) behind ONE read of the
 * input: `value` is an [E] array of dtype TSAMD_F32 or TSAMD_I32, `reduce` one of TSAMD_SUM .. TSAMD_MAX.  When the
 * bucket path sorts the input, value_u[p] (capacity E) = REDUCE over the p-th run of equal pairs, taken in sorted
 * (= stable input) order in the accumulator type of tsamd_segment_reduce -- the same bits -- counts[3] (DEVICE) = 1,
 * and NEITHER seg_ptr NOR value_out is written.  Otherwise counts[3] = 0 and the call leaves exactly what
 * tsamd_sort_coalesce leaves (seg_ptr, value_out = the values in sorted order): the caller reduces them with
 * tsamd_segment_reduce once it has read counts.  counts has FOUR entries here.  value / value_out / value_u may be
 * NULL together: index only (the common `coalesce(edge_index)` of a graph without edge attributes) -- counts[3] = 0
 * and the bucket route writes NO seg_ptr (8 bytes per entry nobody would read).  Workspace: tsamd_sort_coalesce's. */
int tsamd_sort_coalesce_reduce(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                               int64_t *row_tmp, int64_t *col_tmp, int64_t *row_u, int64_t *col_u, int64_t *seg_ptr,
                               int64_t *counts, int dtype, int reduce, const void *value, void *value_out,
                               void *value_u, void *workspace, size_t workspace_bytes, void *stream);
size_t tsamd_coalesce_workspace_bytes(int64_t E);
int tsamd_coalesce_index(const int64_t *row, const int64_t *col, int64_t E,
                         int64_t *row_out, int64_t *col_out, int64_t *seg_ptr,
                         int64_t *nnz_out, void *workspace, size_t workspace_bytes,
                         void *stream);
int tsamd_segment_reduce(int dtype, int reduce, const void *value, const int64_t *perm,
                         const int64_t *seg_ptr, int64_t nseg, int64_t D, void *out,
                         void *stream);

/* Device-wide exclusive scan of int64 (building block, exported for tests and hosts):
 * out[i] = sum_{j<i} in[j]; in == out allowed; *total (device, nullable) = sum of all. */
size_t tsamd_exclusive_scan_workspace_bytes(int64_t n);
int tsamd_exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, int64_t *total,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ *
 * SpSpMM  C = A * B  (CSR x CSR -> CSR, sum).  Replaces the torch.sparse.mm
 * call behind spspmm_sum (torch_sparse/matmul.py:94-111) and therefore the
 * functional spspmm (torch_sparse/spspmm.py:6-33).  Result contract = what the
 * reference relies on (matmul.py:104-111): every row of C sorted by column,
 * duplicates summed, explicit zeros kept.  fp32 / fp64 only (torch.sparse.mm
 * rejects the other dtypes too).  Count first, write once -- the host allocates between the
 * stages:
 *
 *   1. tsamd_spspmm_plan      prod[M] = products per row of A (sum of the lengths of the B rows
 *        it references); bins[2*M] = ids of the rows with more than 512 products by size class
 *        (medium <= 4096 | large, M slots each; rows of <= 512 products are not listed, their
 *        kernels run over all rows in natural order); stats (DEVICE int64[8]): [2]=#medium
 *        [3]=#large  [4]=products in large rows  [5]=products of the largest row (must be < 2^31: the
 *        per-(row, column range) counters of stage 2 are 32-bit; the caller rejects bigger rows).  colB32 [nnz(B)] = the column ids of B as 32-bit
 *        words (caller-allocated): stages 2 and 4 gather short B rows from all over the array and read
 *        this copy (half the lines per row).
 *        --> host reads stats (sync 1: grid sizes and the workspace of the large rows).
 *   2. tsamd_spspmm_symbolic  nnzC[i] = exact number of entries of row i of C (LDS hash sets for
 *        small / medium rows; large rows: products binned by column range into `workspace`
 *        -- 4 + sizeof(value) bytes per product, which stage 4 reuses -- and counted with LDS
 *        bitmaps).  nnzC has M entries, all written.  `dtype` is the value type of stage 4 (it
 *        fixes the width of a column range).  bin_values != 0: the products of the large rows are
 *        binned together with their values (valA / valB of type `dtype`, either may be NULL = ones),
 *        and stage 4 is then called with values_binned = 1 and does not expand those rows again;
 *        with bin_values = 0 (structure only, or values not known yet) valA / valB are ignored.
 *   3. host: rowptrC = exclusive scan of nnzC over M + 1 entries (tsamd_exclusive_scan_i64, entry
 *        M = 0), reads nnz(C) (sync 2), allocates colC / valC at their FINAL size.
 *   4. tsamd_spspmm_numeric   every row is expanded, sorted by column (registers / LDS), its equal
 *        columns summed in product order (large rows: in atomic order, not bit-reproducible), and
 *        stored at rowptrC[i].  valA / valB may be NULL (all ones); valC may be NULL (structure
 *        only).  `workspace` is the one stage 2 filled.
 * M < 2^31, N < 2^32 - 1.
 * ------------------------------------------------------------------------ */
int tsamd_spspmm_plan(const int64_t *rowptrA, const int64_t *colA, const int64_t *rowptrB,
                      const int64_t *colB, int64_t nnzB, int64_t M, int64_t *prod, int64_t *bins,
                      uint32_t *colB32, int64_t *stats, void *stream);
size_t tsamd_spspmm_workspace_bytes(int dtype, int64_t n_large, int64_t P_large, int64_t N);
int tsamd_spspmm_symbolic(int dtype, const int64_t *rowptrA, const int64_t *colA, const void *valA,
                          const int64_t *rowptrB, const uint32_t *colB32, const void *valB, int bin_values,
                          int64_t M, int64_t N, const int64_t *prod, const int64_t *bins, int64_t n_medium,
                          int64_t n_large, int64_t P_large, int64_t *nnzC, void *workspace,
                          size_t workspace_bytes, void *stream);
int tsamd_spspmm_numeric(int dtype, const int64_t *rowptrA, const int64_t *colA, const void *valA,
                         const int64_t *rowptrB, const uint32_t *colB32, const void *valB, int64_t M,
                         int64_t N, const int64_t *prod, const int64_t *bins, int64_t n_medium,
                         int64_t n_large, int64_t P_large, const int64_t *rowptrC, int64_t *colC,
                         void *valC, int values_binned, void *workspace, size_t workspace_bytes,
                         void *stream);

/* ------------------------------------------------------------------------
 * Sub-matrix extraction (the callers either side of the sharded
 * SpMM path).  Pure index work on a sorted pattern; order preserving.
 *
 * select -- pick K segments (rows of a CSR, or columns of a CSC) by id, duplicates and
 * negative (wrapping) ids allowed.  Replaces torch_sparse/index_select.py:13-68 (rowcount[idx],
 * cumsum, repeat_interleave, torch_scatter.gather_csr, fancy-index gathers).
 *   1. tsamd_select_plan: out_ptr[K+1] = exclusive scan of the picked segment lengths;
 *        info[0] = total entries, info[1] = number of ids outside [-S, S)  (device, 2 int64).
 *   2. host reads info (the one sync: the output size is data dependent), allocates.
 *   3. tsamd_select_fill: for output entry e of output segment i:
 *        seg_out[e] = i, ind_out[e] = ind[src], pos_out[e] = src  with
 *        src = ptr[idx[i]] + (e - out_ptr[i]).  Any of the three outputs may be NULL.
 *
 * filter -- keep the entries of a COO list that satisfy a predicate.  Replaces the
 * boolean-mask compositions of torch_sparse/narrow.py:44-50, masked_select.py:15-90 and
 * diag.py:10-17 (compare -> nonzero -> gathers).
 *   1. tsamd_filter_plan: pos[n+1] = exclusive scan of the keep flags, *count = kept entries.
 *        pred TSAMD_KEEP_COL_RANGE: a <= col < a + b      TSAMD_KEEP_OFF_DIAG: row != col - a
 *        TSAMD_KEEP_MASK: mask[i]   TSAMD_KEEP_MASK_ROW: mask[row[i]]   TSAMD_KEEP_MASK_COL:
 *        mask[col[i]]   (mask = one byte per element, non-zero keeps)   TSAMD_KEEP_COL_MAPPED:
 *        map[col[i]] >= 0  (map = int64 per column id).  With TSAMD_KEEP_MASK and
 *        row = col = NULL, pos is the rank of every set mask byte (new id of a kept row/column).
 *   2. host reads *count, allocates.
 *   3. tsamd_filter_apply: kept entry i goes to slot pos[i]:
 *        row_out = (row_map ? row_map[row] : row) - row_shift, same for col, src_out = i.
 * ------------------------------------------------------------------------ */
enum {
  TSAMD_KEEP_COL_RANGE = 0,
  TSAMD_KEEP_OFF_DIAG = 1,
  TSAMD_KEEP_MASK = 2,
  TSAMD_KEEP_MASK_ROW = 3,
  TSAMD_KEEP_MASK_COL = 4,
  TSAMD_KEEP_COL_MAPPED = 5
};
size_t tsamd_select_workspace_bytes(int64_t K);
int tsamd_select_plan(const int64_t *ptr, int64_t S, const int64_t *idx, int64_t K,
                      int64_t *out_ptr, int64_t *info, void *workspace, size_t workspace_bytes,
                      void *stream);
int tsamd_select_fill(const int64_t *ptr, int64_t S, const int64_t *ind, const int64_t *idx,
                      int64_t K, const int64_t *out_ptr, int64_t total, int64_t *seg_out,
                      int64_t *ind_out, int64_t *pos_out, void *stream);
size_t tsamd_filter_workspace_bytes(int64_t n);
int tsamd_filter_plan(int pred, const int64_t *row, const int64_t *col, const uint8_t *mask,
                      const int64_t *map, int64_t n, int64_t a, int64_t b, int64_t *pos,
                      int64_t *count, void *workspace, size_t workspace_bytes, void *stream);
int tsamd_filter_apply(const int64_t *pos, const int64_t *row, const int64_t *col, int64_t n,
                       const int64_t *row_map, const int64_t *col_map, int64_t row_shift,
                       int64_t col_shift, int64_t *row_out, int64_t *col_out, int64_t *src_out,
                       void *stream);

/* The same filter in two passes over the INPUTS instead of flags + positions per entry (what
 * filter_coo uses; tsamd_filter_plan / _apply stay for callers that need the positions themselves:
 * the mask rank map and the fused set_diag):
 *   tsamd_filter_count  kept entries per 2048-entry tile -> exclusive scan in `workspace`, *count.
 *   host reads *count, allocates.
 *   tsamd_filter_write  re-evaluates the predicate (same arguments!) and writes the kept entries in
 *                       input order at tile offset + rank inside the tile. */
size_t tsamd_filter_tiles_workspace_bytes(int64_t n);
int tsamd_filter_count(int pred, const int64_t *row, const int64_t *col, const uint8_t *mask,
                       const int64_t *map, int64_t n, int64_t a, int64_t b, int64_t *count,
                       void *workspace, size_t workspace_bytes, void *stream);
int tsamd_filter_write(int pred, const int64_t *row, const int64_t *col, const uint8_t *mask,
                       const int64_t *map, int64_t n, int64_t a, int64_t b, const void *workspace,
                       const int64_t *row_map, const int64_t *col_map, int64_t row_shift,
                       int64_t col_shift, int64_t *row_out, int64_t *col_out, int64_t *src_out,
                       void *stream);

/* Column-wise concatenation without a sort (replaces the cat + re-sort of
 * torch_sparse/cat.py:117-165): entry i of one operand goes to slot i + delta[row[i]] of the
 * row-interleaved output, with  delta[r] = out_rowptr[r] + (entries of earlier operands in
 * row r) - rowptr[r];  row_out = row, col_out = col + col_shift, src_out = src_offset + i.
 * Called once per operand on the same output arrays. */
int tsamd_scatter_rows(const int64_t *row, const int64_t *col, int64_t n, const int64_t *delta,
                       int64_t col_shift, int64_t src_offset, int64_t *row_out, int64_t *col_out,
                       int64_t *src_out, void *stream);

/* ------------------------------------------------------------------------
 * Diagonal insertion into a sorted pattern WITHOUT entries on the k-th diagonal.
 * tsamd_num_diag        = length of the k-th diagonal of an M x N matrix.
 * tsamd_non_diag_mask   replaces non_diag_mask_cpu / non_diag_mask_cuda
 *                       (csrc/cpu/diag_cpu.cpp:5-47, csrc/cuda/diag_cuda.cu:9-62):
 *                       mask[E + num_diag] (bytes) is 1 at the slots the existing entries keep
 *                       once the full diagonal is merged in, 0 at the diagonal's slots.
 * tsamd_insert_diag     does the merge itself (replaces the mask + four boolean-mask scatters of
 *                       torch_sparse/diag.py:37-79): row_out/col_out[E + num_diag] is the merged
 *                       sorted pattern, src_out[p] = old position of an existing entry, or
 *                       E + j for the j-th diagonal entry (one gather assembles the values).
 * ------------------------------------------------------------------------ */
int64_t tsamd_num_diag(int64_t M, int64_t N, int64_t k);
int tsamd_non_diag_mask(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                        int64_t k, uint8_t *mask, void *stream);
int tsamd_insert_diag(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                      int64_t k, int64_t *row_out, int64_t *col_out, int64_t *src_out, void *stream);
/* Fused remove + insert (set_diag / fill_diag in one pass over a pattern that may still have
 * entries on the k-th diagonal): pos[E + 1] from tsamd_filter_plan(TSAMD_KEEP_OFF_DIAG, ..., a = k);
 * outputs have pos[E] + tsamd_num_diag(M, N, k) entries; src_out[p] = input position of a kept
 * entry, or E + j for the j-th diagonal entry. */
int tsamd_set_diag_apply(const int64_t *pos, const int64_t *row, const int64_t *col, int64_t E,
                         int64_t M, int64_t N, int64_t k, int64_t *row_out, int64_t *col_out,
                         int64_t *src_out, void *stream);

/* ------------------------------------------------------------------------
 * Mini-batch producers.  CPU-only in the reference except the walk.
 *
 * tsamd_random_walk   replaces random_walk_cpu / random_walk_cuda (csrc/cpu/rw_cpu.cpp:5-45,
 *                     csrc/cuda/rw_cuda.cu:10-53).  rand[n, L] are the uniform floats in [0, 1)
 *                     the reference draws with torch::rand; out[n, L + 1], out[w, 0] = start[w],
 *                     out[w, l + 1] = col[rowptr[cur] + int64(rand[w, l] * deg(cur))].
 *                     A node without neighbours keeps the walk where it is.
 *
 * sample_adj (csrc/cpu/sample_cpu.cpp:10-140) = plan + draw + relabel + a per-row sort:
 *   tsamd_sample_plan   out_ptr[n + 1] = exclusive scan of the per-row sample counts
 *                       (num_neighbors < 0: all; with replacement: num_neighbors if deg > 0;
 *                       else min(deg, num_neighbors)); info[0] = total, info[1] = #bad ids.
 *   tsamd_sample_draw   num_neighbors >= 0: e_id[t] = position of the drawn entry, nbr[t] = its
 *                       column.  Without replacement the draws of a row are distinct (keyed
 *                       bijection of [0, deg) evaluated at 0..k-1, see csrc/sample.hip).
 *                       (num_neighbors < 0 is tsamd_select_fill with ind_out = nbr, pos_out = e_id.)
 *   tsamd_relabel_plan / _apply   replaces the std::unordered_map walk of relabel_cpu /
 *                       relabel_one_hop_cpu / sample_adj_cpu (csrc/cpu/relabel_cpu.cpp:5-155):
 *                       ids in idx[n] keep local id = their position, every other id in nbr[T]
 *                       gets n + (rank of its first occurrence in nbr order).  slot[M] (M = number
 *                       of node ids) and rank[T + 1] are scratch the caller provides; info[0] =
 *                       number of new nodes, info[1] = #ids outside [0, M).  _apply writes
 *                       local[T] and n_id[n + info[0]] (either may be NULL).
 * tsamd_relabel_seed / _extend   the same numbering for MULTI-HOP samplers (neighbor_sample_cpu.cpp:23-133,
 *                       135-400: one std::unordered_map per node type that lives across hops and relations):
 *                       slot[M] is set up ONCE per call and node type (_seed: seeds idx[n] hold -(i + 1), *count
 *                       (device) = n, *err (device) = #ids outside [0, M); a seed listed twice keeps its FIRST position
 *                       with first_wins = 1 -- the map insert of neighbor_sample_cpu.cpp:31, 195 -- else its last); every _extend numbers the ids of
 *                       nbr[T] that are new (n_id[*count + rank] = id, first-occurrence order), turns their slots
 *                       into -(id + 1), writes local[T] (may be NULL) and adds the number of new nodes to *count --
 *                       all on the device: n_id is a buffer of `capacity` ids (the caller knows *count + T as an
 *                       upper bound), *err counts bad ids and appends beyond the capacity.  No host read-back, no
 *                       M-sized fill and no re-seeding per hop.  rank[T + 1] scratch, workspace as _relabel_plan.
 * tsamd_temporal_*    the building blocks of hetero_temporal_neighbor_sample (neighbor_sample_cpu.cpp:222-340: one
 *                     computation tree per root, nodes are (node, root) PAIRS, a neighbour v may be drawn for a node of
 *                     root time t only if node_time[v] <= t).  The constraint is a FLAG per draw (keep[T], int64 0 / 1)
 *                     until the end; one size read-back (info) per relation and hop.
 *   _mark             keep[t] = src_time[nbr[t]] <= f_time[seg[t]] (src_time == NULL: 1); seg[t] = frontier node of draw t.
 *   _redraw           sampling WITH replacement (:268-300): k uniform picks per frontier node among its draws with
 *                     keep = 1 (nbr / e / keep list ALL its neighbours, out_ptr[F + 1] delimits them) -> nbr2, e2, seg2,
 *                     keep2 of F * k entries (keep2 = 0 for a node without a valid neighbour).
 *   _relabel          first-occurrence numbering of the pairs (nbr[t], f_root[seg[t]]) with keep[t] = 1 behind the n pairs
 *                     (old_node, old_root) already numbered (distinct): local[T] (undefined where keep = 0),
 *                     keep_rank[T + 1] / open_rank[T + 1] = exclusive ranks of the kept draws / of the draws that open a new
 *                     pair, info[0] = #kept, info[1] = #new pairs (device).  One stable sort of n + T pairs
 *                     (tsamd_sort_coo); node ids < num_nodes, roots < num_roots.
 *   _emit             after the caller has read info: rows / cols / edges [info[0]] = (local, seg + begin, e) of the kept
 *                     draws, node_out / root_out / time_out [info[1]] = (nbr, f_root[seg], f_time[seg]) of the new pairs.
 * tsamd_subset_assoc  assoc[M] = position of every node in idx, -1 elsewhere (the association
 *                     array of subgraph_cpu, csrc/cpu/saint_cpu.cpp:17-18); *err = #bad ids.
 *                     SAINT sub-graphs are then select + filter(TSAMD_KEEP_COL_MAPPED).
 * ------------------------------------------------------------------------ */
int tsamd_random_walk(const int64_t *rowptr, const int64_t *col, const int64_t *start,
                      const float *rand, int64_t n, int64_t walk_length, int64_t *out,
                      void *stream);
size_t tsamd_sample_workspace_bytes(int64_t n);
int tsamd_sample_plan(const int64_t *rowptr, int64_t M, const int64_t *idx, int64_t n,
                      int64_t num_neighbors, int replace, int64_t *out_ptr, int64_t *info,
                      void *workspace, size_t workspace_bytes, void *stream);
int tsamd_sample_draw(const int64_t *rowptr, const int64_t *col, const int64_t *idx, int64_t n,
                      int64_t num_neighbors, int replace, uint64_t seed, const int64_t *out_ptr,
                      int64_t *e_id, int64_t *nbr, void *stream);
size_t tsamd_relabel_workspace_bytes(int64_t T);
int tsamd_relabel_plan(const int64_t *idx, int64_t n, const int64_t *nbr, int64_t T, int64_t M,
                       int64_t *slot, int64_t *rank, int64_t *info, void *workspace,
                       size_t workspace_bytes, void *stream);
int tsamd_relabel_apply(const int64_t *idx, int64_t n, const int64_t *nbr, int64_t T, int64_t M,
                        const int64_t *slot, const int64_t *rank, int64_t *local, int64_t *n_id,
                        void *stream);
int tsamd_relabel_seed(const int64_t *idx, int64_t n, int64_t M, int64_t *slot, int64_t *count,
                       int64_t *err, int first_wins, void *stream);
int tsamd_relabel_extend(const int64_t *nbr, int64_t T, int64_t M, int64_t *slot, int64_t *rank,
                         int64_t *count, int64_t *local, int64_t *n_id, int64_t capacity, int64_t *err,
                         void *workspace, size_t workspace_bytes, void *stream);
int tsamd_temporal_mark(const int64_t *nbr, const int64_t *seg, int64_t T, const int64_t *src_time,
                        const int64_t *f_time, int64_t *keep, void *stream);
size_t tsamd_temporal_redraw_workspace_bytes(int64_t T);
int tsamd_temporal_redraw(const int64_t *out_ptr, int64_t F, int64_t T, int64_t k, uint64_t seed,
                          const int64_t *nbr, const int64_t *e, const int64_t *keep, int64_t *nbr2, int64_t *e2,
                          int64_t *seg2, int64_t *keep2, void *workspace, size_t workspace_bytes, void *stream);
size_t tsamd_temporal_relabel_workspace_bytes(int64_t n, int64_t T);
int tsamd_temporal_relabel(const int64_t *old_node, const int64_t *old_root, int64_t n, const int64_t *nbr,
                           const int64_t *seg, const int64_t *f_root, const int64_t *keep, int64_t T,
                           int64_t num_nodes, int64_t num_roots, int64_t *local, int64_t *keep_rank,
                           int64_t *open_rank, int64_t *info, void *workspace, size_t workspace_bytes, void *stream);
int tsamd_temporal_emit(const int64_t *nbr, const int64_t *e, const int64_t *seg, const int64_t *f_root,
                        const int64_t *f_time, const int64_t *keep_rank, const int64_t *open_rank,
                        const int64_t *local, int64_t T, int64_t begin, int64_t *rows, int64_t *cols, int64_t *edges,
                        int64_t *node_out, int64_t *root_out, int64_t *time_out, void *stream);
int tsamd_subset_assoc(const int64_t *idx, int64_t n, int64_t M, int64_t *assoc, int64_t *err,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TSAMD_H_ */
