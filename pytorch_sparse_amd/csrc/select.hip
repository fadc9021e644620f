// Sub-matrix extraction on gfx950: segment (row / column) selection and predicate filtering of
// a sorted COO/CSR pattern.  SURVEY.md section 8f rank 3 -- the callers either side of the
// sharded SpMM path.  Replaces the ATen/torch_scatter compositions of the reference:
//   * index_select(dim 0/1)    torch_sparse/index_select.py:13-68   (rowcount[idx], cumsum,
//                              repeat_interleave, gather_csr, three fancy-index gathers)
//   * masked_select(dim 0/1)   torch_sparse/masked_select.py:15-63  (mask[row], boolean-index
//                              compaction = nonzero + gathers, repeat_interleave)
//   * narrow(dim 1)            torch_sparse/narrow.py:44-50         (two compares, boolean index)
//   * remove_diag              torch_sparse/diag.py:10-17
//   * masked_select_nnz        torch_sparse/masked_select.py:76-90
// with two fused primitives, each a fixed three-launch sequence and one data-dependent size:
//   select  : counts of the picked segments -> device scan -> one streaming fill that writes
//             the new segment ids, the gathered indices and the source positions (for values)
//   filter  : predicate flags -> device scan -> one compaction that writes shifted / remapped
//             (row, col) and the source positions
// Pure index work, HBM-bound; order preserving, so sorted input gives sorted output.
#include "common.h"
#include "expand.h"
#include "scan.h"

namespace tsamd {
namespace {

// cnt[i] = length of segment idx[i] (negative ids wrap like torch indexing); ids outside
// [-S, S) are counted in *err and contribute an empty segment.
__global__ void select_count_kernel(const int64_t *__restrict__ ptr, int64_t S,
                                    const int64_t *__restrict__ idx, int64_t K,
                                    int64_t *__restrict__ cnt, unsigned long long *err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  int64_t j = idx[i];
  if (j < 0) j += S;
  if (j < 0 || j >= S) {
    atomicAdd(err, 1ull);
    cnt[i] = 0;
    return;
  }
  cnt[i] = ptr[j + 1] - ptr[j];
}

// Balanced by output entries (expand.h): a tile of 2048 consecutive output entries per workgroup;
// the output offsets and source starts of the segments that intersect it go to LDS, every thread
// finds the segment of its entries by an LDS binary search; coalesced stores.
__global__ __launch_bounds__(256) void select_fill_kernel(
    const int64_t *__restrict__ ptr, int64_t S, const int64_t *__restrict__ ind,
    const int64_t *__restrict__ idx, int64_t K, const int64_t *__restrict__ out_ptr, int64_t total,
    int64_t *__restrict__ seg_out, int64_t *__restrict__ ind_out, int64_t *__restrict__ pos_out) {
  __shared__ int64_t so[kExpandTile];
  __shared__ int64_t ss[kExpandTile];
  __shared__ int64_t span[2];
  const int64_t e0 = (int64_t)blockIdx.x * kExpandTile;
  const int64_t e1 = e0 + kExpandTile < total ? e0 + kExpandTile : total;
  int64_t lo, hi;
  tile_span(out_ptr, K, e0, e1, span, &lo, &hi);
  const int64_t n = hi - lo + 1;
  const bool staged = n <= kExpandTile;
  if (staged) {
    for (int i = threadIdx.x; i < (int)n; i += blockDim.x) {
      so[i] = out_ptr[lo + i];
      int64_t j = idx[lo + i];
      if (j < 0) j += S;
      ss[i] = (j >= 0 && j < S) ? ptr[j] : 0;
    }
    __syncthreads();
  }
  for (int64_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
    int64_t seg, src;
    if (staged) {
      const int i = segment_of_lds(so, (int)n, e);
      seg = lo + i;
      src = ss[i] + (e - so[i]);
    } else {  // long runs of empty picks: search the global offsets in place
      seg = lo + segment_of(out_ptr + lo, n, e);
      int64_t j = idx[seg];
      if (j < 0) j += S;
      src = ((j >= 0 && j < S) ? ptr[j] : 0) + (e - out_ptr[seg]);
    }
    if (seg_out) seg_out[e] = seg;
    if (ind_out) ind_out[e] = ind[src];
    if (pos_out) pos_out[e] = src;
  }
}

__device__ inline bool keep_entry(int pred, const int64_t *row, const int64_t *col,
                                  const uint8_t *mask, const int64_t *map, int64_t i, int64_t a,
                                  int64_t b) {
  switch (pred) {
    case TSAMD_KEEP_COL_RANGE: {
      const int64_t c = col[i];
      return c >= a && c < a + b;
    }
    case TSAMD_KEEP_OFF_DIAG: return row[i] != col[i] - a;
    case TSAMD_KEEP_MASK: return mask[i] != 0;
    case TSAMD_KEEP_MASK_ROW: return mask[row[i]] != 0;
    case TSAMD_KEEP_MASK_COL: return mask[col[i]] != 0;
    case TSAMD_KEEP_COL_MAPPED: return map[col[i]] >= 0;
    default: return false;
  }
}

// The same predicate with `pred` known at compile time: inside an unrolled loop the run-time switch above puts every
// entry's loads into their own basic block -- load, wait, compare, next entry: one round trip PER ENTRY and thread
// (scripts/scan_serial_loads.py: 64 of the count kernel's 80 loads and 96 of the write kernel's 112 were waited for
// alone).  The tile kernels dispatch on `pred` ONCE and run a straight-line body whose loads are issued together.
template <int PRED>
__device__ __forceinline__ bool keep_entry_t(const int64_t *row, const int64_t *col, const uint8_t *mask,
                                             const int64_t *map, int64_t i, int64_t a, int64_t b) {
  if constexpr (PRED == TSAMD_KEEP_COL_RANGE) {
    const int64_t c = col[i];
    return c >= a && c < a + b;
  } else if constexpr (PRED == TSAMD_KEEP_OFF_DIAG) {
    return row[i] != col[i] - a;
  } else if constexpr (PRED == TSAMD_KEEP_MASK) {
    return mask[i] != 0;
  } else if constexpr (PRED == TSAMD_KEEP_MASK_ROW) {
    return mask[row[i]] != 0;
  } else if constexpr (PRED == TSAMD_KEEP_MASK_COL) {
    return mask[col[i]] != 0;
  } else if constexpr (PRED == TSAMD_KEEP_COL_MAPPED) {
    return map[col[i]] >= 0;
  } else {
    return false;
  }
}
#define TSAMD_FILTER_DISPATCH(pred, BODY)                                 \
  switch (pred) {                                                         \
    case TSAMD_KEEP_COL_RANGE: BODY(TSAMD_KEEP_COL_RANGE); break;         \
    case TSAMD_KEEP_OFF_DIAG: BODY(TSAMD_KEEP_OFF_DIAG); break;           \
    case TSAMD_KEEP_MASK: BODY(TSAMD_KEEP_MASK); break;                   \
    case TSAMD_KEEP_MASK_ROW: BODY(TSAMD_KEEP_MASK_ROW); break;           \
    case TSAMD_KEEP_MASK_COL: BODY(TSAMD_KEEP_MASK_COL); break;           \
    case TSAMD_KEEP_COL_MAPPED: BODY(TSAMD_KEEP_COL_MAPPED); break;       \
    default: break;                                                       \
  }

__global__ void filter_flags_kernel(int pred, const int64_t *__restrict__ row,
                                    const int64_t *__restrict__ col,
                                    const uint8_t *__restrict__ mask,
                                    const int64_t *__restrict__ map, int64_t n, int64_t a,
                                    int64_t b, int64_t *__restrict__ pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[i] = keep_entry(pred, row, col, mask, map, i, a, b) ? 1 : 0;
}

// pos = exclusive scan of the keep flags with pos[n] = count; entry i is kept iff pos[i+1] > pos[i]
__global__ void filter_apply_kernel(const int64_t *__restrict__ pos, const int64_t *__restrict__ row,
                                    const int64_t *__restrict__ col, int64_t n,
                                    const int64_t *__restrict__ row_map,
                                    const int64_t *__restrict__ col_map, int64_t row_shift,
                                    int64_t col_shift, int64_t *__restrict__ row_out,
                                    int64_t *__restrict__ col_out, int64_t *__restrict__ src_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = pos[i];
  if (pos[i + 1] == p) return;
  if (row_out) {
    const int64_t r = row[i];
    row_out[p] = (row_map ? row_map[r] : r) - row_shift;
  }
  if (col_out) {
    const int64_t c = col[i];
    col_out[p] = (col_map ? col_map[c] : c) - col_shift;
  }
  if (src_out) src_out[p] = i;
}

// Column-wise concatenation (cat_second): entry e of one operand lands at e + delta[row[e]] of the
// row-interleaved output, delta[r] = out_rowptr[r] + (entries of earlier operands in row r) - rowptr[r].
__global__ void scatter_rows_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                    int64_t n, const int64_t *__restrict__ delta, int64_t col_shift,
                                    int64_t src_offset, int64_t *__restrict__ row_out,
                                    int64_t *__restrict__ col_out, int64_t *__restrict__ src_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t r = row[i];
  const int64_t p = i + delta[r];
  if (row_out) row_out[p] = r;
  if (col_out) col_out[p] = col[i] + col_shift;
  if (src_out) src_out[p] = src_offset + i;
}

// ---- diagonal insertion (reference: csrc/cpu/diag_cpu.cpp:5-47, torch_sparse/diag.py:37-79) ----
// Number of entries of the k-th diagonal {(d, d + k) : start <= d < start + num_diag} that precede
// the off-diagonal entry (r, c) in row-major order.
__device__ inline int64_t diag_before(int64_t r, int64_t c, int64_t k, int64_t start,
                                      int64_t num_diag) {
  int64_t below = r - start;
  below = below < 0 ? 0 : (below > num_diag ? num_diag : below);
  const bool own = r >= start && r < start + num_diag && r + k < c;
  return below + (own ? 1 : 0);
}

// mask[E + num_diag]: true at the slots the existing (off-diagonal, sorted) entries move to
__global__ void non_diag_mask_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                     int64_t E, int64_t k, int64_t start, int64_t num_diag,
                                     uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const int64_t r = row[i], c = col[i];
  // an entry ON the diagonal keeps no slot (the reference leaves its mask byte unset as well)
  if (r >= start && r < start + num_diag && r + k == c) return;
  mask[i + diag_before(r, c, k, start, num_diag)] = 1;
}

// fused set_diag: existing entries move to their slots, src = their old position ...
__global__ void insert_diag_move_kernel(const int64_t *__restrict__ row,
                                        const int64_t *__restrict__ col, int64_t E, int64_t k,
                                        int64_t start, int64_t num_diag, int64_t *__restrict__ row_out,
                                        int64_t *__restrict__ col_out, int64_t *__restrict__ src_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const int64_t r = row[i], c = col[i];
  const int64_t p = i + diag_before(r, c, k, start, num_diag);
  row_out[p] = r;
  col_out[p] = c;
  src_out[p] = i;
}

// ... and diagonal entry j lands after the `lower_bound` of (d, d + k) among the existing
// entries plus the j diagonal entries before it; src = E + j.
__global__ void insert_diag_fill_kernel(const int64_t *__restrict__ row,
                                        const int64_t *__restrict__ col, int64_t E, int64_t k,
                                        int64_t start, int64_t num_diag, int64_t *__restrict__ row_out,
                                        int64_t *__restrict__ col_out, int64_t *__restrict__ src_out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= num_diag) return;
  const int64_t d = start + j, c = d + k;
  int64_t lo = 0, hi = E;  // first i with (row[i], col[i]) >= (d, c)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int64_t r = row[mid];
    if (r < d || (r == d && col[mid] < c)) lo = mid + 1; else hi = mid;
  }
  const int64_t p = lo + j;
  row_out[p] = d;
  col_out[p] = c;
  src_out[p] = E + j;
}

// fused set_diag on a pattern that may still hold entries on the k-th diagonal: `pos` is the
// exclusive scan of the OFF_DIAG keep flags (tsamd_filter_plan).  Kept entry i lands at
// pos[i] + (diagonal slots before it); entries on the diagonal are dropped.
__global__ void set_diag_move_kernel(const int64_t *__restrict__ pos, const int64_t *__restrict__ row,
                                     const int64_t *__restrict__ col, int64_t E, int64_t k,
                                     int64_t start, int64_t num_diag, int64_t *__restrict__ row_out,
                                     int64_t *__restrict__ col_out, int64_t *__restrict__ src_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const int64_t q = pos[i];
  if (pos[i + 1] == q) return;
  const int64_t r = row[i], c = col[i];
  const int64_t p = q + diag_before(r, c, k, start, num_diag);
  row_out[p] = r;
  col_out[p] = c;
  src_out[p] = i;
}

// diagonal entry j: kept entries before (d, d + k) = pos[lower_bound of (d, d + k) in the input]
__global__ void set_diag_fill_kernel(const int64_t *__restrict__ pos, const int64_t *__restrict__ row,
                                     const int64_t *__restrict__ col, int64_t E, int64_t k,
                                     int64_t start, int64_t num_diag, int64_t *__restrict__ row_out,
                                     int64_t *__restrict__ col_out, int64_t *__restrict__ src_out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= num_diag) return;
  const int64_t d = start + j, c = d + k;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int64_t r = row[mid];
    if (r < d || (r == d && col[mid] < c)) lo = mid + 1; else hi = mid;
  }
  const int64_t p = pos[lo] + j;
  row_out[p] = d;
  col_out[p] = c;
  src_out[p] = E + j;
}

// ---- tile compaction: the same filter without an 8-byte flag / position per entry --------------
// Pass 1 counts the kept entries of every 2048-entry tile, a scan over the (few) tile counts gives
// each tile its output offset, pass 2 re-evaluates the predicate and places the kept entries in
// input order (wave ballots + a 4-entry LDS prefix per 256-entry slice).  Traffic: the predicate's
// inputs twice + the outputs once, against flags + scan + positions (5 x 8 B per entry) before.
constexpr int kFilterTile = 2048;
constexpr int kFilterSlices = kFilterTile / 256;

__global__ __launch_bounds__(256) void filter_count_kernel(int pred, const int64_t *__restrict__ row,
                                                           const int64_t *__restrict__ col,
                                                           const uint8_t *__restrict__ mask,
                                                           const int64_t *__restrict__ map, int64_t n,
                                                           int64_t a, int64_t b, int64_t *__restrict__ tile_cnt) {
  __shared__ int wsum[4];
  const int64_t base = (int64_t)blockIdx.x * kFilterTile;
  int c = 0;
#define TSAMD_COUNT_BODY(P)                                                                    \
  _Pragma("unroll") for (int j = 0; j < kFilterSlices; ++j) {                                  \
    const int64_t i = base + (int64_t)j * 256 + threadIdx.x;                                   \
    const int64_t ic = i < n ? i : n - 1; /* the loads of all slices are issued unconditionally */ \
    c += (keep_entry_t<P>(row, col, mask, map, ic, a, b) && i < n) ? 1 : 0;                    \
  }
  if (n > 0) {
    TSAMD_FILTER_DISPATCH(pred, TSAMD_COUNT_BODY)
  }
#undef TSAMD_COUNT_BODY
  for (int off = 32; off > 0; off >>= 1) c += lane_xor(c, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = (int64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void filter_write_kernel(
    int pred, const int64_t *__restrict__ row, const int64_t *__restrict__ col,
    const uint8_t *__restrict__ mask, const int64_t *__restrict__ map, int64_t n, int64_t a, int64_t b,
    const int64_t *__restrict__ tile_off, const int64_t *__restrict__ row_map,
    const int64_t *__restrict__ col_map, int64_t row_shift, int64_t col_shift,
    int64_t *__restrict__ row_out, int64_t *__restrict__ col_out, int64_t *__restrict__ src_out) {
  __shared__ int wcnt[kFilterSlices][4];
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  const int64_t base = (int64_t)blockIdx.x * kFilterTile;
  bool keep[kFilterSlices];
  int before[kFilterSlices];  // kept entries of this wave's slice part in lower lanes
#pragma unroll
  for (int j = 0; j < kFilterSlices; ++j) keep[j] = false;
#define TSAMD_KEEP_BODY(P)                                                     \
  _Pragma("unroll") for (int j = 0; j < kFilterSlices; ++j) {                  \
    const int64_t i = base + (int64_t)j * 256 + threadIdx.x;                   \
    const int64_t ic = i < n ? i : n - 1;                                      \
    keep[j] = keep_entry_t<P>(row, col, mask, map, ic, a, b) && i < n;         \
  }
  if (n > 0) {
    TSAMD_FILTER_DISPATCH(pred, TSAMD_KEEP_BODY)
  }
#undef TSAMD_KEEP_BODY
#pragma unroll
  for (int j = 0; j < kFilterSlices; ++j) {
    const unsigned long long m = __ballot(keep[j]);
    before[j] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[j][wave] = __popcll(m);
  }
  __syncthreads();
  int64_t run = tile_off[blockIdx.x];
  // the ids of all slices first (unconditional loads of a clamped position: issued together), then the maps, then the
  // stores -- per slice inside `if (keep[j])` every id was a round trip of its own, and its map entry another one
  int64_t rv[kFilterSlices], cv[kFilterSlices];
#pragma unroll
  for (int j = 0; j < kFilterSlices; ++j) {
    const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
    const int64_t ic = (i < n ? i : n - 1);
    rv[j] = (row_out && n > 0) ? row[ic] : 0;
    cv[j] = (col_out && n > 0) ? col[ic] : 0;
  }
  if (row_map != nullptr && row_out != nullptr && n > 0) {
#pragma unroll
    for (int j = 0; j < kFilterSlices; ++j) rv[j] = row_map[rv[j]];
  }
  if (col_map != nullptr && col_out != nullptr && n > 0) {
#pragma unroll
    for (int j = 0; j < kFilterSlices; ++j) cv[j] = col_map[cv[j]];
  }
#pragma unroll
  for (int j = 0; j < kFilterSlices; ++j) {
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int cw = wcnt[j][w];
      wbase += w < wave ? cw : 0;
      total += cw;
    }
    if (keep[j]) {
      const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
      const int64_t p = run + wbase + before[j];
      if (row_out) row_out[p] = rv[j] - row_shift;
      if (col_out) col_out[p] = cv[j] - col_shift;
      if (src_out) src_out[p] = i;
    }
    run += total;
  }
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" size_t tsamd_select_workspace_bytes(int64_t K) { return scan_workspace_bytes(K + 1); }

extern "C" int tsamd_select_plan(const int64_t *ptr, int64_t S, const int64_t *idx, int64_t K,
                                 int64_t *out_ptr, int64_t *info, void *workspace,
                                 size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (S < 0 || K < 0 || !out_ptr || !info || !ptr) return TSAMD_ERR_INVALID;
  if (K > 0 && !idx) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_select_workspace_bytes(K)) return TSAMD_ERR_WORKSPACE;
  TSAMD_HIP_TRY(hipMemsetAsync(info, 0, 2 * sizeof(int64_t), stream));
  // out_ptr[K] = 0 so that the scan over K+1 counts leaves the total in out_ptr[K]
  TSAMD_HIP_TRY(hipMemsetAsync(out_ptr + K, 0, sizeof(int64_t), stream));
  if (K > 0) {
    hipLaunchKernelGGL(select_count_kernel, dim3((unsigned int)ceil_div(K, 256)), dim3(256), 0,
                       stream, ptr, S, idx, K, out_ptr,
                       reinterpret_cast<unsigned long long *>(info + 1));
    TSAMD_LAUNCH_CHECK();
  }
  return exclusive_scan_i64(out_ptr, out_ptr, K + 1, info, workspace, stream);
}

extern "C" int tsamd_select_fill(const int64_t *ptr, int64_t S, const int64_t *ind,
                                 const int64_t *idx, int64_t K, const int64_t *out_ptr,
                                 int64_t total, int64_t *seg_out, int64_t *ind_out,
                                 int64_t *pos_out, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (S < 0 || K < 0 || total < 0) return TSAMD_ERR_INVALID;
  if (K == 0 || total == 0) return TSAMD_OK;
  if (!ptr || !idx || !out_ptr || (ind_out && !ind)) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(select_fill_kernel, dim3((unsigned int)ceil_div(total, kExpandTile)), dim3(256), 0,
                     stream, ptr, S, ind, idx, K, out_ptr, total, seg_out, ind_out, pos_out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_filter_workspace_bytes(int64_t n) { return scan_workspace_bytes(n + 1); }

extern "C" int tsamd_filter_plan(int pred, const int64_t *row, const int64_t *col,
                                 const uint8_t *mask, const int64_t *map, int64_t n, int64_t a,
                                 int64_t b, int64_t *pos, int64_t *count, void *workspace,
                                 size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || !pos || !count) return TSAMD_ERR_INVALID;
  if (pred < TSAMD_KEEP_COL_RANGE || pred > TSAMD_KEEP_COL_MAPPED) return TSAMD_ERR_UNSUPPORTED;
  const bool need_row = pred == TSAMD_KEEP_OFF_DIAG || pred == TSAMD_KEEP_MASK_ROW;
  const bool need_col = pred == TSAMD_KEEP_COL_RANGE || pred == TSAMD_KEEP_OFF_DIAG ||
                        pred == TSAMD_KEEP_MASK_COL || pred == TSAMD_KEEP_COL_MAPPED;
  const bool need_mask = pred >= TSAMD_KEEP_MASK && pred <= TSAMD_KEEP_MASK_COL;
  const bool need_map = pred == TSAMD_KEEP_COL_MAPPED;
  if (n > 0 && ((need_row && !row) || (need_col && !col) || (need_mask && !mask) || (need_map && !map)))
    return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_filter_workspace_bytes(n)) return TSAMD_ERR_WORKSPACE;
  TSAMD_HIP_TRY(hipMemsetAsync(pos + n, 0, sizeof(int64_t), stream));
  if (n > 0) {
    hipLaunchKernelGGL(filter_flags_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0,
                       stream, pred, row, col, mask, map, n, a, b, pos);
    TSAMD_LAUNCH_CHECK();
  }
  return exclusive_scan_i64(pos, pos, n + 1, count, workspace, stream);
}

extern "C" int tsamd_filter_apply(const int64_t *pos, const int64_t *row, const int64_t *col,
                                  int64_t n, const int64_t *row_map, const int64_t *col_map,
                                  int64_t row_shift, int64_t col_shift, int64_t *row_out,
                                  int64_t *col_out, int64_t *src_out, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0) return TSAMD_ERR_INVALID;
  if (n == 0) return TSAMD_OK;
  if (!pos || (row_out && !row) || (col_out && !col)) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(filter_apply_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                     pos, row, col, n, row_map, col_map, row_shift, col_shift, row_out, col_out,
                     src_out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_scatter_rows(const int64_t *row, const int64_t *col, int64_t n,
                                  const int64_t *delta, int64_t col_shift, int64_t src_offset,
                                  int64_t *row_out, int64_t *col_out, int64_t *src_out,
                                  void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0) return TSAMD_ERR_INVALID;
  if (n == 0) return TSAMD_OK;
  if (!row || !delta || (col_out && !col)) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                     row, col, n, delta, col_shift, src_offset, row_out, col_out, src_out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

static inline void diag_extent(int64_t M, int64_t N, int64_t k, int64_t *start, int64_t *num_diag) {
  int64_t n = k < 0 ? (M + k < N ? M + k : N) : (M < N - k ? M : N - k);
  *num_diag = n < 0 ? 0 : n;
  *start = k < 0 ? -k : 0;
}

extern "C" int64_t tsamd_num_diag(int64_t M, int64_t N, int64_t k) {
  int64_t start, n;
  diag_extent(M, N, k, &start, &n);
  return n;
}

extern "C" int tsamd_non_diag_mask(const int64_t *row, const int64_t *col, int64_t E, int64_t M,
                                   int64_t N, int64_t k, uint8_t *mask, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0) return TSAMD_ERR_INVALID;
  int64_t start, num_diag;
  diag_extent(M, N, k, &start, &num_diag);
  if (E + num_diag == 0) return TSAMD_OK;
  if (!mask || (E > 0 && (!row || !col))) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(mask, 0, (size_t)(E + num_diag), stream));
  if (E == 0) return TSAMD_OK;
  hipLaunchKernelGGL(non_diag_mask_kernel, dim3((unsigned int)ceil_div(E, 256)), dim3(256), 0, stream,
                     row, col, E, k, start, num_diag, mask);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_insert_diag(const int64_t *row, const int64_t *col, int64_t E, int64_t M,
                                 int64_t N, int64_t k, int64_t *row_out, int64_t *col_out,
                                 int64_t *src_out, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0) return TSAMD_ERR_INVALID;
  int64_t start, num_diag;
  diag_extent(M, N, k, &start, &num_diag);
  if (E + num_diag == 0) return TSAMD_OK;
  if (!row_out || !col_out || !src_out || (E > 0 && (!row || !col))) return TSAMD_ERR_INVALID;
  if (E > 0) {
    hipLaunchKernelGGL(insert_diag_move_kernel, dim3((unsigned int)ceil_div(E, 256)), dim3(256), 0,
                       stream, row, col, E, k, start, num_diag, row_out, col_out, src_out);
    TSAMD_LAUNCH_CHECK();
  }
  if (num_diag > 0) {
    hipLaunchKernelGGL(insert_diag_fill_kernel, dim3((unsigned int)ceil_div(num_diag, 256)), dim3(256),
                       0, stream, row, col, E, k, start, num_diag, row_out, col_out, src_out);
    TSAMD_LAUNCH_CHECK();
  }
  return TSAMD_OK;
}

extern "C" int tsamd_set_diag_apply(const int64_t *pos, const int64_t *row, const int64_t *col,
                                    int64_t E, int64_t M, int64_t N, int64_t k, int64_t *row_out,
                                    int64_t *col_out, int64_t *src_out, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0 || !pos) return TSAMD_ERR_INVALID;
  int64_t start, num_diag;
  diag_extent(M, N, k, &start, &num_diag);
  if (E + num_diag == 0) return TSAMD_OK;
  if (!row_out || !col_out || !src_out || (E > 0 && (!row || !col))) return TSAMD_ERR_INVALID;
  if (E > 0) {
    hipLaunchKernelGGL(set_diag_move_kernel, dim3((unsigned int)ceil_div(E, 256)), dim3(256), 0, stream,
                       pos, row, col, E, k, start, num_diag, row_out, col_out, src_out);
    TSAMD_LAUNCH_CHECK();
  }
  if (num_diag > 0) {
    hipLaunchKernelGGL(set_diag_fill_kernel, dim3((unsigned int)ceil_div(num_diag, 256)), dim3(256), 0,
                       stream, pos, row, col, E, k, start, num_diag, row_out, col_out, src_out);
    TSAMD_LAUNCH_CHECK();
  }
  return TSAMD_OK;
}

extern "C" size_t tsamd_filter_tiles_workspace_bytes(int64_t n) {
  const int64_t tiles = ceil_div(n > 0 ? n : 1, (int64_t)kFilterTile);
  return align_up(sizeof(int64_t) * (size_t)(tiles + 1), 256) + scan_workspace_bytes(tiles + 1);
}

static bool filter_args_ok(int pred, const int64_t *row, const int64_t *col, const uint8_t *mask,
                           const int64_t *map, int64_t n) {
  if (pred < TSAMD_KEEP_COL_RANGE || pred > TSAMD_KEEP_COL_MAPPED) return false;
  const bool need_row = pred == TSAMD_KEEP_OFF_DIAG || pred == TSAMD_KEEP_MASK_ROW;
  const bool need_col = pred == TSAMD_KEEP_COL_RANGE || pred == TSAMD_KEEP_OFF_DIAG ||
                        pred == TSAMD_KEEP_MASK_COL || pred == TSAMD_KEEP_COL_MAPPED;
  const bool need_mask = pred >= TSAMD_KEEP_MASK && pred <= TSAMD_KEEP_MASK_COL;
  const bool need_map = pred == TSAMD_KEEP_COL_MAPPED;
  return n == 0 || !((need_row && !row) || (need_col && !col) || (need_mask && !mask) || (need_map && !map));
}

extern "C" int tsamd_filter_count(int pred, const int64_t *row, const int64_t *col, const uint8_t *mask,
                                  const int64_t *map, int64_t n, int64_t a, int64_t b, int64_t *count,
                                  void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || !count) return TSAMD_ERR_INVALID;
  if (pred < TSAMD_KEEP_COL_RANGE || pred > TSAMD_KEEP_COL_MAPPED) return TSAMD_ERR_UNSUPPORTED;
  if (!filter_args_ok(pred, row, col, mask, map, n)) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_filter_tiles_workspace_bytes(n)) return TSAMD_ERR_WORKSPACE;
  const int64_t tiles = ceil_div(n > 0 ? n : 1, (int64_t)kFilterTile);
  int64_t *tile_off = reinterpret_cast<int64_t *>(workspace);
  void *scan_ws = reinterpret_cast<char *>(workspace) + align_up(sizeof(int64_t) * (size_t)(tiles + 1), 256);
  TSAMD_HIP_TRY(hipMemsetAsync(tile_off + tiles, 0, sizeof(int64_t), stream));
  hipLaunchKernelGGL(filter_count_kernel, dim3((unsigned int)tiles), dim3(256), 0, stream, pred, row, col,
                     mask, map, n, a, b, tile_off);
  TSAMD_LAUNCH_CHECK();
  return exclusive_scan_i64(tile_off, tile_off, tiles + 1, count, scan_ws, stream);
}

extern "C" int tsamd_filter_write(int pred, const int64_t *row, const int64_t *col, const uint8_t *mask,
                                  const int64_t *map, int64_t n, int64_t a, int64_t b,
                                  const void *workspace, const int64_t *row_map, const int64_t *col_map,
                                  int64_t row_shift, int64_t col_shift, int64_t *row_out,
                                  int64_t *col_out, int64_t *src_out, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || !workspace) return TSAMD_ERR_INVALID;
  if (n == 0) return TSAMD_OK;
  if (!filter_args_ok(pred, row, col, mask, map, n) || (row_out && !row) || (col_out && !col))
    return TSAMD_ERR_INVALID;
  const int64_t tiles = ceil_div(n, (int64_t)kFilterTile);
  hipLaunchKernelGGL(filter_write_kernel, dim3((unsigned int)tiles), dim3(256), 0, stream, pred, row, col,
                     mask, map, n, a, b, reinterpret_cast<const int64_t *>(workspace), row_map, col_map,
                     row_shift, col_shift, row_out, col_out, src_out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}
