// COO ordering, sorting and coalescing on gfx950.
//
// Replaces the Python/ATen compositions of the reference's SparseStorage:
//   * sort-on-construct          torch_sparse/storage.py:149-162  (key build, `.any()` probe,
//                                index_sort, three gathers)
//   * csr2csc / csc2csr          torch_sparse/storage.py:407-429  (argsort of col*M+row)
//   * is_coalesced / coalesce    torch_sparse/storage.py:431-466  (adjacent-duplicate mask,
//                                boolean compaction, torch_scatter.segment_csr)
// with fused kernels: one order probe, one radix sort that emits sorted (row, col) and the
// permutation, one flag+scan+compaction, one segmented reduction that reads the values
// through the permutation (no materialised value[perm]).
#include "common.h"
#include "scan.h"
#include "sort.h"

#include <type_traits>

namespace tsamd {
namespace {

__global__ void make_keys_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                 int64_t n, int64_t ncols, int64_t *__restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = row[i] * ncols + col[i];
}

__global__ void decode_keys_kernel(const int64_t *__restrict__ keys, int64_t n, int64_t ncols,
                                   int64_t *__restrict__ row, int64_t *__restrict__ col) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t k = keys[i];
  const int64_t r = k / ncols;
  if (row) row[i] = r;
  if (col) col[i] = k - r * ncols;
}

// counts[0] += #{i : key[i] < key[i-1]}, counts[1] += #{i : key[i] == key[i-1]}
// grid-stride with per-thread counters: one pair of atomics per wave at the very end
__global__ __launch_bounds__(256) void order_probe_kernel(const int64_t *__restrict__ row,
                                                         const int64_t *__restrict__ col, int64_t n,
                                                         int64_t ncols, unsigned long long *counts) {
  unsigned int desc = 0, dup = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = row[i - 1] * ncols + col[i - 1];
    const int64_t b = row[i] * ncols + col[i];
    desc += b < a;
    dup += b == a;
  }
  for (int off = 32; off > 0; off >>= 1) {
    desc += lane_xor(desc, off);
    dup += lane_xor(dup, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (desc) atomicAdd(&counts[0], (unsigned long long)desc);
    if (dup) atomicAdd(&counts[1], (unsigned long long)dup);
  }
}

// order probe + range check in one pass: counts[0..1] as above, counts[2] = max row id, counts[3] = max col id
// (counts[2..3] start at 0; ids are assumed non-negative)
__global__ __launch_bounds__(256) void order_check_kernel(const int64_t *__restrict__ row,
                                                         const int64_t *__restrict__ col, int64_t n,
                                                         unsigned long long *counts) {
  unsigned int desc = 0, dup = 0;
  int64_t mr = 0, mc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = row[i], c = col[i];
    mr = r > mr ? r : mr;
    mc = c > mc ? c : mc;
    if (i > 0) {  // lexicographic (row, col): the same order as row * ncols + col, without knowing ncols
      const int64_t pr = row[i - 1], pc = col[i - 1];
      desc += (r < pr) || (r == pr && c < pc);
      dup += (r == pr) && (c == pc);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    desc += lane_xor(desc, off);
    dup += lane_xor(dup, off);
    const int64_t orr = lane_xor(mr, off), oc = lane_xor(mc, off);
    mr = orr > mr ? orr : mr;
    mc = oc > mc ? oc : mc;
  }
  if ((threadIdx.x & 63) == 0) {
    if (desc) atomicAdd(&counts[0], (unsigned long long)desc);
    if (dup) atomicAdd(&counts[1], (unsigned long long)dup);
    atomicMax(&counts[2], (unsigned long long)mr);
    atomicMax(&counts[3], (unsigned long long)mc);
  }
}

// the outputs of a device-decided sort when the input turned out to be sorted: a copy and the identity
__global__ void sort_auto_finish_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                        const int64_t *__restrict__ keys_sorted, int64_t n, int64_t ncols,
                                        const int64_t *__restrict__ todo, int64_t *__restrict__ row_out,
                                        int64_t *__restrict__ col_out, int64_t *__restrict__ perm_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (*todo == 0) {
    if (row_out) row_out[i] = row[i];
    if (col_out) col_out[i] = col[i];
    perm_out[i] = i;
  } else {
    const int64_t k = keys_sorted[i];
    const int64_t r = k / ncols;
    if (row_out) row_out[i] = r;
    if (col_out) col_out[i] = k - r * ncols;
  }
}

__device__ inline bool is_head(const int64_t *row, const int64_t *col, int64_t i) {
  return i == 0 || row[i] != row[i - 1] || col[i] != col[i - 1];
}

__global__ void head_flags_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                  int64_t n, int64_t *__restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = is_head(row, col, i) ? 1 : 0;
}

__global__ void compact_heads_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                     int64_t n, const int64_t *__restrict__ pos,
                                     const int64_t *__restrict__ nnz, int64_t *__restrict__ row_out,
                                     int64_t *__restrict__ col_out, int64_t *__restrict__ seg_ptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) seg_ptr[*nnz] = n;
  if (i >= n) return;
  if (is_head(row, col, i)) {
    const int64_t p = pos[i];
    row_out[p] = row[i];
    col_out[p] = col[i];
    seg_ptr[p] = i;
  }
}

constexpr int SEG_MEAN = 1, SEG_MIN = 2, SEG_MAX = 3;  // sum = 0

template <typename T>
__global__ void segment_reduce_kernel(const T *__restrict__ value, const int64_t *__restrict__ perm,
                                      const int64_t *__restrict__ seg_ptr, int64_t nseg, int64_t D,
                                      int reduce, T *__restrict__ out) {
  using A = typename Traits<T>::acc_t;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nseg * D) return;
  const int64_t j = t / D, d = t - j * D;
  const int64_t s = seg_ptr[j], e = seg_ptr[j + 1];
  if (s >= e) {  // empty segment -> 0 (torch_scatter's convention for segment_csr)
    out[t] = Traits<T>::from_acc(A(0));
    return;
  }
  A acc = Traits<T>::to_acc(value[(perm ? perm[s] : s) * D + d]);
  for (int64_t i = s + 1; i < e; ++i) {
    const A v = Traits<T>::to_acc(value[(perm ? perm[i] : i) * D + d]);
    if (reduce == SEG_MIN) acc = v < acc ? v : acc;
    else if (reduce == SEG_MAX) acc = v > acc ? v : acc;
    else acc += v;
  }
  if (reduce == SEG_MEAN) {
    const int64_t cnt = e - s;
    if constexpr (std::is_integral<T>::value) {  // floor division, as torch_scatter does
      A q = acc / (A)cnt;
      if ((acc % (A)cnt != 0) && ((acc < 0) != (cnt < 0))) --q;
      acc = q;
    } else {
      acc = acc / (A)cnt;
    }
  }
  out[t] = Traits<T>::from_acc(acc);
}

int key_bits_for(int64_t rows, int64_t cols) {
  // keys are < rows * cols
  unsigned __int128 lim = (unsigned __int128)(rows > 0 ? rows : 1) * (unsigned __int128)(cols > 0 ? cols : 1);
  int bits = 0;
  while (bits < 63 && ((unsigned __int128)1 << bits) < lim) ++bits;
  return bits;
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" size_t tsamd_sort_coo_workspace_bytes(int64_t E) {
  const size_t n = (size_t)(E > 0 ? E : 1);
  return 2 * align_up(sizeof(int64_t) * n, 256) + sort_pairs_workspace_bytes(E);
}

extern "C" int tsamd_sort_coo(const int64_t *row, const int64_t *col, int64_t E, int64_t M,
                              int64_t N, int64_t *row_out, int64_t *col_out, int64_t *perm_out,
                              void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0) return TSAMD_ERR_INVALID;
  if (E == 0) return TSAMD_OK;
  if (!row || !col || !perm_out) return TSAMD_ERR_INVALID;
  if ((unsigned __int128)M * (unsigned __int128)N >= ((unsigned __int128)1 << 63))
    return TSAMD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < tsamd_sort_coo_workspace_bytes(E)) return TSAMD_ERR_WORKSPACE;
  char *p = reinterpret_cast<char *>(workspace);
  int64_t *keys = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)E, 256);
  int64_t *keys_sorted = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)E, 256);
  const unsigned int blocks = (unsigned int)ceil_div(E, 256);
  hipLaunchKernelGGL(make_keys_kernel, dim3(blocks), dim3(256), 0, stream, row, col, E, N, keys);
  TSAMD_LAUNCH_CHECK();
  int st = sort_pairs(keys, nullptr, keys_sorted, perm_out, E, key_bits_for(M, N), p, stream);
  if (st != TSAMD_OK) return st;
  if (row_out || col_out) {
    hipLaunchKernelGGL(decode_keys_kernel, dim3(blocks), dim3(256), 0, stream,
                       (const int64_t *)keys_sorted, E, N, row_out, col_out);
    TSAMD_LAUNCH_CHECK();
  }
  return TSAMD_OK;
}

// sort_coo decided on the device: counts_out[0..1] = (#descents, #adjacent duplicates) of the INPUT; when
// there is no descent the radix passes return at once and the outputs are a copy + the identity.
extern "C" int tsamd_sort_coo_auto(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                                   int64_t *row_out, int64_t *col_out, int64_t *perm_out,
                                   int64_t *counts_out, void *workspace, size_t workspace_bytes,
                                   void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0 || !counts_out) return TSAMD_ERR_INVALID;
  int st = tsamd_coo_order(row, col, E, N, counts_out, stream_);
  if (st != TSAMD_OK || E == 0) return st;
  if (!row || !col || !perm_out) return TSAMD_ERR_INVALID;
  if ((unsigned __int128)M * (unsigned __int128)N >= ((unsigned __int128)1 << 63))
    return TSAMD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < tsamd_sort_coo_workspace_bytes(E)) return TSAMD_ERR_WORKSPACE;
  char *p = reinterpret_cast<char *>(workspace);
  int64_t *keys = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)E, 256);
  int64_t *keys_sorted = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)E, 256);
  const unsigned int blocks = (unsigned int)ceil_div(E, 256);
  hipLaunchKernelGGL(make_keys_kernel, dim3(blocks), dim3(256), 0, stream, row, col, E, N, keys);
  TSAMD_LAUNCH_CHECK();
  st = sort_pairs(keys, nullptr, keys_sorted, perm_out, E, key_bits_for(M, N), p, stream, counts_out);
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL(sort_auto_finish_kernel, dim3(blocks), dim3(256), 0, stream, row, col,
                     (const int64_t *)keys_sorted, E, N, (const int64_t *)counts_out, row_out, col_out, perm_out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

// counts_out[0..3] = (#descents, #adjacent duplicates, max row id, max col id): everything the
// SparseStorage constructor has to read back, in one pass and one transfer
extern "C" int tsamd_coo_check(const int64_t *row, const int64_t *col, int64_t E, int64_t *counts_out,
                               void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || !counts_out) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(counts_out, 0, 4 * sizeof(int64_t), stream));
  if (E == 0) return TSAMD_OK;
  if (!row || !col) return TSAMD_ERR_INVALID;
  const int64_t nblk = ceil_div(E, 256);
  hipLaunchKernelGGL(order_check_kernel, dim3((unsigned int)(nblk < 2048 ? nblk : 2048)), dim3(256), 0,
                     stream, row, col, E, reinterpret_cast<unsigned long long *>(counts_out));
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_coo_order(const int64_t *row, const int64_t *col, int64_t E, int64_t N,
                               int64_t *counts_out, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || !counts_out) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(counts_out, 0, 2 * sizeof(int64_t), stream));
  if (E <= 1) return TSAMD_OK;
  if (!row || !col) return TSAMD_ERR_INVALID;
  const int64_t nblk = ceil_div(E, 256);
  hipLaunchKernelGGL(order_probe_kernel, dim3((unsigned int)(nblk < 2048 ? nblk : 2048)), dim3(256), 0,
                     stream, row, col, E, N, reinterpret_cast<unsigned long long *>(counts_out));
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_coalesce_workspace_bytes(int64_t E) {
  return align_up(sizeof(int64_t) * (size_t)(E > 0 ? E : 1), 256) + scan_workspace_bytes(E);
}

extern "C" int tsamd_coalesce_index(const int64_t *row, const int64_t *col, int64_t E,
                                    int64_t *row_out, int64_t *col_out, int64_t *seg_ptr,
                                    int64_t *nnz_out, void *workspace, size_t workspace_bytes,
                                    void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || !nnz_out || !seg_ptr) return TSAMD_ERR_INVALID;
  if (E == 0) {
    TSAMD_HIP_TRY(hipMemsetAsync(nnz_out, 0, sizeof(int64_t), stream));
    TSAMD_HIP_TRY(hipMemsetAsync(seg_ptr, 0, sizeof(int64_t), stream));
    return TSAMD_OK;
  }
  if (!row || !col || !row_out || !col_out) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_coalesce_workspace_bytes(E)) return TSAMD_ERR_WORKSPACE;
  int64_t *pos = reinterpret_cast<int64_t *>(workspace);
  void *scan_ws = reinterpret_cast<char *>(workspace) + align_up(sizeof(int64_t) * (size_t)E, 256);
  const unsigned int blocks = (unsigned int)ceil_div(E, 256);
  hipLaunchKernelGGL(head_flags_kernel, dim3(blocks), dim3(256), 0, stream, row, col, E, pos);
  TSAMD_LAUNCH_CHECK();
  int st = exclusive_scan_i64(pos, pos, E, nnz_out, scan_ws, stream);
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL(compact_heads_kernel, dim3(blocks), dim3(256), 0, stream, row, col, E,
                     (const int64_t *)pos, (const int64_t *)nnz_out, row_out, col_out, seg_ptr);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_segment_reduce(int dtype, int reduce, const void *value, const int64_t *perm,
                                    const int64_t *seg_ptr, int64_t nseg, int64_t D, void *out,
                                    void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (nseg < 0 || D < 0) return TSAMD_ERR_INVALID;
  if (reduce < TSAMD_SUM || reduce > TSAMD_MAX) return TSAMD_ERR_UNSUPPORTED;
  if (dtype_size(dtype) == 0) return TSAMD_ERR_UNSUPPORTED;
  const int64_t total = nseg * D;
  if (total == 0) return TSAMD_OK;
  if (!value || !seg_ptr || !out) return TSAMD_ERR_INVALID;
  return TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    hipLaunchKernelGGL((segment_reduce_kernel<scalar_t>), dim3((unsigned int)ceil_div(total, 256)),
                       dim3(256), 0, stream, reinterpret_cast<const scalar_t *>(value), perm,
                       seg_ptr, nseg, D, reduce, reinterpret_cast<scalar_t *>(out));
    TSAMD_LAUNCH_CHECK();
    return (int)TSAMD_OK;
  });
}
