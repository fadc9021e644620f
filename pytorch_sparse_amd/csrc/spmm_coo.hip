// Legacy torch_sparse.spmm(index, value, m, n, matrix) on a SMALL unsorted COO: one launch.
//
// Replaces the three ATen calls of torch_sparse/spmm.py:25-31 (index_select, multiply, scatter_add) for inputs
// where launch latency is all there is (BASELINE.json configs[0]: 1 000 x 1 000, 5 000 draws, F = 16: the
// general route -- order probe, radix sort, ind2ptr, CSR SpMM, 8+ launches -- took 72 us against 37 us for the
// reference on one host core).  No sort, no host sync, no zero fill, no global atomics:
//   * workgroup g owns the rows [g R, (g + 1) R) of `out`, R F accumulators in LDS (64 KB);
//   * every workgroup scans ALL entries (E is small: the scan is G E row ids out of L2), and a thread that finds
//     an entry of its workgroup's rows adds value * matrix[col, :] into the LDS accumulators (LDS atomics: the
//     order of the additions of one row is not fixed -- fp sums agree with the reference up to association, as
//     a device scatter_add's do; integer sums are exact, wrapping like the element type);
//   * the accumulators are converted and stored once, coalesced: rows without entries get their zeros here.
// Duplicated (row, col) pairs add up, as in the reference.  f16 / bf16: products rounded to the element type
// (what the reference multiplies in), sums in fp32, one rounding at the end (the CSR kernels' convention).
#include "common.h"

namespace tsamd {
namespace {

constexpr int kCooThreads = 1024;
constexpr int kCooAccBytes = 65536;
constexpr int kCooScan = 4;  // entries per thread and scan step: all their loads in flight together

template <typename A>
__device__ __forceinline__ void lds_add(A *p, A v) {
  atomicAdd(p, v);
}
template <>
__device__ __forceinline__ void lds_add<int64_t>(int64_t *p, int64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}

template <typename T>
__global__ __launch_bounds__(kCooThreads) void spmm_coo_small_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ col, const T *__restrict__ value,
    const T *__restrict__ mat, T *__restrict__ out, int64_t E, int64_t M, int64_t N, int K, int rows_per_wg) {
  using A = typename Traits<T>::acc_t;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kCooAccBytes];
  A *acc = reinterpret_cast<A *>(s_raw);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < M ? r0 + rows_per_wg : M;
  const int cells = (int)(r1 - r0) * K;
  for (int i = threadIdx.x; i < cells; i += kCooThreads) acc[i] = (A)0;
  __syncthreads();
  for (int64_t base = 0; base < E; base += (int64_t)kCooThreads * kCooScan) {
    int64_t r[kCooScan], c[kCooScan];
    A w[kCooScan];
    bool mine[kCooScan];
#pragma unroll
    for (int u = 0; u < kCooScan; ++u) {
      const int64_t e = base + u * kCooThreads + threadIdx.x;
      r[u] = e < E ? row[e] : -1;
    }
#pragma unroll
    for (int u = 0; u < kCooScan; ++u) {
      const int64_t e = base + u * kCooThreads + threadIdx.x;
      mine[u] = r[u] >= r0 && r[u] < r1;
      c[u] = mine[u] ? col[e] : 0;
      // an entry whose column id lies outside [0, N) is skipped, like one whose row id lies outside [0, M): the
      // reference's index_select / scatter_add raise a device assert for both (torch_sparse/spmm.py:25-31); nothing
      // here reads or writes out of bounds
      mine[u] = mine[u] && (uint64_t)c[u] < (uint64_t)N;
      w[u] = mine[u] ? Traits<T>::to_acc(value[e]) : (A)0;
    }
#pragma unroll
    for (int u = 0; u < kCooScan; ++u) {
      if (!mine[u]) continue;
      const T *__restrict__ x = mat + (size_t)c[u] * K;
      A *__restrict__ dst = acc + (size_t)(r[u] - r0) * K;
      constexpr int kVec = 16 / sizeof(T) > 8 ? 8 : 16 / sizeof(T);
      int f = 0;
      if ((K % kVec) == 0) {  // rows of 16-byte packets (the matrix is contiguous: every row starts aligned)
        for (; f < K; f += kVec) {
          const Pack<T, kVec> p = *reinterpret_cast<const Pack<T, kVec> *>(x + f);
#pragma unroll
          for (int j = 0; j < kVec; ++j) lds_add(dst + f + j, Traits<T>::round_acc(w[u] * Traits<T>::to_acc(p.v[j])));
        }
      } else {
        for (; f < K; ++f) lds_add(dst + f, Traits<T>::round_acc(w[u] * Traits<T>::to_acc(x[f])));
      }
    }
  }
  __syncthreads();
  T *__restrict__ o = out + (size_t)r0 * K;
  for (int i = threadIdx.x; i < cells; i += kCooThreads) o[i] = Traits<T>::from_acc(acc[i]);
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

// rows of `out` one workgroup can hold, 0 = the direct route does not apply
static int coo_rows_per_wg(int dtype, int64_t K) {
  const size_t a = acc_size(dtype);
  if (a == 0 || K <= 0) return 0;
  return (int)((size_t)kCooAccBytes / a / (size_t)K);
}

// 1 when tsamd_spmm_coo_small takes this problem (small enough that launch latency dominates the sorted route)
extern "C" int tsamd_spmm_coo_small_supported(int dtype, int64_t E, int64_t M, int64_t K) {
  const int rmax = coo_rows_per_wg(dtype, K);
  if (rmax <= 0 || E < 0 || M <= 0) return 0;
  if (E > 65536 || E * K > (1 << 20)) return 0;
  return ceil_div(M, rmax) <= 64 ? 1 : 0;
}

extern "C" int tsamd_spmm_coo_small(int dtype, const int64_t *row, const int64_t *col, const void *value,
                                    const void *mat, void *out, int64_t E, int64_t M, int64_t N, int64_t K,
                                    void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0 || K < 0 || (E > 0 && (!row || !col || !value || !mat))) return TSAMD_ERR_INVALID;
  if (M == 0 || K == 0) return TSAMD_OK;
  if (!out) return TSAMD_ERR_INVALID;
  if (!tsamd_spmm_coo_small_supported(dtype, E, M, K)) return TSAMD_ERR_UNSUPPORTED;
  const int rmax = coo_rows_per_wg(dtype, K);
  // enough workgroups that the per-entry work of a workgroup stays ~512 entries, no more than the rows allow
  int64_t want = E / 512;
  want = want < 1 ? 1 : (want > 32 ? 32 : want);
  int64_t R = ceil_div(M, want);
  R = R > rmax ? rmax : R;
  const int64_t G = ceil_div(M, R);
  return TSAMD_DISPATCH_DTYPE_ALL(dtype, [&]() -> int {
    hipLaunchKernelGGL((spmm_coo_small_kernel<scalar_t>), dim3((unsigned int)G), dim3(kCooThreads), 0, stream, row, col,
                       reinterpret_cast<const scalar_t *>(value), reinterpret_cast<const scalar_t *>(mat),
                       reinterpret_cast<scalar_t *>(out), E, M, N, (int)K, (int)R);
    TSAMD_LAUNCH_CHECK();
    return TSAMD_OK;
  });
}
