// COO row ids <-> CSR row pointer on gfx950.
// Replaces ind2ptr_cuda / ptr2ind_cuda (csrc/cuda/convert_cuda.cu:9-67) and
// ind2ptr_cpu / ptr2ind_cpu (csrc/cpu/convert_cpu.cpp:7-57) of the reference.
#include "common.h"
#include "expand.h"

namespace tsamd {
namespace {

// Thread t in [0, E] owns the boundary between ind[t-1] and ind[t] and fills every row pointer
// that falls into it (empty rows make the run longer than 1).  Runs of more than kLongRun empty rows
// are left at the -1 the output was pre-set to and resolved by ind2ptr_long_kernel, one thread per
// row pointer with a binary search: a single thread filling a 32 M-row gap took 840 ms.
constexpr int64_t kLongRun = 1024;

__global__ void ind2ptr_kernel(const int64_t *__restrict__ ind, int64_t *__restrict__ out,
                               int64_t M, int64_t E) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > E) return;
  // ids outside [0, M) write nothing (the sort-on-construct path enqueues this kernel BEFORE its range check is
  // read back: an invalid id must not reach past the (M + 1)-word output before the constructor raises)
  int64_t lo = t == 0 ? 0 : ind[t - 1] + 1;
  int64_t hi = t == E ? M : ind[t];
  lo = lo < 0 ? 0 : lo;
  hi = hi > M ? M : hi;
  if (t == E) out[M] = E;  // (always: the pre-set below may stop one word short of it)
  if (hi - lo >= kLongRun) return;
  for (int64_t i = lo; i <= hi; ++i) out[i] = t;
}

// out[i] = number of entries with ind < i  (first e with ind[e] >= i), only where still unset
__global__ void ind2ptr_long_kernel(const int64_t *__restrict__ ind, int64_t *__restrict__ out,
                                    int64_t M, int64_t E) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > M || out[i] >= 0) return;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (ind[mid] < i) lo = mid + 1; else hi = mid;
  }
  out[i] = lo;
}

// Balanced by output entries (expand.h): a tile of 2048 consecutive entries per workgroup, the rows
// that intersect it staged in LDS, one LDS binary search per entry, coalesced stores.  Hub rows are
// spread over many workgroups (a row-per-workgroup mapping took 65 ms on a 30 M-entry row).
__global__ __launch_bounds__(256) void ptr2ind_kernel(const int64_t *__restrict__ ptr,
                                                      int64_t *__restrict__ out, int64_t M, int64_t E) {
  __shared__ int64_t sp[kExpandTile];
  __shared__ int64_t span[2];
  const int64_t e0 = (int64_t)blockIdx.x * kExpandTile;
  const int64_t e1 = e0 + kExpandTile < E ? e0 + kExpandTile : E;
  int64_t lo, hi;
  tile_span(ptr, M, e0, e1, span, &lo, &hi);
  const int64_t S = hi - lo + 1;
  if (S <= kExpandTile) {
    for (int i = threadIdx.x; i < (int)S; i += blockDim.x) sp[i] = ptr[lo + i];
    __syncthreads();
    for (int64_t e = e0 + threadIdx.x; e < e1; e += blockDim.x)
      out[e] = lo + segment_of_lds(sp, (int)S, e);
  } else {  // more rows than entries in this tile (long runs of empty rows): search in place
    for (int64_t e = e0 + threadIdx.x; e < e1; e += blockDim.x)
      out[e] = lo + segment_of(ptr + lo, S, e);
  }
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" int tsamd_ind2ptr(const int64_t *ind, int64_t M, int64_t E, int64_t *out,
                             void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || E < 0 || !out || (E > 0 && !ind)) return TSAMD_ERR_INVALID;
  if (E == 0) {
    TSAMD_HIP_TRY(hipMemsetAsync(out, 0, sizeof(int64_t) * (size_t)(M + 1), stream));
    return TSAMD_OK;
  }
  const int64_t n = E + 1;
  // -1 = unset.  A fill whose length is not a multiple of 16 bytes is TWO fill kernels (bulk + tail, ~5 us each): the
  // pre-set covers the multiple of 16 below (M + 1) * 8 and the kernel writes out[M] itself.
  TSAMD_HIP_TRY(hipMemsetAsync(out, 0xff, (sizeof(int64_t) * (size_t)(M + 1)) & ~(size_t)15, stream));
  hipLaunchKernelGGL(ind2ptr_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                     ind, out, M, E);
  TSAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ind2ptr_long_kernel, dim3((unsigned int)ceil_div(M + 1, 256)), dim3(256), 0,
                     stream, ind, out, M, E);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_ptr2ind(const int64_t *ptr, int64_t M, int64_t E, int64_t *out,
                             void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || E < 0 || !ptr || (E > 0 && !out)) return TSAMD_ERR_INVALID;
  if (E == 0 || M == 0) return TSAMD_OK;
  hipLaunchKernelGGL(ptr2ind_kernel, dim3((unsigned int)ceil_div(E, kExpandTile)), dim3(256), 0,
                     stream, ptr, out, M, E);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}
