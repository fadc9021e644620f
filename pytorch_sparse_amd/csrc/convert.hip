// COO row ids <-> CSR row pointer on gfx950.
// Replaces ind2ptr_cuda / ptr2ind_cuda (csrc/cuda/convert_cuda.cu:9-67) and
// ind2ptr_cpu / ptr2ind_cpu (csrc/cpu/convert_cpu.cpp:7-57) of the reference.
#include "common.h"

namespace tsamd {
namespace {

// Thread t in [0, E] owns the boundary between ind[t-1] and ind[t] and fills
// every row pointer that falls into it (empty rows make the run longer than 1).
__global__ void ind2ptr_kernel(const int64_t *__restrict__ ind, int64_t *__restrict__ out,
                               int64_t M, int64_t E) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > E) return;
  const int64_t lo = t == 0 ? 0 : ind[t - 1] + 1;
  const int64_t hi = t == E ? M : ind[t];
  for (int64_t i = lo; i <= hi; ++i) out[i] = t;
}

constexpr int kRowsPerBlock = 256;

// A workgroup owns 256 consecutive rows: their pointers go to LDS, then the
// block streams over the rows' edge range with coalesced stores, each thread
// locating its edge's row by a binary search in LDS (hub rows cost nothing extra).
__global__ __launch_bounds__(256) void ptr2ind_kernel(const int64_t *__restrict__ ptr,
                                                      int64_t *__restrict__ out, int64_t M) {
  __shared__ int64_t sp[kRowsPerBlock + 1];
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
  const int nrows = (int)((M - r0) < kRowsPerBlock ? (M - r0) : kRowsPerBlock);
  for (int i = threadIdx.x; i <= nrows; i += blockDim.x) sp[i] = ptr[r0 + i];
  __syncthreads();
  const int64_t e0 = sp[0], e1 = sp[nrows];
  for (int64_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
    int lo = 0, hi = nrows;  // last i with sp[i] <= e
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (sp[mid] <= e) lo = mid; else hi = mid;
    }
    out[e] = r0 + lo;
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
  hipLaunchKernelGGL(ind2ptr_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                     ind, out, M, E);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_ptr2ind(const int64_t *ptr, int64_t M, int64_t E, int64_t *out,
                             void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || E < 0 || !ptr || (E > 0 && !out)) return TSAMD_ERR_INVALID;
  if (E == 0 || M == 0) return TSAMD_OK;
  hipLaunchKernelGGL(ptr2ind_kernel, dim3((unsigned int)ceil_div(M, kRowsPerBlock)), dim3(256), 0,
                     stream, ptr, out, M);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}
