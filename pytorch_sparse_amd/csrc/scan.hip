// Device-wide exclusive scan (see scan.h).
#include "scan.h"

namespace tsamd {
namespace {

__device__ inline void load_items(const int64_t *__restrict__ in, int64_t base, int64_t n,
                                  int64_t (&v)[kScanItems]) {
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const int64_t idx = base + i;
    v[i] = idx < n ? in[idx] : 0;
  }
}

__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const int64_t *__restrict__ in,
                                                                  int64_t *__restrict__ sums,
                                                                  int64_t n) {
  __shared__ int64_t smem[8];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t v[kScanItems];
  load_items(in, base, n, v);
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) s += v[i];
  int64_t tot;
  block_exclusive_scan_256(s, smem, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// offsets == nullptr: single tile (offset 0); total != nullptr: block 0 writes the grand total
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(const int64_t *__restrict__ in,
                                                                 int64_t *__restrict__ out,
                                                                 const int64_t *__restrict__ offsets,
                                                                 int64_t *__restrict__ total,
                                                                 int64_t n) {
  __shared__ int64_t smem[8];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t v[kScanItems];
  load_items(in, base, n, v);
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) s += v[i];
  int64_t tot;
  int64_t run = block_exclusive_scan_256(s, smem, &tot) + (offsets ? offsets[blockIdx.x] : 0);
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const int64_t idx = base + i;
    if (idx < n) out[idx] = run;
    run += v[i];
  }
  if (total != nullptr && offsets == nullptr && threadIdx.x == 0 && blockIdx.x == 0) *total = tot;
}

}  // namespace

size_t scan_workspace_bytes(int64_t n) {
  size_t bytes = 0;
  while (n > kScanTile) {
    n = ceil_div(n, kScanTile);
    bytes += align_up(sizeof(int64_t) * (size_t)n, 256);
  }
  return bytes + 256;
}

int exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, int64_t *total, void *workspace,
                       hipStream_t stream) {
  if (n <= 0) {
    if (total) TSAMD_HIP_TRY(hipMemsetAsync(total, 0, sizeof(int64_t), stream));
    return TSAMD_OK;
  }
  if (n <= kScanTile) {
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kScanThreads), 0, stream, in, out,
                       (const int64_t *)nullptr, total, n);
    TSAMD_LAUNCH_CHECK();
    return TSAMD_OK;
  }
  const int64_t nb = ceil_div(n, kScanTile);
  int64_t *sums = reinterpret_cast<int64_t *>(workspace);
  char *next = reinterpret_cast<char *>(workspace) + align_up(sizeof(int64_t) * (size_t)nb, 256);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned int)nb), dim3(kScanThreads), 0, stream, in,
                     sums, n);
  TSAMD_LAUNCH_CHECK();
  int st = exclusive_scan_i64(sums, sums, nb, total, next, stream);
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned int)nb), dim3(kScanThreads), 0, stream, in,
                     out, (const int64_t *)sums, (int64_t *)nullptr, n);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

}  // namespace tsamd
