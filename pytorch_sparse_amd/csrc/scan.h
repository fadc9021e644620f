// Device-wide exclusive scan of int64 values (three small kernels, recursive on the
// block sums).  Building block of the radix sort, coalesce and SpSpMM.
#pragma once

#include "common.h"

namespace tsamd {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // elements per workgroup

// Bytes of scratch an exclusive scan over n elements needs (block sums of every level).
size_t scan_workspace_bytes(int64_t n);

// out[i] = sum_{j<i} in[j]  (in == out allowed).  If total != nullptr, *total (device) = sum of
// all elements.  `workspace` must hold scan_workspace_bytes(n) bytes.
int exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, int64_t *total, void *workspace,
                       hipStream_t stream);

// wave64 inclusive scan of one value per lane (DPP-free: bpermute shifts)
__device__ inline int64_t wave_inclusive_scan(int64_t v) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t o = lane_read(v, lane >= off ? lane - off : lane);
    if (lane >= off) v += o;
  }
  return v;
}
__device__ inline uint32_t wave_inclusive_scan_u32(uint32_t v) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = lane_read(v, lane >= off ? lane - off : lane);
    if (lane >= off) v += o;
  }
  return v;
}

// Exclusive scan of one int64 per thread across a 256-thread block.  `smem` needs 8 int64.
// Returns the exclusive prefix; *block_total receives the block sum (all threads).
__device__ inline int64_t block_exclusive_scan_256(int64_t v, int64_t *smem, int64_t *block_total) {
  const int lane = (int)(threadIdx.x & 63);
  const int wid = (int)(threadIdx.x >> 6);
  const int64_t inc = wave_inclusive_scan(v);
  if (lane == 63) smem[wid] = inc;
  __syncthreads();
  int64_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int64_t s = smem[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *block_total = tot;
  return base + inc - v;
}

}  // namespace tsamd
