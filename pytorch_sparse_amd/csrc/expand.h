// Expanding a pointer array into per-entry ids, balanced by OUTPUT entries: a workgroup owns a tile
// of kExpandTile consecutive entries, locates the first and last segment that intersect it with two
// binary searches over the global pointer array, stages that slice of the pointers in LDS and lets
// every thread find the segment of its entries there.  A hub segment with millions of entries is
// spread over thousands of workgroups; a segment-per-workgroup mapping leaves it to one.
#pragma once

#include "common.h"

namespace tsamd {

constexpr int kExpandTile = 2048;

// last i in [0, n) with ptr[i] <= e; needs ptr non-decreasing and ptr[0] <= e.  Among equal
// pointers (empty segments) this is the last one, i.e. the segment that really holds entry e.
__device__ inline int64_t segment_of(const int64_t *__restrict__ ptr, int64_t n, int64_t e) {
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (ptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ inline int segment_of_lds(const int64_t *sp, int n, int64_t e) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (sp[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// Block-wide: the segments [lo, hi] that intersect the tile of entries [e0, e1).  `span` is 2
// int64 of LDS; contains a __syncthreads().
__device__ inline void tile_span(const int64_t *__restrict__ ptr, int64_t nseg, int64_t e0, int64_t e1,
                                 int64_t *span, int64_t *lo, int64_t *hi) {
  if (threadIdx.x == 0) span[0] = segment_of(ptr, nseg, e0);
  if (threadIdx.x == 64) span[1] = segment_of(ptr, nseg, e1 - 1);
  __syncthreads();
  *lo = span[0];
  *hi = span[1];
}

}  // namespace tsamd
