// Stable LSD radix sort of (int64 key, int64 payload) pairs on gfx950.
//
// Replaces the generic device sort the reference calls through `index_sort` /
// `torch.sort` when a SparseStorage is built from unsorted COO, when csr2csc is
// computed and inside coalesce/transpose (torch_sparse/utils.py:14-21, called from
// torch_sparse/storage.py:149-162, 407-429).  Keys are `row * N + col`, so only
// ceil(log2(M*N)) bits are significant: the number of 8-bit passes is chosen per call.
// Unlike the reference's default `torch.sort` the order of equal keys is stable.
//
// One pass = three launches:
//   radix_hist_kernel     per-workgroup digit histogram          -> hist[digit][block]
//   exclusive_scan_i64    over the digit-major histogram matrix  (scan.hip)
//   radix_scatter_kernel  wave-level match ranking (8 ballots per key) keeps equal digits in
//                         input order; the tile is reordered in LDS so that every digit run
//                         leaves the workgroup as one contiguous, coalesced write.
#include "common.h"
#include "scan.h"

namespace tsamd {
namespace {

constexpr int kSortThreads = 256;
#ifndef TSAMD_SORT_ITEMS
#define TSAMD_SORT_ITEMS 16
#endif
constexpr int kSortItems = TSAMD_SORT_ITEMS;
constexpr int kSortTile = kSortThreads * kSortItems;  // 4096 pairs per workgroup (A/B: 4 / 8 / 12 / 16 items -> 0.83 / 0.61 / 0.58 / 0.56 ms for 7.5 M pairs)
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

// `todo` (may be NULL): device word that says whether there is anything to sort -- the number of descents
// of the key sequence (tsamd_sort_coo_auto).  Zero = already sorted: the pass kernels return at once.
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const int64_t *__restrict__ keys,
                                                                 int64_t n, int shift,
                                                                 int64_t *__restrict__ hist,
                                                                 int64_t nb, const int64_t *__restrict__ todo) {
  if (todo != nullptr && *todo == 0) return;
  __shared__ uint32_t cnt[kRadix];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int64_t idx = base + i * kSortThreads + threadIdx.x;
    if (idx < n) atomicAdd(&cnt[(uint32_t)((uint64_t)keys[idx] >> shift) & (kRadix - 1)], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nb + blockIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
    const int64_t *__restrict__ keys_in, const int64_t *__restrict__ vals_in,
    int64_t *__restrict__ keys_out, int64_t *__restrict__ vals_out, int64_t n, int shift,
    const int64_t *__restrict__ hist_scanned, int64_t nb, const int64_t *__restrict__ todo) {
  if (todo != nullptr && *todo == 0) return;
  // gfx950 only: the tile lives in LDS (160 KB per CU there, 64 KB on older parts)
  static_assert(sizeof(int64_t) * 2 * kSortTile + sizeof(uint32_t) * 5 * kRadix + sizeof(int64_t) * (kRadix + 8) <=
                    160 * 1024,
                "radix_scatter_kernel: the tile (TSAMD_SORT_ITEMS) no longer fits the 160 KB LDS of gfx950");
  __shared__ int64_t skey[kSortTile];
  __shared__ int64_t sval[kSortTile];
  __shared__ uint32_t cnt[4][kRadix];
  __shared__ uint32_t dig_off[kRadix];
  __shared__ int64_t goff[kRadix];
  __shared__ int64_t sscan[8];

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int64_t tile0 = (int64_t)blockIdx.x * kSortTile;
  const int64_t base = tile0 + (int64_t)w * (64 * kSortItems);

  int64_t key[kSortItems], val[kSortItems];
  uint32_t dig[kSortItems], lrank[kSortItems];
  bool valid[kSortItems];
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int64_t idx = base + i * 64 + lane;
    valid[i] = idx < n;
    key[i] = valid[i] ? keys_in[idx] : 0;
    val[i] = valid[i] ? (vals_in ? vals_in[idx] : idx) : 0;
    dig[i] = (uint32_t)((uint64_t)key[i] >> shift) & (kRadix - 1);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) cnt[i][tid] = 0;
  __syncthreads();

  // rank of every key among the equal digits of its wave, in input order
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    unsigned long long peers = __ballot(valid[i]);
#pragma unroll
    for (int b = 0; b < kRadixBits; ++b) {
      const bool bit = (dig[i] >> b) & 1u;
      const unsigned long long m = __ballot(valid[i] && bit);
      peers &= bit ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t rank = (uint32_t)__popcll(peers & lt);
    const int leader = valid[i] ? (__ffsll((long long)peers) - 1) : lane;
    uint32_t pre = 0;
    if (valid[i] && lane == leader) {
      pre = cnt[w][dig[i]];
      cnt[w][dig[i]] = pre + (uint32_t)__popcll(peers);
    }
    pre = lane_read(pre, leader);
    lrank[i] = pre + rank;
  }
  __syncthreads();

  // thread t owns digit t: exclusive prefix over the 4 waves, then over the digits
  {
    uint32_t run = 0;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const uint32_t c = cnt[ww][tid];
      cnt[ww][tid] = run;
      run += c;
    }
    int64_t tot;
    const int64_t ex = block_exclusive_scan_256((int64_t)run, sscan, &tot);
    dig_off[tid] = (uint32_t)ex;
    goff[tid] = hist_scanned[(int64_t)tid * nb + blockIdx.x] - ex;
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    if (valid[i]) {
      const uint32_t pos = dig_off[dig[i]] + cnt[w][dig[i]] + lrank[i];
      skey[pos] = key[i];
      sval[pos] = val[i];
    }
  }
  __syncthreads();

  const int64_t rem = n - tile0;
  const int count = rem < kSortTile ? (int)rem : kSortTile;
  for (int j = tid; j < count; j += kSortThreads) {
    const int64_t k = skey[j];
    const uint32_t d = (uint32_t)((uint64_t)k >> shift) & (kRadix - 1);
    const int64_t o = goff[d] + j;
    keys_out[o] = k;
    vals_out[o] = sval[j];
  }
}

__global__ void copy_iota_kernel(const int64_t *__restrict__ keys_in,
                                 const int64_t *__restrict__ vals_in,
                                 int64_t *__restrict__ keys_out, int64_t *__restrict__ vals_out,
                                 int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys_out[i] = keys_in[i];
  vals_out[i] = vals_in ? vals_in[i] : i;
}

}  // namespace

size_t sort_pairs_workspace_bytes(int64_t n) {
  const int64_t nb = ceil_div(n > 0 ? n : 1, kSortTile);
  const size_t hist = align_up(sizeof(int64_t) * (size_t)(kRadix * nb), 256);
  return 2 * align_up(sizeof(int64_t) * (size_t)(n > 0 ? n : 1), 256) + hist +
         scan_workspace_bytes(kRadix * nb);
}

int sort_pairs(const int64_t *keys_in, const int64_t *vals_in, int64_t *keys_out,
               int64_t *vals_out, int64_t n, int key_bits, void *workspace, hipStream_t stream,
               const int64_t *todo) {
  if (n <= 0) return TSAMD_OK;
  if (key_bits < 0) key_bits = 0;
  if (key_bits > 63) key_bits = 63;
  const int passes = n > 1 ? (key_bits + kRadixBits - 1) / kRadixBits : 0;
  if (passes == 0) {
    hipLaunchKernelGGL(copy_iota_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                       keys_in, vals_in, keys_out, vals_out, n);
    TSAMD_LAUNCH_CHECK();
    return TSAMD_OK;
  }
  const int64_t nb = ceil_div(n, kSortTile);
  char *p = reinterpret_cast<char *>(workspace);
  int64_t *tkeys = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)n, 256);
  int64_t *tvals = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)n, 256);
  int64_t *hist = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)(kRadix * nb), 256);
  void *scan_ws = p;

  const int64_t *src_k = keys_in, *src_v = vals_in;
  for (int pass = 0; pass < passes; ++pass) {
    const bool to_out = ((passes - 1 - pass) % 2) == 0;
    int64_t *dst_k = to_out ? keys_out : tkeys;
    int64_t *dst_v = to_out ? vals_out : tvals;
    const int shift = pass * kRadixBits;
    hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned int)nb), dim3(kSortThreads), 0, stream,
                       src_k, n, shift, hist, nb, todo);
    TSAMD_LAUNCH_CHECK();
    int st = exclusive_scan_i64(hist, hist, kRadix * nb, nullptr, scan_ws, stream);
    if (st != TSAMD_OK) return st;
    hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned int)nb), dim3(kSortThreads), 0, stream,
                       src_k, src_v, dst_k, dst_v, n, shift, (const int64_t *)hist, nb, todo);
    TSAMD_LAUNCH_CHECK();
    src_k = dst_k;
    src_v = dst_v;
  }
  return TSAMD_OK;
}

}  // namespace tsamd
