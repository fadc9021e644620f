// Entry-balanced segmented reduction: out[j] = REDUCE value[perm[i]] over i in [ptr[j], ptr[j+1]).
//
// The segments are the rows / columns of a matrix (SparseTensor.sum/mean/min/max(dim), reference
// torch_sparse/reduce.py -> torch_scatter.segment_csr) or long runs of duplicates (coalesce):
// power-law data has segments with 1e5..1e7 entries, which a thread-per-segment kernel serialises
// (13 ms for the row sums of the 40 M-entry north-star graph).  Here the ENTRIES are split evenly:
//
//   tile kernel   a workgroup owns 2048 consecutive entries (4 waves x 8 windows of 64), stages the
//                 pointer slice that intersects them in LDS (expand.h); every lane finds the
//                 segment of its entry by an LDS binary search; a wave-level segmented inclusive scan
//                 (6 bpermute steps, equal-segment guard) reduces the window, a wave-uniform carry
//                 links the windows.  A lane that holds the last entry of a segment writes the result
//                 -- unless the segment began before the wave's chunk: that piece becomes the chunk's
//                 HEAD record; a segment still open at the end of the chunk leaves a TAIL record.
//   fix-up        one wave per chunk with a head record folds the tail records of the chunks before
//                 it (lanes stride over them, butterfly; fp32 sums fold in fp64) and writes the row.
//
// Deterministic (fixed combine tree), no atomics.  Empty segments give 0 (torch_scatter's
// convention); mean divides by the segment length (floor division for integers).
#include "common.h"
#include "expand.h"

#include <type_traits>

namespace tsamd {
namespace {

constexpr int SR_ADD = 0, SR_MIN = 1, SR_MAX = 2;
constexpr int kSrWindows = 8;
constexpr int kSrChunk = kWave * kSrWindows;         // entries per wave
constexpr int kSrWaves = kExpandTile / kSrChunk;     // waves per workgroup (4)
static_assert(kSrWaves * kSrChunk == kExpandTile, "tile = waves x chunk");

template <typename A, int RED>
__device__ __forceinline__ A sr_combine(A a, A b) {
  if constexpr (RED == SR_ADD) return a + b;
  else if constexpr (RED == SR_MIN) return b < a ? b : a;
  else return b > a ? b : a;
}

template <typename T, typename A>
__device__ __forceinline__ void sr_write(T *__restrict__ out, A v, int64_t cnt, bool mean) {
  if (mean) {
    if constexpr (std::is_integral<A>::value) {  // floor division, as torch_scatter does
      A q = v / (A)cnt;
      if ((v % (A)cnt != 0) && ((v < 0) != (cnt < 0))) --q;
      v = q;
    } else {
      v = v / (A)cnt;
    }
  }
  *out = Traits<T>::from_acc(v);
}

template <typename T, int RED>
__global__ __launch_bounds__(kSrWaves *kWave) void segreduce_tile_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ perm, const int64_t *__restrict__ ptr,
    int64_t nseg, int64_t E, int64_t D, bool mean, T *__restrict__ out,
    typename Traits<T>::acc_t *__restrict__ head_val, typename Traits<T>::acc_t *__restrict__ tail_val,
    int64_t *__restrict__ head_seg, int64_t *__restrict__ tail_seg, int64_t nchunks) {
  using A = typename Traits<T>::acc_t;
  __shared__ int64_t sp[kExpandTile + 1];
  __shared__ int64_t span[2];
  const int64_t e0 = (int64_t)blockIdx.x * kExpandTile;
  const int64_t e1 = e0 + kExpandTile < E ? e0 + kExpandTile : E;
  const int64_t d = blockIdx.y;
  int64_t lo, hi;
  tile_span(ptr, nseg, e0, e1, span, &lo, &hi);
  const int64_t S = hi - lo + 1;
  const bool staged = S <= kExpandTile;
  if (staged) {
    for (int i = threadIdx.x; i <= (int)S; i += blockDim.x) sp[i] = ptr[lo + i];
    __syncthreads();
  }
  const int lane = (int)(threadIdx.x & 63);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t chunk = (int64_t)blockIdx.x * kSrWaves + wave;
  const int64_t c0 = e0 + (int64_t)wave * kSrChunk;
  const int64_t c1 = c0 + kSrChunk < e1 ? c0 + kSrChunk : e1;
  if (c0 >= e1) {
    if (lane == 0 && chunk < nchunks) head_seg[chunk] = tail_seg[chunk] = -1;
    return;
  }

  int64_t cseg = -1;  // wave-uniform carry: the segment still open at the end of the last window
  A cval = A(0);
  int64_t hseg = -1;  // head record of this chunk (set by at most one lane)
  A hval = A(0);
  for (int w = 0; w < kSrWindows; ++w) {
    const int64_t base = c0 + (int64_t)w * kWave;
    if (base >= c1) break;
    const int64_t e = base + lane;
    const bool valid = e < c1;
    int64_t seg = -2 - lane, sstart = 0, send = 0;  // invalid lanes never match a neighbour
    A v = A(0);
    if (valid) {
      if (staged) {
        const int i = segment_of_lds(sp, (int)S, e);
        seg = lo + i;
        sstart = sp[i];
        send = sp[i + 1];
      } else {
        seg = lo + segment_of(ptr + lo, S, e);
        sstart = ptr[seg];
        send = ptr[seg + 1];
      }
      v = Traits<T>::to_acc(value[(perm ? perm[e] : e) * D + d]);
    }
    // segmented inclusive scan: segments are contiguous runs, so "lane - off has my segment"
    // implies every lane in between has it too
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int src = lane >= off ? lane - off : lane;
      const A ov = lane_read(v, src);
      const int64_t os = lane_read(seg, src);
      if (lane >= off && os == seg) v = sr_combine<A, RED>(ov, v);
    }
    if (valid && seg == cseg) v = sr_combine<A, RED>(cval, v);
    const bool is_end = valid && e + 1 == send;
    if (is_end) {
      if (sstart >= c0) {
        sr_write<T, A>(out + seg * D + d, v, send - sstart, mean);
      } else {  // began before this chunk: the fix-up kernel finishes it
        hseg = seg;
        hval = v;
      }
    }
    const int last = (int)((c1 - base < kWave ? c1 - base : kWave) - 1);  // wave-uniform
    const bool last_end = lane_read((int32_t)is_end, last) != 0;
    cseg = last_end ? -1 : lane_read(seg, last);
    cval = lane_read(v, last);
  }
  const unsigned long long hm = __ballot(hseg != -1);
  const int hl = hm ? (int)__builtin_ctzll(hm) : 0;
  const int64_t hs = lane_read(hseg, hl);
  const A hv = lane_read(hval, hl);
  if (lane == 0) {
    head_seg[chunk] = hm ? hs : -1;
    tail_seg[chunk] = cseg;
    head_val[d * nchunks + chunk] = hv;
    tail_val[d * nchunks + chunk] = cval;
  }
}

template <typename T, int RED>
__global__ __launch_bounds__(kSrWaves *kWave) void segreduce_fixup_kernel(
    const int64_t *__restrict__ ptr, int64_t D, bool mean, T *__restrict__ out,
    const typename Traits<T>::acc_t *__restrict__ head_val,
    const typename Traits<T>::acc_t *__restrict__ tail_val, const int64_t *__restrict__ head_seg,
    const int64_t *__restrict__ tail_seg, int64_t nchunks) {
  using A = typename Traits<T>::acc_t;
  // fp32 partial sums of a long segment fold in fp64 (as the SpMM fix-up does)
  using W = typename std::conditional<RED == SR_ADD && std::is_same<A, float>::value, double, A>::type;
  const int lane = (int)(threadIdx.x & 63);
  const int64_t b = (int64_t)blockIdx.x * kSrWaves + (threadIdx.x >> 6);
  if (b >= nchunks) return;
  const int64_t s = head_seg[b];
  if (s < 0) return;
  const int64_t d = blockIdx.y;
  int64_t run = 0;  // chunks b-1, b-2, ... whose open segment is s
  for (;;) {
    const int64_t idx = b - 1 - run - lane;
    const bool ok = idx >= 0 && tail_seg[idx] == s;
    const unsigned long long m = __ballot(ok);
    const int c = m == ~0ull ? 64 : (int)__builtin_ctzll(~m);
    run += c;
    if (c < 64) break;
  }
  const A *tv = tail_val + d * nchunks;
  W acc = W(0);
  bool have = false;
  for (int64_t i = lane; i < run; i += kWave) {
    const W v = (W)tv[b - 1 - i];
    acc = have ? sr_combine<W, RED>(acc, v) : v;
    have = true;
  }
  for (int off = 32; off > 0; off >>= 1) {  // lanes without a record must not contribute
    const W ov = lane_xor(acc, off);
    const bool oh = lane_xor((int32_t)have, off) != 0;
    if (oh) acc = have ? sr_combine<W, RED>(acc, ov) : ov;
    have = have || oh;
  }
  if (lane == 0) {
    const W hv = (W)head_val[d * nchunks + b];
    const W tot = have ? sr_combine<W, RED>(acc, hv) : hv;
    sr_write<T, A>(out + s * D + d, (A)tot, ptr[s + 1] - ptr[s], mean);
  }
}

template <typename T, int RED>
int launch_segreduce(const T *value, const int64_t *perm, const int64_t *ptr, int64_t nseg, int64_t E,
                     int64_t D, bool mean, T *out, void *workspace, hipStream_t stream) {
  using A = typename Traits<T>::acc_t;
  const int64_t nchunks = ceil_div(E, (int64_t)kSrChunk);
  char *p = reinterpret_cast<char *>(workspace);
  A *head_val = reinterpret_cast<A *>(p);
  p += align_up(sizeof(A) * (size_t)(D * nchunks), 256);
  A *tail_val = reinterpret_cast<A *>(p);
  p += align_up(sizeof(A) * (size_t)(D * nchunks), 256);
  int64_t *head_seg = reinterpret_cast<int64_t *>(p);
  p += align_up(sizeof(int64_t) * (size_t)nchunks, 256);
  int64_t *tail_seg = reinterpret_cast<int64_t *>(p);
  const dim3 grid((unsigned int)ceil_div(E, (int64_t)kExpandTile), (unsigned int)D);
  hipLaunchKernelGGL((segreduce_tile_kernel<T, RED>), grid, dim3(kSrWaves * kWave), 0, stream, value, perm, ptr,
                     nseg, E, D, mean, out, head_val, tail_val, head_seg, tail_seg, nchunks);
  TSAMD_LAUNCH_CHECK();
  const dim3 grid2((unsigned int)ceil_div(nchunks, (int64_t)kSrWaves), (unsigned int)D);
  hipLaunchKernelGGL((segreduce_fixup_kernel<T, RED>), grid2, dim3(kSrWaves * kWave), 0, stream, ptr, D, mean,
                     out, (const A *)head_val, (const A *)tail_val, (const int64_t *)head_seg,
                     (const int64_t *)tail_seg, nchunks);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" size_t tsamd_segment_reduce_balanced_workspace_bytes(int dtype, int64_t E, int64_t D) {
  const size_t nchunks = (size_t)ceil_div(E > 0 ? E : 1, (int64_t)kSrChunk);
  const size_t d = (size_t)(D > 0 ? D : 1);
  return 2 * align_up(acc_size(dtype) * d * nchunks, 256) + 2 * align_up(sizeof(int64_t) * nchunks, 256);
}

extern "C" int tsamd_segment_reduce_balanced(int dtype, int reduce, const void *value,
                                             const int64_t *perm, const int64_t *seg_ptr, int64_t nseg,
                                             int64_t E, int64_t D, void *out, void *workspace,
                                             size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (nseg < 0 || E < 0 || D < 0) return TSAMD_ERR_INVALID;
  if (reduce < TSAMD_SUM || reduce > TSAMD_MAX) return TSAMD_ERR_UNSUPPORTED;
  if (dtype_size(dtype) == 0) return TSAMD_ERR_UNSUPPORTED;
  if (D > 65535) return TSAMD_ERR_UNSUPPORTED;
  if (nseg * D == 0) return TSAMD_OK;
  if (!out || !seg_ptr) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(out, 0, dtype_size(dtype) * (size_t)(nseg * D), stream));  // empty segments
  if (E == 0) return TSAMD_OK;
  if (!value) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_segment_reduce_balanced_workspace_bytes(dtype, E, D))
    return TSAMD_ERR_WORKSPACE;
  const bool mean = reduce == TSAMD_MEAN;
  return TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    const scalar_t *v = reinterpret_cast<const scalar_t *>(value);
    scalar_t *o = reinterpret_cast<scalar_t *>(out);
    if (reduce == TSAMD_MIN)
      return launch_segreduce<scalar_t, SR_MIN>(v, perm, seg_ptr, nseg, E, D, mean, o, workspace, stream);
    if (reduce == TSAMD_MAX)
      return launch_segreduce<scalar_t, SR_MAX>(v, perm, seg_ptr, nseg, E, D, mean, o, workspace, stream);
    return launch_segreduce<scalar_t, SR_ADD>(v, perm, seg_ptr, nseg, E, D, mean, o, workspace, stream);
  });
}
