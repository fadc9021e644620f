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
#include "sort.h"

#include <type_traits>

namespace tsamd {
namespace {

// counts[0] += #{i : key[i] < key[i-1]}, counts[1] += #{i : key[i] == key[i-1]}
// grid-stride with per-thread counters: one pair of atomics per workgroup at the very end
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
  // one pair of atomics per WORKGROUP: on unsorted input every wave has descents, and the two result words
  // are hot addresses (~12 ns per atomic, serialised: 0.12 ms for 7.5 M entries with per-wave atomics)
  __shared__ unsigned int s_cnt[2][4];
  if ((threadIdx.x & 63) == 0) {
    s_cnt[0][threadIdx.x >> 6] = desc;
    s_cnt[1][threadIdx.x >> 6] = dup;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const unsigned int t = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
    if (t) atomicAdd(&counts[threadIdx.x], (unsigned long long)t);
  }
}

// order probe + range check in one pass: counts[0..1] as above, counts[2] = max row id, counts[3] = max col id
// (counts[2..3] start at 0; ids are assumed non-negative)
constexpr int kCheckThreads = 1024;
__global__ __launch_bounds__(kCheckThreads) void order_check_kernel(const int64_t *__restrict__ row,
                                                         const int64_t *__restrict__ col, int64_t n,
                                                         unsigned long long *counts) {
  unsigned int desc = 0, dup = 0;
  int64_t mr = 0, mc = 0;
  const int lane = (int)(threadIdx.x & 63);
  constexpr int kB = 4;  // entries per thread and step, all loads of a step in flight together
  for (int64_t base = (int64_t)blockIdx.x * (kCheckThreads * kB); base < n; base += (int64_t)gridDim.x * (kCheckThreads * kB)) {
    int64_t r[kB], c[kB], pr0[kB], pc0[kB];  // (lane 0's predecessor: requested with the step's other loads)
#pragma unroll
    for (int u = 0; u < kB; ++u) {
      const int64_t i = base + u * kCheckThreads + threadIdx.x;
      r[u] = i < n ? row[i] : 0;
      c[u] = i < n ? col[i] : 0;
      pr0[u] = 0;
      pc0[u] = 0;
      if (lane == 0 && i < n && i > 0) {
        pr0[u] = row[i - 1];
        pc0[u] = col[i - 1];
      }
    }
#pragma unroll
    for (int u = 0; u < kB; ++u) {
      const int64_t i = base + u * kCheckThreads + threadIdx.x;
      // the entry before: the neighbour lane's registers, except for lane 0 (one cached load per wave)
      int64_t pr = lane_below(r[u]), pc = lane_below(c[u]);  // (DPP: as ds_bpermute these were four LDS-pipe round trips per entry)
      if (i < n) {
        if (lane == 0 && i > 0) {
          pr = pr0[u];
          pc = pc0[u];
        }
        mr = (uint64_t)r[u] > (uint64_t)mr ? r[u] : mr;  // unsigned: a negative id reads as a huge one and fails the range check
        mc = (uint64_t)c[u] > (uint64_t)mc ? c[u] : mc;
        if (i > 0) {  // lexicographic (row, col): the same order as row * ncols + col, without knowing ncols
          desc += (r[u] < pr) || (r[u] == pr && c[u] < pc);
          dup += (r[u] == pr) && (c[u] == pc);
        }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    desc += lane_xor(desc, off);
    dup += lane_xor(dup, off);
    const int64_t orr = lane_xor(mr, off), oc = lane_xor(mc, off);
    mr = (uint64_t)orr > (uint64_t)mr ? orr : mr;
    mc = (uint64_t)oc > (uint64_t)mc ? oc : mc;
  }
  // one set of atomics per WORKGROUP (the four result words are hot addresses: ~12 ns per atomic, serialised --
  // a first version with one set per wave took 0.30 ms for 7.5 M entries instead of 0.07), and the maxima only
  // when they would change anything
  __shared__ unsigned int s_desc[kCheckThreads / 64], s_dup[kCheckThreads / 64];
  __shared__ long long s_mr[kCheckThreads / 64], s_mc[kCheckThreads / 64];
  const int wid = (int)(threadIdx.x >> 6);
  if ((threadIdx.x & 63) == 0) {
    s_desc[wid] = desc;
    s_dup[wid] = dup;
    s_mr[wid] = mr;
    s_mc[wid] = mc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int d = 0, u = 0;
    long long a = 0, b = 0;
    for (int w = 0; w < kCheckThreads / 64; ++w) {
      d += s_desc[w];
      u += s_dup[w];
      a = (unsigned long long)s_mr[w] > (unsigned long long)a ? s_mr[w] : a;
      b = (unsigned long long)s_mc[w] > (unsigned long long)b ? s_mc[w] : b;
    }
    if (d) atomicAdd(&counts[0], (unsigned long long)d);
    if (u) atomicAdd(&counts[1], (unsigned long long)u);
    if ((unsigned long long)a > counts[2]) atomicMax(&counts[2], (unsigned long long)a);
    if ((unsigned long long)b > counts[3]) atomicMax(&counts[3], (unsigned long long)b);
  }
}

__device__ inline bool is_head(const int64_t *row, const int64_t *col, int64_t i) {
  return i == 0 || row[i] != row[i - 1] || col[i] != col[i - 1];
}

// Adjacent-duplicate compaction of a sorted COO list in ONE kernel: head flags, their scan and the compaction.
// A tile of 2048 entries counts its heads, publishes the count and looks back over the tiles before it
// (decoupled look-back, one status word per tile: [63:62] 1 = own count, 2 = inclusive prefix; tiles are handed
// out by a ticket), then writes its heads at their final positions.
constexpr int kCompactItems = 8;
constexpr int kCompactTile = 256 * kCompactItems;

__global__ __launch_bounds__(256) void coalesce_compact_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ col, int64_t n, int64_t *__restrict__ row_out,
    int64_t *__restrict__ col_out, int64_t *__restrict__ seg_ptr, int64_t *__restrict__ nnz_out,
    unsigned long long *__restrict__ state /* [1] error, [8 + tile] status */,
    const unsigned long long *__restrict__ skip /* nullable: non-zero = the outputs are there already (compacting sort) */,
    int64_t *__restrict__ fused_out = nullptr /* nullable: tsamd_sort_coalesce_reduce's counts[3], 0 on this route */) {
  if (skip != nullptr && *skip != 0) return;
  if (fused_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *fused_out = 0;
  __shared__ unsigned long long s_base;
  __shared__ unsigned int s_cnt[kCompactItems][4];
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t tile = (int64_t)blockIdx.x;  // tiles wait for lower tiles only (see sort.hip on the dispatch order)
  const int64_t ntiles = (n + kCompactTile - 1) / kCompactTile;
  const int64_t tile0 = tile * kCompactTile;
  // entry e = tile0 + i * 256 + tid: every load instruction of a wave covers 512 contiguous bytes
  int64_t r[kCompactItems], c[kCompactItems], pr0[kCompactItems], pc0[kCompactItems];
  unsigned long long hmask[kCompactItems];  // the wave's head flags of step i
  unsigned int heads = 0;
#pragma unroll
  for (int i = 0; i < kCompactItems; ++i) {
    const int64_t e = tile0 + i * 256 + tid;
    r[i] = e < n ? row[e] : 0;
    c[i] = e < n ? col[e] : 0;
    pr0[i] = 0;
    pc0[i] = 0;
    if (lane == 0 && e < n && e > 0) {  // (the predecessor of the wave's first entry: in flight with the loads above)
      pr0[i] = row[e - 1];
      pc0[i] = col[e - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < kCompactItems; ++i) {
    const int64_t e = tile0 + i * 256 + tid;
    int64_t pr = lane_below(r[i]), pc = lane_below(c[i]);
    bool h = false;
    if (e < n) {
      if (lane == 0 && e > 0) {
        pr = pr0[i];
        pc = pc0[i];
      }
      h = e == 0 || r[i] != pr || c[i] != pc;
    }
    hmask[i] = __ballot(h);
    heads |= (h ? 1u : 0u) << i;
    if (lane == 0) s_cnt[i][w] = (unsigned int)__popcll(hmask[i]);
  }
  __syncthreads();
  // heads before (step i, wave w) in entry order: steps first, waves inside a step
  unsigned int before_step[kCompactItems];
  unsigned int total = 0;
#pragma unroll
  for (int i = 0; i < kCompactItems; ++i) {
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      if (ww == w) before_step[i] = total;
      total += s_cnt[i][ww];
    }
  }
  constexpr unsigned long long kLocal = 1ull << 62, kPrefix = 2ull << 62, kMask = (1ull << 62) - 1ull;
  if (w == 0) {  // wave 0 publishes and looks back, 64 tiles per step
    unsigned long long *mine = state + 8 + tile;
    if (lane == 0)
      __hip_atomic_store(mine, (tile == 0 ? kPrefix : kLocal) | (unsigned long long)total, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long before = 0;
    int64_t t = tile - 1;
    unsigned int spins = 0;
    while (t >= 0) {
      const int64_t mt = t - lane;
      unsigned long long s = kPrefix;  // lanes past tile 0 read as "prefix 0"
      if (mt >= 0) s = __hip_atomic_load(state + 8 + mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long ready = __ballot((s >> 62) != 0);
      const unsigned long long pref = __ballot((s >> 62) == 2);
      // usable lanes: the ready ones up to and including the first prefix, with no gap before it
      const int first_gap = ~ready ? __builtin_ctzll(~ready) : 64;
      const int first_pref = pref ? __builtin_ctzll(pref) : 64;
      const int take = first_pref < first_gap ? first_pref + 1 : first_gap;
      unsigned long long v = lane < take ? (s & kMask) : 0ull;
      for (int off = 32; off > 0; off >>= 1) v += (unsigned long long)lane_xor((int64_t)v, off);
      before += v;
      if (first_pref < first_gap) break;  // reached an inclusive prefix
      t -= take;
      if (take == 0) {
        if (++spins > (1u << 20)) {
          if (lane == 0) state[1] = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (lane == 0) {
      if (tile > 0)
        __hip_atomic_store(mine, kPrefix | (before + (unsigned long long)total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_base = before;
      if (tile == ntiles - 1) {
        nnz_out[0] = (int64_t)(before + total);
        seg_ptr[before + total] = n;
      }
    }
  }
  __syncthreads();
  const int64_t base = (int64_t)s_base;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < kCompactItems; ++i) {
    if ((heads >> i) & 1u) {
      const int64_t p = base + before_step[i] + (unsigned int)__popcll(hmask[i] & lt);
      row_out[p] = r[i];
      col_out[p] = c[i];
      seg_ptr[p] = tile0 + i * 256 + tid;
    }
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

// ---------------------------------------------------------------------------
// Small inputs: the whole sort_coo in ONE launch.  A COO set of a few thousand entries (a mini-batch
// sub-graph, BASELINE config 1) spends its time in ~16 dependent launches of the general path (probe, keys,
// (histogram, scan, scatter) per digit, decode: ~4 us each on the GPU, a HIP graph replays them no faster).
// Here one 1024-thread workgroup keeps the (32-bit key, 16-bit index) pairs of up to 8192 entries in LDS and
// runs every 8-bit LSD pass there: per-wave match ranking (8 ballots per key, stable), per-wave digit counts,
// one scan over the 256 digits, scatter into the other LDS buffer.  Needs row * N + col < 2^32.
// counts (nullable): [#descents, #adjacent duplicates] of the input (lexicographic); with `auto_mode` an input
// without descents skips the passes (outputs = copy + identity), exactly like tsamd_sort_coo_auto.
// ---------------------------------------------------------------------------
constexpr int kSmallSortThreads = 1024;
constexpr int kSmallSortItems = 8;
constexpr int kSmallSortMax = kSmallSortThreads * kSmallSortItems;  // 8192

__global__ __launch_bounds__(kSmallSortThreads) void small_sort_coo_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ col, int n, uint32_t ncols, int passes,
    int64_t *__restrict__ row_out, int64_t *__restrict__ col_out, int64_t *__restrict__ perm_out,
    unsigned long long *__restrict__ counts, int auto_mode) {
  __shared__ uint32_t kbuf[2][kSmallSortMax];
  __shared__ uint16_t vbuf[2][kSmallSortMax];
  __shared__ uint32_t cnt[kSmallSortThreads / 64][256];
  __shared__ uint32_t dig_off[256];
  __shared__ uint32_t wsum[4];
  __shared__ unsigned int s_order[2];
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 2) s_order[tid] = 0;
  __syncthreads();
  // load, build keys, probe the order
  unsigned int desc = 0, dup = 0;
#pragma unroll
  for (int j = 0; j < kSmallSortItems; ++j) {
    const int i = w * (64 * kSmallSortItems) + j * 64 + lane;
    if (i < n) {
      const int64_t r = row[i], c = col[i];
      kbuf[0][i] = (uint32_t)((uint64_t)r * ncols + (uint64_t)c);
      vbuf[0][i] = (uint16_t)i;
      if (i > 0) {
        const int64_t pr = row[i - 1], pc = col[i - 1];
        desc += (r < pr) || (r == pr && c < pc);
        dup += (r == pr) && (c == pc);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    desc += lane_xor(desc, off);
    dup += lane_xor(dup, off);
  }
  if (lane == 0) {
    if (desc) atomicAdd(&s_order[0], desc);
    if (dup) atomicAdd(&s_order[1], dup);
  }
  __syncthreads();
  if (counts != nullptr && tid < 2) counts[tid] = s_order[tid];
  const bool skip = auto_mode != 0 && s_order[0] == 0;
  int cur = 0;
  for (int pass = 0; pass < passes && !skip; ++pass) {
    const int shift = pass * 8;
    const uint32_t *ks = kbuf[cur];
    const uint16_t *vs = vbuf[cur];
#pragma unroll
    for (int q = 0; q < 4; ++q) cnt[w][q * 64 + lane] = 0;  // this wave's counters (wave-local: no barrier)
    uint32_t key[kSmallSortItems], lrank[kSmallSortItems];
    uint16_t val[kSmallSortItems];
#pragma unroll
    for (int j = 0; j < kSmallSortItems; ++j) {
      const int i = w * (64 * kSmallSortItems) + j * 64 + lane;
      const bool valid = i < n;
      key[j] = valid ? ks[i] : 0u;
      val[j] = valid ? vs[i] : (uint16_t)0;
      const uint32_t d = (key[j] >> shift) & 255u;
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
      }
      const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
      const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
      uint32_t pre = 0;
      if (valid && lane == leader) {
        pre = cnt[w][d];
        cnt[w][d] = pre + (uint32_t)__popcll(peers);
      }
      pre = lane_read(pre, leader);
      lrank[j] = pre + rank;
    }
    __syncthreads();
    if (tid < 256) {  // thread t owns digit t: exclusive prefix over the waves, then over the digits
      uint32_t run = 0;
#pragma unroll
      for (int ww = 0; ww < kSmallSortThreads / 64; ++ww) {
        const uint32_t c = cnt[ww][tid];
        cnt[ww][tid] = run;
        run += c;
      }
      uint32_t inc = run;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = lane_read(inc, lane >= off ? lane - off : lane);
        if (lane >= off) inc += o;
      }
      if (lane == 63) wsum[w] = inc;
      dig_off[tid] = inc - run;  // exclusive inside the wave; the wave bases are added below
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t base = 0;
      for (int ww = 0; ww < w; ++ww) base += wsum[ww];
      dig_off[tid] += base;
    }
    __syncthreads();
    uint32_t *kd = kbuf[cur ^ 1];
    uint16_t *vd = vbuf[cur ^ 1];
#pragma unroll
    for (int j = 0; j < kSmallSortItems; ++j) {
      const int i = w * (64 * kSmallSortItems) + j * 64 + lane;
      if (i < n) {
        const uint32_t d = (key[j] >> shift) & 255u;
        const uint32_t pos = dig_off[d] + cnt[w][d] + lrank[j];
        kd[pos] = key[j];
        vd[pos] = val[j];
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int i = tid; i < n; i += kSmallSortThreads) {
    if (skip) {
      if (row_out) row_out[i] = row[i];
      if (col_out) col_out[i] = col[i];
      perm_out[i] = i;
    } else {
      const uint32_t k = kbuf[cur][i];
      const uint32_t r = k / ncols;
      if (row_out) row_out[i] = (int64_t)r;
      if (col_out) col_out[i] = (int64_t)(k - r * ncols);
      perm_out[i] = (int64_t)vbuf[cur][i];
    }
  }
}

// true when the one-launch path applies (and was launched)
bool small_sort_coo(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N, int64_t *row_out,
                    int64_t *col_out, int64_t *perm_out, int64_t *counts, bool auto_mode, hipStream_t stream) {
  if (E > kSmallSortMax || N <= 0 || N >= ((int64_t)1 << 32)) return false;
  const int bits = key_bits_for(M, N);
  if (bits > 32) return false;
  const int passes = E > 1 ? (bits + 7) / 8 : 0;
  hipLaunchKernelGGL(small_sort_coo_kernel, dim3(1), dim3(kSmallSortThreads), 0, stream, row, col, (int)E,
                     (uint32_t)N, passes, row_out, col_out, perm_out,
                     reinterpret_cast<unsigned long long *>(counts), auto_mode ? 1 : 0);
  return true;
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" size_t tsamd_sort_coo_workspace_bytes(int64_t E) { return sort_coo_workspace_bytes(E); }

extern "C" int tsamd_sort_rank_mode(int set) {
  if (set == 0 || set == 1) sort_set_rank_mode(set);
  else if (set == 2) sort_set_rank_mode(-1);
  return sort_rank_mode(nullptr);
}

extern "C" int tsamd_sort_coo(const int64_t *row, const int64_t *col, int64_t E, int64_t M,
                              int64_t N, int64_t *row_out, int64_t *col_out, int64_t *perm_out,
                              void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0) return TSAMD_ERR_INVALID;
  if (E == 0) return TSAMD_OK;
  if (!row || !col || !perm_out) return TSAMD_ERR_INVALID;
  if (!sort_coo_supported(E, M, N)) return TSAMD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < tsamd_sort_coo_workspace_bytes(E)) return TSAMD_ERR_WORKSPACE;
  if (small_sort_coo(row, col, E, M, N, row_out, col_out, perm_out, nullptr, false, stream)) {
    TSAMD_LAUNCH_CHECK();
    return TSAMD_OK;
  }
  return sort_coo_onesweep(row, col, E, M, N, row_out, col_out, perm_out, nullptr, false, nullptr, workspace, stream);
}

// sort_coo decided on the device: counts_out[0..1] = (#descents, #adjacent duplicates) of the INPUT; when
// there is no descent the radix passes return at once and the outputs are a copy + the identity.
static int sort_coo_auto_impl(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                              int64_t *row_out, int64_t *col_out, int64_t *perm_out, int64_t *counts_out,
                              bool probe, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0 || !counts_out) return TSAMD_ERR_INVALID;
  if (E == 0) {
    if (probe) TSAMD_HIP_TRY(hipMemsetAsync(counts_out, 0, 2 * sizeof(int64_t), stream));
    return TSAMD_OK;
  }
  if (!row || !col || !perm_out) return TSAMD_ERR_INVALID;
  if (!sort_coo_supported(E, M, N)) return TSAMD_ERR_UNSUPPORTED;
  if (small_sort_coo(row, col, E, M, N, row_out, col_out, perm_out, probe ? counts_out : nullptr, true, stream)) {
    TSAMD_LAUNCH_CHECK();  // probe, sort and decode in one launch (the small kernel probes for itself)
    return TSAMD_OK;
  }
  if (!workspace || workspace_bytes < tsamd_sort_coo_workspace_bytes(E)) return TSAMD_ERR_WORKSPACE;
  // probe = true: the build kernel counts the descents itself (one read of the input serves the probe, the keys and
  // the digit histograms); probe = false: counts_out[0] already holds them (tsamd_coo_check)
  return sort_coo_onesweep(row, col, E, M, N, row_out, col_out, perm_out, probe ? nullptr : counts_out, probe,
                           probe ? counts_out : nullptr, workspace, stream);
}

extern "C" int tsamd_sort_coo_auto(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                                   int64_t *row_out, int64_t *col_out, int64_t *perm_out,
                                   int64_t *counts_out, void *workspace, size_t workspace_bytes,
                                   void *stream_) {
  return sort_coo_auto_impl(row, col, E, M, N, row_out, col_out, perm_out, counts_out, true, workspace,
                            workspace_bytes, stream_);
}

// The same with the order already probed: descents[0] (device) = #descents of the input, e.g. counts[0] of
// tsamd_coo_check -- the constructor enqueues check, sort and gathers back to back and reads the check's
// result once everything is in flight.
extern "C" int tsamd_sort_coo_probed(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                                     int64_t *row_out, int64_t *col_out, int64_t *perm_out,
                                     const int64_t *descents, void *workspace, size_t workspace_bytes,
                                     void *stream_) {
  return sort_coo_auto_impl(row, col, E, M, N, row_out, col_out, perm_out, const_cast<int64_t *>(descents), false,
                            workspace, workspace_bytes, stream_);
}

// One entry point for the three flavours with the entries' values riding along (include/tsamd.h).
extern "C" int tsamd_sort_coo_values(int mode, const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                                     int64_t *row_out, int64_t *col_out, int64_t *perm_out, int64_t *counts,
                                     const void *value, void *value_out, int64_t value_bytes, void *workspace,
                                     size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0 || mode < 0 || mode > 3 || (mode != 0 && !counts)) return TSAMD_ERR_INVALID;
  if ((value == nullptr) != (value_out == nullptr)) return TSAMD_ERR_INVALID;
  if (value != nullptr && value_bytes != 4 && value_bytes != 8) return TSAMD_ERR_UNSUPPORTED;
  if (E == 0) {
    if (mode == 1 || mode == 3) TSAMD_HIP_TRY(hipMemsetAsync(counts, 0, (mode == 3 ? 4 : 2) * sizeof(int64_t), stream));
    return TSAMD_OK;
  }
  if (!row || !col || !perm_out) return TSAMD_ERR_INVALID;
  if (!sort_coo_supported(E, M, N)) return TSAMD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < tsamd_sort_coo_workspace_bytes(E)) return TSAMD_ERR_WORKSPACE;
  if (mode == 3) {
    // the constructor's range check (max row / col id) rides in the sort's build pass -- except on the one-launch
    // path and for keys of zero bits, where the check is its own (tiny) launch and the sort runs "probed"
    if (E <= kSmallSortMax || key_bits_for(M, N) == 0 || (M <= 1 && N <= 1)) {
      int st = tsamd_coo_check(row, col, E, counts, stream_);
      if (st != TSAMD_OK) return st;
      mode = 2;
    } else {
      return sort_coo_onesweep(row, col, E, M, N, row_out, col_out, perm_out, nullptr, true, counts, workspace, stream,
                               value, value_out, (int)value_bytes, true);
    }
  }
  if (small_sort_coo(row, col, E, M, N, row_out, col_out, perm_out, mode == 1 ? counts : nullptr, mode != 0, stream)) {
    TSAMD_LAUNCH_CHECK();
    if (value != nullptr)  // the one-launch path has no payload: a gather through the permutation behind it
      return tsamd_gather_rows(value, perm_out, value_out, E, E, value_bytes, stream_);
    return TSAMD_OK;
  }
  return sort_coo_onesweep(row, col, E, M, N, row_out, col_out, perm_out, mode == 2 ? counts : nullptr, mode == 1,
                           mode == 1 ? counts : nullptr, workspace, stream, value, value_out, (int)value_bytes);
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
  // few, big workgroups: the four result words are hot addresses (~12 ns per atomic, serialised)
  const int64_t nblk = ceil_div(E, kCheckThreads * 4);
  hipLaunchKernelGGL(order_check_kernel, dim3((unsigned int)(nblk < 512 ? nblk : 512)), dim3(kCheckThreads), 0,
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
  hipLaunchKernelGGL(order_probe_kernel, dim3((unsigned int)(nblk < 1024 ? nblk : 1024)), dim3(256), 0,
                     stream, row, col, E, N, reinterpret_cast<unsigned long long *>(counts_out));
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_coalesce_workspace_bytes(int64_t E) {
  const size_t ntiles = ((size_t)(E > 0 ? E : 1) + kCompactTile - 1) / kCompactTile;
  return align_up(sizeof(unsigned long long) * (8 + ntiles), 256);
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
  const size_t need = tsamd_coalesce_workspace_bytes(E);
  if (!workspace || workspace_bytes < need) return TSAMD_ERR_WORKSPACE;
  TSAMD_HIP_TRY(hipMemsetAsync(workspace, 0, need, stream));
  hipLaunchKernelGGL(coalesce_compact_kernel, dim3((unsigned int)ceil_div(E, kCompactTile)), dim3(256), 0, stream,
                     row, col, E, row_out, col_out, seg_ptr, nnz_out,
                     reinterpret_cast<unsigned long long *>(workspace), (const unsigned long long *)nullptr);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

// Sort + duplicate compaction in one go (include/tsamd.h): the functional coalesce / transpose.
extern "C" size_t tsamd_sort_coalesce_workspace_bytes(int64_t E) {
  return align_up(sort_coo_workspace_bytes(E), 256) + align_up(sizeof(unsigned long long) * kSortCoalesceStatusWords, 256) +
         tsamd_coalesce_workspace_bytes(E);
}

namespace {
// reduce < 0: tsamd_sort_coalesce (counts has 3 entries); else tsamd_sort_coalesce_reduce (4 entries, value_u)
int sort_coalesce_impl(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                       int64_t *row_tmp, int64_t *col_tmp, int64_t *row_u, int64_t *col_u, int64_t *seg_ptr,
                       int64_t *counts, const void *value, void *value_out, int64_t value_bytes, void *value_u,
                       int reduce, int is_float, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (E < 0 || M < 0 || N < 0 || !counts || !seg_ptr) return TSAMD_ERR_INVALID;
  if ((value == nullptr) != (value_out == nullptr)) return TSAMD_ERR_INVALID;
  if (value != nullptr && value_bytes != 4 && value_bytes != 8) return TSAMD_ERR_UNSUPPORTED;
  const int ncounts = reduce >= 0 ? 4 : 3;
  if (E == 0) {
    TSAMD_HIP_TRY(hipMemsetAsync(counts, 0, ncounts * sizeof(int64_t), stream));
    TSAMD_HIP_TRY(hipMemsetAsync(seg_ptr, 0, sizeof(int64_t), stream));
    return TSAMD_OK;
  }
  if (!row || !col || !row_tmp || !col_tmp || !row_u || !col_u) return TSAMD_ERR_INVALID;
  if (!sort_coo_supported(E, M, N)) return TSAMD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < tsamd_sort_coalesce_workspace_bytes(E)) return TSAMD_ERR_WORKSPACE;
  // [status words of the compacting bucket sort | state of the compaction kernel | the sort's workspace]: the two
  // zero-initialised pieces sit directly in front of the sort's own, so that ONE fill covers all three
  char *wsp = reinterpret_cast<char *>(workspace);
  unsigned long long *status = reinterpret_cast<unsigned long long *>(wsp);
  void *co_ws = wsp + align_up(sizeof(unsigned long long) * kSortCoalesceStatusWords, 256);
  const size_t co_bytes = tsamd_coalesce_workspace_bytes(E);
  const size_t pre_zero = align_up(sizeof(unsigned long long) * kSortCoalesceStatusWords, 256) + co_bytes;
  void *sort_ws = wsp + pre_zero;
  const unsigned long long *skip = nullptr;
  if (E <= kSmallSortMax && small_sort_coo(row, col, E, M, N, row_tmp, col_tmp, seg_ptr /* perm: scratch, rewritten below */,
                                           counts, true, stream)) {
    TSAMD_LAUNCH_CHECK();
    if (value != nullptr) {
      int st = tsamd_gather_rows(value, seg_ptr, value_out, E, E, value_bytes, stream_);
      if (st != TSAMD_OK) return st;
    }
    TSAMD_HIP_TRY(hipMemsetAsync(co_ws, 0, co_bytes, stream));
  } else {
    SortCoalesce co{row_u, col_u, seg_ptr, counts + 2, status};
    co.pre_zero_bytes = pre_zero;
#if defined(TSAMD_EXP_COAL_SEPARATE_FILLS)  // A/B builds (scripts/variants.py): a fill per piece, as before
    co.pre_zero_bytes = 0;
    TSAMD_HIP_TRY(hipMemsetAsync(co_ws, 0, co_bytes, stream));
#endif
    if (reduce >= 0) co.fused_out = counts + 3;
    co.no_seg = reduce >= 0 && value == nullptr;  // tsamd_sort_coalesce_reduce without a value: index only
    if (reduce >= 0 && value != nullptr && value_bytes == 4 && value_u != nullptr) {
      co.value_u = value_u;
      co.reduce = reduce;
      co.is_float = is_float;
    }
    int st = sort_coo_onesweep(row, col, E, M, N, row_tmp, col_tmp, nullptr, nullptr, true, counts, sort_ws, stream, value,
                               value_out, (int)value_bytes, false, &co);
    if (st != TSAMD_OK) return st;
    skip = sort_fast_flag(sort_ws, E);
  }
  // the one-sweep passes (or the one-launch sort) left sorted pairs in row_tmp / col_tmp: compact them -- returns at
  // once when the bucket path wrote the compacted outputs itself
  hipLaunchKernelGGL(coalesce_compact_kernel, dim3((unsigned int)ceil_div(E, kCompactTile)), dim3(256), 0, stream, row_tmp,
                     col_tmp, E, row_u, col_u, seg_ptr, counts + 2, reinterpret_cast<unsigned long long *>(co_ws), skip,
                     reduce >= 0 ? counts + 3 : (int64_t *)nullptr);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}
}  // namespace

extern "C" int tsamd_sort_coalesce(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                                   int64_t *row_tmp, int64_t *col_tmp, int64_t *row_u, int64_t *col_u, int64_t *seg_ptr,
                                   int64_t *counts, const void *value, void *value_out, int64_t value_bytes,
                                   void *workspace, size_t workspace_bytes, void *stream_) {
  return sort_coalesce_impl(row, col, E, M, N, row_tmp, col_tmp, row_u, col_u, seg_ptr, counts, value, value_out,
                            value_bytes, nullptr, -1, 1, workspace, workspace_bytes, stream_);
}

extern "C" int tsamd_sort_coalesce_reduce(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N,
                                          int64_t *row_tmp, int64_t *col_tmp, int64_t *row_u, int64_t *col_u,
                                          int64_t *seg_ptr, int64_t *counts, int dtype, int reduce, const void *value,
                                          void *value_out, void *value_u, void *workspace, size_t workspace_bytes,
                                          void *stream_) {
  if (reduce < TSAMD_SUM || reduce > TSAMD_MAX) return TSAMD_ERR_UNSUPPORTED;
  if (dtype != TSAMD_F32 && dtype != TSAMD_I32) return TSAMD_ERR_UNSUPPORTED;
  // value, value_out, value_u: all three or none (none = index only: dtype / reduce are not looked at beyond the checks
  // above, and the bucket route writes no seg_ptr -- nobody is going to reduce values by the run starts)
  if ((value == nullptr) != (value_out == nullptr) || (value == nullptr) != (value_u == nullptr)) return TSAMD_ERR_INVALID;
  return sort_coalesce_impl(row, col, E, M, N, row_tmp, col_tmp, row_u, col_u, seg_ptr, counts, value, value_out, 4,
                            value_u, reduce, dtype == TSAMD_F32 ? 1 : 0, workspace, workspace_bytes, stream_);
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
