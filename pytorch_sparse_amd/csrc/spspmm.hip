// SpSpMM  C = A * B  (CSR x CSR -> CSR, sum) on gfx950: row-wise expand / sort / compress.
//
// The reference has no native code for this: torch_sparse/matmul.py:94-111 converts both
// operands to torch sparse COO and calls torch.sparse.mm (PyTorch's CPU SpGEMM / hipSPARSE),
// then trusts the result to be row-major sorted and coalesced.  This file produces exactly
// that result shape: every row of C sorted by column, duplicates summed, explicit zeros kept.
//
// Pipeline (stages are separate C-ABI calls because the host allocates between them):
//   plan     wave per row of A: products(i) = sum_{k in A_i} |B_k|; rows are binned
//            (small <= 512 products, medium <= 4096, large) and an exclusive scan gives
//            every row a slot of `products(i)` entries in a temporary (col, val) buffer.
//   rows     small/medium rows: ONE workgroup (64 / 256 threads) expands the row's products
//            straight into LDS (they never touch HBM), sorts them by column in LDS (one wave: stable
//            radix sort with ballot ranking; 256 threads: bitonic),
//            sums equal columns and writes the compressed row into its slot.
//   large    rows whose products do not fit LDS are expanded to HBM and go through the
//            global radix sort + coalesce + segmented sum (sort.hip / coalesce.hip).
//   compact  exclusive scan of the per-row counts -> rowptrC; slots are copied to their
//            final, dense position.
#include "common.h"
#include "scan.h"

#include <type_traits>

extern "C" int tsamd_sort_coo(const int64_t *, const int64_t *, int64_t, int64_t, int64_t,
                              int64_t *, int64_t *, int64_t *, void *, size_t, void *);
extern "C" size_t tsamd_sort_coo_workspace_bytes(int64_t);
extern "C" int tsamd_coalesce_index(const int64_t *, const int64_t *, int64_t, int64_t *,
                                    int64_t *, int64_t *, int64_t *, void *, size_t, void *);
extern "C" size_t tsamd_coalesce_workspace_bytes(int64_t);

namespace tsamd {
namespace {

constexpr int kSmallCap = 512;    // products handled by one wave (LDS radix sort, no workgroup barriers)
constexpr int kMediumCap = 4096;  // products handled by one 256-thread workgroup

// stats layout (device int64[8])
enum { ST_P = 0, ST_NSMALL = 1, ST_NMEDIUM = 2, ST_NLARGE = 3, ST_PLARGE = 4, ST_NNZC = 5 };

__global__ __launch_bounds__(256) void spspmm_count_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const int64_t *__restrict__ rowptrB, int64_t M, int64_t *__restrict__ prod) {
  const int lane = (int)(threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= M) return;
  const int64_t s = rowptrA[i], e = rowptrA[i + 1];
  int64_t p = 0;
  for (int64_t k = s + lane; k < e; k += 64) {
    const int64_t c = colA[k];
    p += rowptrB[c + 1] - rowptrB[c];
  }
  for (int off = 32; off > 0; off >>= 1) p += lane_xor(p, off);
  if (lane == 0) prod[i] = p;
}

// Bin rows by product count.  One thread per row; a wave reserves its slots in each bin with a
// single atomic (ballot + popcount) instead of one atomic per row on three hot counters.
__global__ __launch_bounds__(256) void spspmm_bin_kernel(const int64_t *__restrict__ prod, int64_t M,
                                                        int64_t *__restrict__ bins,
                                                        unsigned long long *stats) {
  const int lane = (int)(threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t p = i < M ? prod[i] : 0;
  const int b = p == 0 ? -1 : (p <= kSmallCap ? 0 : (p <= kMediumCap ? 1 : 2));
  for (int bin = 0; bin < 3; ++bin) {
    const unsigned long long m = __ballot(b == bin);
    if (m == 0) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&stats[ST_NSMALL + bin], (unsigned long long)__popcll(m));
    base = (unsigned long long)lane_read((int64_t)base, 0);
    if (b == bin) bins[(int64_t)bin * M + (int64_t)base + __popcll(m & ((1ull << lane) - 1ull))] = i;
  }
  if (b == 2) {
    int64_t pl = p;  // products in large rows (few rows: per-row atomics are fine)
    atomicAdd(&stats[ST_PLARGE], (unsigned long long)pl);
  }
}

template <int NW>
__device__ inline int block_exclusive_scan_small(int v, int *smem, int *total) {
  const int lane = (int)(threadIdx.x & 63);
  const int wid = (int)(threadIdx.x >> 6);
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = lane_read(inc, lane >= off ? lane - off : lane);
    if (lane >= off) inc += o;
  }
  if (NW == 1) {
    *total = lane_read(inc, 63);
    return inc - v;
  }
  if (lane == 63) smem[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int s = smem[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// Stable LSD radix sort of n <= 512 (key, value) pairs in LDS by ONE wavefront: 8-bit digits,
// ranks from 8 ballots per key (same scheme as radix_scatter_kernel in sort.hip), ping-pong
// between two LDS buffers.  ~10x fewer dependent LDS round trips than a bitonic network.
template <typename A>
__device__ inline void wave_radix_sort_lds(uint32_t *&ka, A *&va, uint32_t *&kb, A *&vb, int n,
                                           int passes, uint32_t *cnt) {
  const int lane = (int)(threadIdx.x & 63);
  constexpr int kItems = kSmallCap / 64;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = pass * 8;
    for (int c = lane; c < 256; c += 64) cnt[c] = 0;
    __syncthreads();
    uint32_t key[kItems], lr[kItems];
    A val[kItems];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      key[i] = 0;
      lr[i] = 0;
      val[i] = A(0);
      if (i * 64 < n) {  // wave-uniform
        const int idx = i * 64 + lane;
        const bool valid = idx < n;
        if (valid) {
          key[i] = ka[idx];
          val[i] = va[idx];
        }
        const uint32_t d = (key[i] >> shift) & 255u;
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
          pre = cnt[d];
          cnt[d] = pre + (uint32_t)__popcll(peers);
        }
        pre = lane_read(pre, leader);
        lr[i] = pre + rank;
      }
    }
    __syncthreads();
    {  // exclusive scan of the 256 digit counts: 4 digits per lane
      const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2],
                     c3 = cnt[4 * lane + 3];
      const uint32_t s = c0 + c1 + c2 + c3;
      uint32_t incl = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = lane_read(incl, lane >= off ? lane - off : lane);
        if (lane >= off) incl += o;
      }
      const uint32_t base = incl - s;
      cnt[4 * lane] = base;
      cnt[4 * lane + 1] = base + c0;
      cnt[4 * lane + 2] = base + c0 + c1;
      cnt[4 * lane + 3] = base + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      if (i * 64 + lane < n) {
        const uint32_t pos = cnt[(key[i] >> shift) & 255u] + lr[i];
        kb[pos] = key[i];
        vb[pos] = val[i];
      }
    }
    __syncthreads();
    uint32_t *tk = ka; ka = kb; kb = tk;
    A *tv = va; va = vb; vb = tv;
  }
}

// One workgroup per row: expand into LDS, sort by column, compress, write.
template <typename T, int BLOCK, int CAP>
__global__ __launch_bounds__(BLOCK) void spspmm_row_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const T *__restrict__ valA, const int64_t *__restrict__ rowptrB,
    const int64_t *__restrict__ colB, const T *__restrict__ valB,
    const int64_t *__restrict__ prodptr, const int64_t *__restrict__ rows,
    int64_t *__restrict__ colT, T *__restrict__ valT, int64_t *__restrict__ nnzC, int passes) {
  using A = typename Traits<T>::acc_t;
  __shared__ uint32_t scol_[CAP];
  __shared__ A sval_[CAP];
  __shared__ uint32_t scol2_[BLOCK == 64 ? CAP : 1];  // ping-pong buffers of the wave radix sort
  __shared__ A sval2_[BLOCK == 64 ? CAP : 1];
  __shared__ uint32_t scnt[BLOCK == 64 ? 256 : 1];
  __shared__ int sscan[8];
  uint32_t *scol = scol_, *scol2 = scol2_;
  A *sval = sval_, *sval2 = sval2_;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int64_t i = rows[blockIdx.x];
  const int64_t as = rowptrA[i], ae = rowptrA[i + 1];
  const int64_t slot = prodptr[i];
  const int p = (int)(prodptr[i + 1] - slot);
  int n2 = 2;
  while (n2 < p) n2 <<= 1;

  // ---- expand: the A row is read in 64-entry chunks (one entry per lane: column, start and
  //      length of the B row, value); the chunk's products are then a flat index space that the
  //      whole workgroup strides over, each thread locating its B row by a binary search over the
  //      chunk's prefix sums in LDS -- independent gathers, several in flight per thread (walking
  //      the B rows one after the other serialises a global-load latency per A entry) ----
  __shared__ int s_off[65];
  __shared__ int64_t s_bs[64];
  __shared__ A s_av[64];
  int filled = 0;
  for (int64_t e0 = as; e0 < ae; e0 += 64) {
    const int64_t e = e0 + lane;
    int64_t bs = 0;
    int d = 0;
    A av = A(1);
    if (e < ae) {
      const int64_t c = colA[e];
      bs = rowptrB[c];
      d = (int)(rowptrB[c + 1] - bs);
      if (valA != nullptr) av = Traits<T>::to_acc(valA[e]);
    }
    int incl = d;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = lane_read(incl, lane >= off ? lane - off : lane);
      if (lane >= off) incl += o;
    }
    if (tid < 64) {  // every wave holds the same chunk; the first one publishes it
      s_off[lane] = incl - d;
      s_bs[lane] = bs;
      s_av[lane] = av;
      if (lane == 63) s_off[64] = incl;
    }
    __syncthreads();
    const int total = s_off[64];
#pragma unroll 4
    for (int q = tid; q < total; q += BLOCK) {
      int lo = 0, hi = 64;  // last entry whose offset is <= q (zero-length entries are skipped)
#pragma unroll
      for (int step = 0; step < 6; ++step) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= q) lo = mid; else hi = mid;
      }
      const int64_t src = s_bs[lo] + (q - s_off[lo]);
      scol[filled + q] = (uint32_t)colB[src];
      sval[filled + q] = valB != nullptr ? s_av[lo] * Traits<T>::to_acc(valB[src]) : s_av[lo];
    }
    filled += total;
    __syncthreads();
  }
  if constexpr (BLOCK == 64) {
    wave_radix_sort_lds<A>(scol, sval, scol2, sval2, p, passes, scnt);
  } else {
  for (int j = p + tid; j < n2; j += BLOCK) {
    scol[j] = 0xFFFFFFFFu;
    sval[j] = A(0);
  }
  __syncthreads();

  // ---- bitonic sort by column (pairs) ----
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (n2 >> 1); t += BLOCK) {
        const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // j is a power of two
        const int b = a + j;
        const bool up = (a & k) == 0;
        const uint32_t ca = scol[a], cb = scol[b];
        if ((ca > cb) == up && ca != cb) {
          scol[a] = cb;
          scol[b] = ca;
          const A va = sval[a];
          sval[a] = sval[b];
          sval[b] = va;
        }
      }
      __syncthreads();
    }
  }
  }

  // ---- compress equal columns, write the row into its slot ----
  int base = 0;
  for (int c0 = 0; c0 < p; c0 += BLOCK) {
    const int idx = c0 + tid;
    const bool head = idx < p && (idx == 0 || scol[idx] != scol[idx - 1]);
    int tot;
    const int pos = base + block_exclusive_scan_small<BLOCK / 64>(head ? 1 : 0, sscan, &tot);
    if (head) {
      A acc = sval[idx];
      const uint32_t c = scol[idx];
      for (int q = idx + 1; q < p && scol[q] == c; ++q) acc += sval[q];
      colT[slot + pos] = (int64_t)c;
      if (valT != nullptr) valT[slot + pos] = Traits<T>::from_acc(acc);
    }
    base += tot;
  }
  if (tid == 0) nnzC[i] = base;
}

// Large rows: expand (row, col, val) triples to HBM at lp[r] (exclusive scan of their products).
template <typename T>
__global__ __launch_bounds__(256) void spspmm_expand_large_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const T *__restrict__ valA, const int64_t *__restrict__ rowptrB,
    const int64_t *__restrict__ colB, const T *__restrict__ valB,
    const int64_t *__restrict__ rows, const int64_t *__restrict__ lp, int64_t *__restrict__ erow,
    int64_t *__restrict__ ecol, T *__restrict__ eval) {
  using A = typename Traits<T>::acc_t;
  const int tid = (int)threadIdx.x;
  const int64_t i = rows[blockIdx.x];
  int64_t out = lp[blockIdx.x];
  for (int64_t e = rowptrA[i]; e < rowptrA[i + 1]; ++e) {
    const int64_t c = colA[e];
    const int64_t bs = rowptrB[c], d = rowptrB[c + 1] - bs;
    const A av = valA != nullptr ? Traits<T>::to_acc(valA[e]) : A(1);
    for (int64_t j = tid; j < d; j += 256) {
      erow[out + j] = i;
      ecol[out + j] = colB[bs + j];
      if (eval != nullptr)
        eval[out + j] = Traits<T>::from_acc(valB != nullptr ? av * Traits<T>::to_acc(valB[bs + j]) : av);
    }
    out += d;
  }
}

__global__ void gather_prod_kernel(const int64_t *__restrict__ rows, const int64_t *__restrict__ prodptr,
                                   int64_t n, int64_t *__restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = prodptr[rows[r] + 1] - prodptr[rows[r]];
}

// unique (row, col) of the large rows -> their slots; T values summed through perm/seg_ptr
template <typename T>
__global__ void spspmm_scatter_large_kernel(const int64_t *__restrict__ row_u,
                                            const int64_t *__restrict__ col_u,
                                            const int64_t *__restrict__ seg_ptr,
                                            const int64_t *__restrict__ perm,
                                            const T *__restrict__ eval,
                                            const int64_t *__restrict__ nuniq,
                                            const int64_t *__restrict__ prodptr,
                                            int64_t *__restrict__ colT, T *__restrict__ valT,
                                            int64_t *__restrict__ nnzC, int64_t cap) {
  using A = typename Traits<T>::acc_t;
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = *nuniq;
  if (q >= n || q >= cap) return;
  const int64_t r = row_u[q];
  int64_t lo = 0, hi = q;  // first q' with row_u[q'] == r
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (row_u[mid] < r) lo = mid + 1; else hi = mid;
  }
  const int64_t j = q - lo;
  const int64_t dst = prodptr[r] + j;
  colT[dst] = col_u[q];
  if (valT != nullptr) {
    A acc = A(0);
    for (int64_t t = seg_ptr[q]; t < seg_ptr[q + 1]; ++t) acc += Traits<T>::to_acc(eval[perm[t]]);
    valT[dst] = Traits<T>::from_acc(acc);
  }
  if (q == n - 1 || row_u[q + 1] != r) nnzC[r] = j + 1;
}

template <typename T>
__global__ void spspmm_compact_kernel(const int64_t *__restrict__ rowC, const int64_t *__restrict__ rowptrC,
                                      const int64_t *__restrict__ prodptr, const int64_t *__restrict__ colT,
                                      const T *__restrict__ valT, int64_t nnz, int64_t *__restrict__ colC,
                                      T *__restrict__ valC) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nnz) return;
  const int64_t r = rowC[q];
  const int64_t src = prodptr[r] + (q - rowptrC[r]);
  colC[q] = colT[src];
  if (valC != nullptr) valC[q] = valT[src];
}

template <typename T>
int run_rows(const int64_t *rowptrA, const int64_t *colA, const void *valA, const int64_t *rowptrB,
             const int64_t *colB, const void *valB, const int64_t *prodptr, const int64_t *bins,
             int64_t M, int64_t N, int64_t n_small, int64_t n_medium, int64_t *colT, void *valT,
             int64_t *nnzC, hipStream_t stream) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < N) ++bits;
  const int passes = (bits + 7) / 8;  // 8-bit radix passes over the column ids
  const T *va = reinterpret_cast<const T *>(valA);
  const T *vb = reinterpret_cast<const T *>(valB);
  T *vt = reinterpret_cast<T *>(valT);
  if (n_small > 0) {
    hipLaunchKernelGGL((spspmm_row_kernel<T, 64, kSmallCap>), dim3((unsigned int)n_small), dim3(64), 0,
                       stream, rowptrA, colA, va, rowptrB, colB, vb, prodptr, bins, colT, vt, nnzC, passes);
    TSAMD_LAUNCH_CHECK();
  }
  if (n_medium > 0) {
    hipLaunchKernelGGL((spspmm_row_kernel<T, 256, kMediumCap>), dim3((unsigned int)n_medium), dim3(256),
                       0, stream, rowptrA, colA, va, rowptrB, colB, vb, prodptr, bins + M, colT, vt,
                       nnzC, passes);
    TSAMD_LAUNCH_CHECK();
  }
  return TSAMD_OK;
}

struct LargeWs {
  int64_t *lp, *erow, *ecol, *row_s, *col_s, *perm, *row_u, *col_u, *seg, *nuniq;
  void *eval, *sort_ws, *coal_ws, *scan_ws;
  size_t sort_bytes, coal_bytes;
};

size_t carve_large(void *base, int64_t n_large, int64_t P_large, size_t esize, LargeWs *w) {
  char *p = reinterpret_cast<char *>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) -> void * {
    void *r = p ? p + off : nullptr;
    off += align_up(bytes > 0 ? bytes : 1, 256);
    return r;
  };
  LargeWs l;
  const size_t P = (size_t)P_large;
  l.lp = (int64_t *)take(8 * (size_t)(n_large + 1));
  l.erow = (int64_t *)take(8 * P);
  l.ecol = (int64_t *)take(8 * P);
  l.eval = take(esize * P);
  l.row_s = (int64_t *)take(8 * P);
  l.col_s = (int64_t *)take(8 * P);
  l.perm = (int64_t *)take(8 * P);
  l.row_u = (int64_t *)take(8 * P);
  l.col_u = (int64_t *)take(8 * P);
  l.seg = (int64_t *)take(8 * (P + 1));
  l.nuniq = (int64_t *)take(8);
  l.sort_bytes = tsamd_sort_coo_workspace_bytes(P_large);
  l.sort_ws = take(l.sort_bytes);
  l.coal_bytes = tsamd_coalesce_workspace_bytes(P_large);
  l.coal_ws = take(l.coal_bytes);
  l.scan_ws = take(scan_workspace_bytes(n_large));
  if (w) *w = l;
  return off;
}

template <typename T>
int run_large(const int64_t *rowptrA, const int64_t *colA, const void *valA, const int64_t *rowptrB,
              const int64_t *colB, const void *valB, const int64_t *prodptr, const int64_t *rows,
              int64_t n_large, int64_t P_large, int64_t M, int64_t N, int64_t *colT, void *valT,
              int64_t *nnzC, void *workspace, hipStream_t stream) {
  LargeWs w;
  carve_large(workspace, n_large, P_large, sizeof(T), &w);
  hipLaunchKernelGGL(gather_prod_kernel, dim3((unsigned int)ceil_div(n_large, 256)), dim3(256), 0,
                     stream, rows, prodptr, n_large, w.lp);
  TSAMD_LAUNCH_CHECK();
  int st = exclusive_scan_i64(w.lp, w.lp, n_large, nullptr, w.scan_ws, stream);
  if (st != TSAMD_OK) return st;
  T *ev = valT ? reinterpret_cast<T *>(w.eval) : nullptr;
  hipLaunchKernelGGL((spspmm_expand_large_kernel<T>), dim3((unsigned int)n_large), dim3(256), 0,
                     stream, rowptrA, colA, reinterpret_cast<const T *>(valA), rowptrB, colB,
                     reinterpret_cast<const T *>(valB), rows, (const int64_t *)w.lp, w.erow, w.ecol, ev);
  TSAMD_LAUNCH_CHECK();
  st = tsamd_sort_coo(w.erow, w.ecol, P_large, M, N, w.row_s, w.col_s, w.perm, w.sort_ws,
                      w.sort_bytes, stream);
  if (st != TSAMD_OK) return st;
  st = tsamd_coalesce_index(w.row_s, w.col_s, P_large, w.row_u, w.col_u, w.seg, w.nuniq, w.coal_ws,
                            w.coal_bytes, stream);
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL((spspmm_scatter_large_kernel<T>), dim3((unsigned int)ceil_div(P_large, 256)),
                     dim3(256), 0, stream, (const int64_t *)w.row_u, (const int64_t *)w.col_u,
                     (const int64_t *)w.seg, (const int64_t *)w.perm, (const T *)ev,
                     (const int64_t *)w.nuniq, prodptr, colT, reinterpret_cast<T *>(valT), nnzC,
                     P_large);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" size_t tsamd_exclusive_scan_workspace_bytes(int64_t n) { return scan_workspace_bytes(n); }

extern "C" int tsamd_exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, int64_t *total,
                                        void *workspace, size_t workspace_bytes, void *stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return TSAMD_ERR_INVALID;
  if (n > kScanTile && (!workspace || workspace_bytes < scan_workspace_bytes(n)))
    return TSAMD_ERR_WORKSPACE;
  return exclusive_scan_i64(in, out, n, total, workspace, reinterpret_cast<hipStream_t>(stream));
}

extern "C" size_t tsamd_spspmm_plan_workspace_bytes(int64_t M) { return scan_workspace_bytes(M + 1); }

extern "C" int tsamd_spspmm_plan(const int64_t *rowptrA, const int64_t *colA,
                                 const int64_t *rowptrB, int64_t M, int64_t *prodptr,
                                 int64_t *bins, int64_t *stats, void *workspace,
                                 size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || !prodptr || !stats) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(stats, 0, 8 * sizeof(int64_t), stream));
  TSAMD_HIP_TRY(hipMemsetAsync(prodptr, 0, sizeof(int64_t) * (size_t)(M + 1), stream));
  if (M == 0) return TSAMD_OK;
  if (!rowptrA || !rowptrB || !bins) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < scan_workspace_bytes(M + 1)) return TSAMD_ERR_WORKSPACE;
  hipLaunchKernelGGL(spspmm_count_kernel, dim3((unsigned int)ceil_div(M, 4)), dim3(256), 0, stream,
                     rowptrA, colA, rowptrB, M, prodptr);
  TSAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(spspmm_bin_kernel, dim3((unsigned int)ceil_div(M, 256)), dim3(256), 0, stream,
                     (const int64_t *)prodptr, M, bins, reinterpret_cast<unsigned long long *>(stats));
  TSAMD_LAUNCH_CHECK();
  // prodptr[0..M) holds counts, prodptr[M] = 0: the scan over M + 1 entries leaves the total there
  return exclusive_scan_i64(prodptr, prodptr, M + 1, stats + ST_P, workspace, stream);
}

extern "C" size_t tsamd_spspmm_rows_workspace_bytes(int dtype, int64_t n_large, int64_t P_large) {
  if (n_large <= 0) return 0;
  return carve_large(nullptr, n_large, P_large, dtype_size(dtype), nullptr);
}

extern "C" int tsamd_spspmm_rows(int dtype, const int64_t *rowptrA, const int64_t *colA,
                                 const void *valA, const int64_t *rowptrB, const int64_t *colB,
                                 const void *valB, int64_t M, int64_t N, const int64_t *prodptr,
                                 const int64_t *bins, int64_t n_small, int64_t n_medium,
                                 int64_t n_large, int64_t P_large, int64_t *colT, void *valT,
                                 int64_t *nnzC, void *workspace, size_t workspace_bytes,
                                 void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64) return TSAMD_ERR_UNSUPPORTED;
  if (M < 0 || N < 0 || N >= (int64_t)1 << 32) return TSAMD_ERR_UNSUPPORTED;
  if (M == 0) return TSAMD_OK;
  if (!nnzC) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(nnzC, 0, sizeof(int64_t) * (size_t)M, stream));
  if (n_small + n_medium + n_large == 0) return TSAMD_OK;
  if (!rowptrA || !colA || !rowptrB || !colB || !prodptr || !bins || !colT) return TSAMD_ERR_INVALID;
  if (n_large > 0 &&
      (!workspace || workspace_bytes < tsamd_spspmm_rows_workspace_bytes(dtype, n_large, P_large)))
    return TSAMD_ERR_WORKSPACE;
  int st;
  if (dtype == TSAMD_F32)
    st = run_rows<float>(rowptrA, colA, valA, rowptrB, colB, valB, prodptr, bins, M, N, n_small, n_medium,
                         colT, valT, nnzC, stream);
  else
    st = run_rows<double>(rowptrA, colA, valA, rowptrB, colB, valB, prodptr, bins, M, N, n_small,
                          n_medium, colT, valT, nnzC, stream);
  if (st != TSAMD_OK || n_large == 0) return st;
  if (dtype == TSAMD_F32)
    return run_large<float>(rowptrA, colA, valA, rowptrB, colB, valB, prodptr, bins + 2 * M, n_large,
                            P_large, M, N, colT, valT, nnzC, workspace, stream);
  return run_large<double>(rowptrA, colA, valA, rowptrB, colB, valB, prodptr, bins + 2 * M, n_large,
                           P_large, M, N, colT, valT, nnzC, workspace, stream);
}

extern "C" int tsamd_spspmm_compact(int dtype, const int64_t *rowC, const int64_t *rowptrC,
                                    const int64_t *prodptr, const int64_t *colT, const void *valT,
                                    int64_t nnz, int64_t *colC, void *valC, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64) return TSAMD_ERR_UNSUPPORTED;
  if (nnz < 0) return TSAMD_ERR_INVALID;
  if (nnz == 0) return TSAMD_OK;
  if (!rowC || !rowptrC || !prodptr || !colT || !colC) return TSAMD_ERR_INVALID;
  const unsigned int blocks = (unsigned int)ceil_div(nnz, 256);
  if (dtype == TSAMD_F32)
    hipLaunchKernelGGL((spspmm_compact_kernel<float>), dim3(blocks), dim3(256), 0, stream, rowC, rowptrC,
                       prodptr, colT, reinterpret_cast<const float *>(valT), nnz, colC,
                       reinterpret_cast<float *>(valC));
  else
    hipLaunchKernelGGL((spspmm_compact_kernel<double>), dim3(blocks), dim3(256), 0, stream, rowC,
                       rowptrC, prodptr, colT, reinterpret_cast<const double *>(valT), nnz, colC,
                       reinterpret_cast<double *>(valC));
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}
