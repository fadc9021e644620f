// SpSpMM  C = A * B  (CSR x CSR -> CSR, sum) on gfx950: count first, write once.
//
// The reference has no native code for this: torch_sparse/matmul.py:94-111 converts both
// operands to torch sparse COO and calls torch.sparse.mm (PyTorch's CPU SpGEMM / hipSPARSE),
// then trusts the result to be row-major sorted and coalesced.  This file produces exactly
// that result shape: every row of C sorted by column, duplicates summed, explicit zeros kept.
//
// Pipeline (stages are separate C-ABI calls because the host allocates between them):
//   plan      wave per row of A: products(i) = sum_{k in A_i} |B_k|; rows are binned (small <= 512
//             products, medium <= 4096, large) with one atomic per wave and bin.
//   symbolic  exact nnz of every row of C.  small / medium rows: one wave / one 256-thread
//             workgroup expands the row's product COLUMNS straight into an LDS hash set (1 Ki / 8 Ki
//             slots) and counts the successful inserts.  Large rows: their products are binned by
//             column range into HBM scratch (4 + sizeof(T) bytes per product) and counted bin by bin
//             with an LDS occupancy bitmap; the bins stay in the workspace for the numeric stage.
//   (host)    exclusive scan of the counts = the FINAL rowptr of C; one sync for nnz(C); colC / valC
//             are allocated at their final size -- no per-row slot buffer of `products` entries,
//             no compaction copy.
//   numeric   small rows: the row's products are expanded into LDS as 32-bit keys
//             (column << 9 | product index) next to their values, pulled into registers
//             (1 / 2 / 4 / 8 keys per lane) and sorted by a wave-level bitonic network -- min/max
//             on unique keys, DPP / swizzle / bpermute exchanges, no LDS allocation, no barriers,
//             no data-dependent step -- written back, equal columns summed in product order
//             (deterministic) and the compressed row stored at its final position.  When
//             ceil(log2 N) > 23 the keys do not fit 32 bits: LDS radix sort of (column, value)
//             pairs instead.  Medium rows: 256-thread bitonic sort of pairs in LDS.  Large rows:
//             one column range at a time in dense LDS accumulators (see "large rows" below).
#include "common.h"
#include "scan.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace tsamd {
namespace {

constexpr int kSmallCap = 512;    // products handled by one wave (LDS radix sort, no workgroup barriers)
// Bins of the large-row path that one wave takes through the register sort (kSmallCap for rows stays 512: the
// config-4 rows of ~240 products want the small LDS footprint).  A persistent 256-thread workgroup spends ~8-10 us
// of barriers and dependent loads on a bin whatever its size, and 720 k of the 790 k "big" bins of the stress
// product hold 513..4096 products.  Same-box stress run, cap 512 / 1024 / 2048: the four count + accum kernels
// take 29.8 / 25.7 / 30.8 ms together (at 2048 the 32-keys-per-lane register sort costs more than the barriers saved).
#ifndef TSAMD_SPSPMM_SMALLBIN_CAP
#define TSAMD_SPSPMM_SMALLBIN_CAP 1024
#endif
constexpr int kSmallBinCap = TSAMD_SPSPMM_SMALLBIN_CAP;
constexpr int kBinIdxBits = kSmallBinCap <= 512 ? 9 : (kSmallBinCap <= 1024 ? 10 : 11);
// Rows of 513..kMediumCap products are sorted by one 256-thread workgroup in LDS; longer rows take the binned
// path.  Same-box A/B (scripts/ab_spspmm_medium.py; R-MAT stress product / uniform product with ~1600 products per
// row): cap 4096: 50.5 / 14.7 ms, 2048: 46.6-47.0 / 11.0-11.1 ms, 1024: 47.0 / 10.1 ms, 512 (no medium class):
// 49.8 / 10.1-10.2 ms -- the LDS bitonic sort pays a workgroup barrier per stage (78 stages at 4096 keys) and its
// LDS footprint costs occupancy; the binned path is faster even for uniform rows of a few thousand products.
#ifndef TSAMD_SPSPMM_MEDIUM_CAP
#define TSAMD_SPSPMM_MEDIUM_CAP 1024
#endif
constexpr int kMediumCap = TSAMD_SPSPMM_MEDIUM_CAP;  // products handled by one 256-thread workgroup
constexpr int kMediumLogT = kMediumCap <= 1024 ? 11 : (kMediumCap <= 2048 ? 12 : 13);  // hash set of the symbolic stage

// stats layout (device int64[8])
enum { ST_NMEDIUM = 2, ST_NLARGE = 3, ST_PLARGE = 4, ST_PMAX = 5 };

// products(i) = sum over the entries k of row i of A of |B_k|.  8 lanes per row (32 rows per
// 256-thread workgroup): rows of a few dozen entries keep most lanes busy, hub rows just loop.
constexpr int kCountLanes = 8;
constexpr int kCountLong = 512;  // entries of a row of A beyond which the whole wave counts it

__global__ __launch_bounds__(256) void spspmm_count_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const int64_t *__restrict__ rowptrB, int64_t M, int64_t *__restrict__ prod) {
  const int sub = (int)(threadIdx.x & (kCountLanes - 1));
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kCountLanes;
  int64_t p = 0;
  int64_t s = 0, e = 0;
  if (i < M) {
    s = rowptrA[i];
    e = rowptrA[i + 1];
  }
  // Rows beyond kCountLong entries are left to the whole wave below: with 8 lanes a hub row is a chain of
  // (entries / 8) x 2 dependent round trips -- the 14 911-entry row of the R-MAT stress operand kept ONE lane group busy
  // for 0.9 ms, which was the kernel's whole time (0.97 ms for 4 M entries).
  const bool is_long = e - s > kCountLong;
  if (!is_long) {
    for (int64_t k = s + sub; k < e; k += kCountLanes) {
      const int64_t c = colA[k];
      p += rowptrB[c + 1] - rowptrB[c];
    }
  }
#pragma unroll
  for (int off = kCountLanes / 2; off > 0; off >>= 1) p += lane_xor(p, off);
  if (i < M && sub == 0 && !is_long) prod[i] = p;
  unsigned long long todo = __ballot(is_long && sub == 0);
  const int lane = (int)(threadIdx.x & 63);
  while (todo != 0ull) {  // (wave-uniform) one long row at a time, 64 lanes x 4 entries in flight
    const int lead = (int)__builtin_ctzll(todo);
    todo &= todo - 1ull;
    const int64_t rs = lane_read(s, lead), re = lane_read(e, lead), ri = lane_read(i, lead);
    int64_t acc = 0;
    for (int64_t k0 = rs + lane; k0 < re; k0 += 64 * 4) {
      int64_t c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + 64 * u;
        c[u] = colA[k < re ? k : re - 1];
      }
      int64_t lo[4], hi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        lo[u] = rowptrB[c[u]];
        hi[u] = rowptrB[c[u] + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += (k0 + 64 * u < re) ? hi[u] - lo[u] : 0;
    }
    for (int off = 32; off > 0; off >>= 1) acc += lane_xor(acc, off);
    if (lane == 0) prod[ri] = acc;
  }
}

// Column ids of B as 32-bit words (N < 2^32 - 1): the expansion gathers short B rows from all over the
// array, and a row of ~15 ids then straddles one or two 128-byte lines instead of two or three.
__global__ __launch_bounds__(256) void spspmm_narrow_cols_kernel(const int64_t *__restrict__ col, int64_t n,
                                                                uint32_t *__restrict__ col32) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    col32[i] = (uint32_t)col[i];
}

// Rows of more than kSmallCap products are listed by size class (medium | large); small rows are
// not listed: their kernels run over all rows in natural order and skip the others (better locality
// of the A rows and of the output, and no atomics at all when every row is small).  One thread per
// row; a wave reserves its slots in a list with a single atomic (ballot + popcount).
__global__ __launch_bounds__(256) void spspmm_bin_kernel(const int64_t *__restrict__ prod, int64_t M,
                                                        int64_t *__restrict__ bins,
                                                        unsigned long long *stats) {
  const int lane = (int)(threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t p = i < M ? prod[i] : 0;
  const int b = p <= kSmallCap ? -1 : (p <= kMediumCap ? 0 : 1);
  // one returning atomic per WORKGROUP and list (the two list counters are hot words, ~12 ns per atomic: a wave-level
  // reservation was 16 k serialised atomics on the stress product)
  __shared__ int s_cnt[2][4];
  __shared__ unsigned long long s_base[2];
  const int wid = (int)(threadIdx.x >> 6);
  unsigned long long m2[2];
  for (int bin = 0; bin < 2; ++bin) {
    m2[bin] = __ballot(b == bin);
    if (lane == 0) s_cnt[bin][wid] = __popcll(m2[bin]);
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int tot = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
    s_base[threadIdx.x] = tot ? atomicAdd(&stats[ST_NMEDIUM + threadIdx.x], (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  for (int bin = 0; bin < 2; ++bin) {
    if (b != bin) continue;
    unsigned long long base = s_base[bin];
    for (int w = 0; w < wid; ++w) base += (unsigned long long)s_cnt[bin][w];
    bins[(int64_t)bin * M + (int64_t)base + __popcll(m2[bin] & ((1ull << lane) - 1ull))] = i;
  }
  // products of the large rows and the largest one: reduced over the wave first -- the two words are hot addresses (77 k
  // large rows of the stress product were 154 k atomics on them: 0.35 ms for a kernel that moves 4 MB)
  unsigned long long ps = b == 1 ? (unsigned long long)p : 0ull, pm = ps;
  if (__ballot(b == 1) != 0ull) {  // wave-uniform
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long os = (unsigned long long)lane_xor((int64_t)ps, off), om = (unsigned long long)lane_xor((int64_t)pm, off);
      ps += os;
      pm = om > pm ? om : pm;
    }
    if (lane == 0) {
      atomicAdd(&stats[ST_PLARGE], ps);
      atomicMax(&stats[ST_PMAX], pm);  // largest row: the (row, range) counters are 32-bit
    }
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

// ---------------------------------------------------------------------------
// expand: the A row is read in 64-entry chunks (one entry per lane: column, start and length
// of the B row, value); the chunk's products are then a flat index space that the whole
// workgroup strides over, each thread locating its B row by a binary search over the chunk's
// prefix sums in LDS -- independent gathers, several in flight per thread (walking the B rows
// one after the other serialises a global-load latency per A entry).
// emit(q, col, value) is called once per product: q = its index in expansion order (A entry
// order, then B entry order), col = its column (32 bits), value = a * b (1 without values).
// The B entries are fetched kExpandBatch at a time per thread: all their addresses are resolved
// first, then the loads are issued back to back and only then consumed (a load / use / LDS-store
// chain per product costs one global-memory latency per product).
// ---------------------------------------------------------------------------
constexpr int kExpandBatch = 4;

#ifndef TSAMD_SPSPMM_OWNER_SCAN
#define TSAMD_SPSPMM_OWNER_SCAN 1  // 0: the one-wave kernels locate a product's A entry by the 6-step search (round 1-3), for A/B builds
#endif

template <typename A>
struct ExpandScratch {
  int off[65];
  int64_t bs[64];
  A av[64];
  alignas(4) uint8_t own[256];  // expand_row_wave: owner (lane + 1) of each of the 256 products of a batch
};

// wave64 inclusive scans on the DPP network (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast 15 / 31
// across them): six VALU instructions, no LDS-pipe round trips (the ds_bpermute form costs six dependent ones).
// `old` = 0 is the identity of both operations on unsigned values; lanes without a source keep it.
#define TSAMD_DPP_STEP(OP, CTRL, ROWMASK) \
  v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false))
__device__ __forceinline__ uint32_t dpp_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_scan_add_dpp(uint32_t v) {
  TSAMD_DPP_STEP(dpp_add, 0x111, 0xF);
  TSAMD_DPP_STEP(dpp_add, 0x112, 0xF);
  TSAMD_DPP_STEP(dpp_add, 0x114, 0xF);
  TSAMD_DPP_STEP(dpp_add, 0x118, 0xF);
  TSAMD_DPP_STEP(dpp_add, 0x142, 0xA);
  TSAMD_DPP_STEP(dpp_add, 0x143, 0xC);
  return v;
}
__device__ __forceinline__ uint32_t wave_scan_max_dpp(uint32_t v) {
  TSAMD_DPP_STEP(dpp_max, 0x111, 0xF);
  TSAMD_DPP_STEP(dpp_max, 0x112, 0xF);
  TSAMD_DPP_STEP(dpp_max, 0x114, 0xF);
  TSAMD_DPP_STEP(dpp_max, 0x118, 0xF);
  TSAMD_DPP_STEP(dpp_max, 0x142, 0xA);
  TSAMD_DPP_STEP(dpp_max, 0x143, 0xC);
  return v;
}
#undef TSAMD_DPP_STEP

// One-wave form of expand_row (rows of at most kSmallCap products: the symbolic and numeric kernels of the small
// rows).  Same products, same indices q in expansion order, but handed over four at a time -- emit(q0, n, col[4],
// value[4]): the lane's products q0 .. q0 + n - 1, n in 0..4 (each_product() adapts a per-product functor) -- because a
// lane takes FOUR CONSECUTIVE products of a 256-product batch and finds their A entries without searching: every entry that reaches
// into the batch leaves (its lane + 1) in a byte at the position of its first product there, and a max-scan over the
// 256 bytes (in-lane over the dword a lane reads back, then the DPP scan across lanes) carries each owner forward to
// the products behind it.  The 6-step LDS binary search per product that this replaces was 184 of the 448 VALU
// instructions a row-wave of the symbolic kernel issued at configs[3] (and 24 dependent LDS reads); the kernels
// are bound by instruction issue (SQ counters, profiles/r03_sq_counters.md).
template <typename Emit>
struct EachProduct {
  Emit emit;
  template <typename A>
  __device__ __forceinline__ void operator()(int q0, int n, const uint32_t (&c)[4], const A (&v)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < n) emit(q0 + u, c[u], v[u]);
  }
};
template <typename Emit>
__device__ __forceinline__ EachProduct<Emit> each_product(Emit emit) {
  return EachProduct<Emit>{emit};
}

// PRE: the caller already holds the per-lane (start of the B row, its length, value of the A entry) of a row of at
// most 64 entries -- the persistent kernels below fetch them a row ahead (RowPipe).
template <typename T, bool WITH_VAL, bool PRE = false, typename Emit>
__device__ __forceinline__ int expand_row_wave(const int64_t *__restrict__ colA, const T *__restrict__ valA,
                                               const int64_t *__restrict__ rowptrB,
                                               const uint32_t *__restrict__ colB, const T *__restrict__ valB,
                                               int64_t as, int64_t ae,
                                               ExpandScratch<typename Traits<T>::acc_t> &sc, Emit emit,
                                               int64_t pre_bs = 0, int pre_d = 0,
                                               typename Traits<T>::acc_t pre_av = typename Traits<T>::acc_t(1)) {
  using A = typename Traits<T>::acc_t;
  const int lane = (int)threadIdx.x;
  uint32_t *own_w = reinterpret_cast<uint32_t *>(sc.own);
  int filled = 0;
  // PRE: exactly one chunk (the caller sends rows of more than 64 entries through the loading form): no load of
  // this function then sits in a loop around the gathers, where the compiler's wait-count bookkeeping would make
  // the gathers wait for the caller's prefetches first
  for (int64_t e0 = as; e0 < (PRE ? (as < ae ? as + 1 : as) : ae); e0 += 64) {
    const int64_t e = e0 + lane;
    int64_t bs = 0;
    int d = 0;
    A av = A(1);
    if constexpr (PRE) {
      bs = pre_bs;
      d = pre_d;
      av = pre_av;
    } else if (e < ae) {
      const int64_t c = colA[e];
      bs = rowptrB[c];
      d = (int)(rowptrB[c + 1] - bs);
      if (WITH_VAL && valA != nullptr) av = Traits<T>::to_acc(valA[e]);
    }
    const int incl = (int)wave_scan_add_dpp((uint32_t)d);
    const int off = incl - d;
    sc.bs[lane] = bs - (int64_t)off;  // position in colB of the chunk's product 0 IF it belonged to this entry: one read per product
    if (WITH_VAL) sc.av[lane] = av;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    for (int b0 = 0; b0 < total; b0 += 256) {
      own_w[lane] = 0u;
      __syncthreads();  // (one wave: orders the LDS traffic)
      if (d > 0 && off < b0 + 256 && off + d > b0) sc.own[(off > b0 ? off : b0) - b0] = (uint8_t)(lane + 1);
      __syncthreads();
      const uint32_t w = own_w[lane];
      uint32_t o[4];
      o[0] = w & 0xFFu;
      o[1] = (w >> 8) & 0xFFu;
      o[2] = (w >> 16) & 0xFFu;
      o[3] = w >> 24;
      o[1] = o[1] > o[0] ? o[1] : o[0];
      o[2] = o[2] > o[1] ? o[2] : o[1];
      o[3] = o[3] > o[2] ? o[3] : o[2];
      const uint32_t inc = wave_scan_max_dpp(o[3]);
      // the lanes below me: the inclusive result of lane - 1 (wave_shr:1; lane 0 keeps 0)
      const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xF, 0xF, false);
      int64_t src[4];
      A a[4];
      const int qbase = b0 + 4 * lane;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int lo = (int)(o[u] > below ? o[u] : below) - 1;  // >= 0: the entry that covers product b0 marked slot 0
        const int qq = qbase + u;
        const int q = qq < total ? qq : total - 1;  // (the owner carried into a slot past the end owns total - 1)
        src[u] = sc.bs[lo] + (int64_t)q;
        a[u] = WITH_VAL ? sc.av[lo] : A(1);
      }
      uint32_t c[4];
      A b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c[u] = colB[src[u]];
        b[u] = (WITH_VAL && valB != nullptr) ? Traits<T>::to_acc(valB[src[u]]) : A(1);
      }
      {
        A pr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = a[u] * b[u];
        const int left = total - qbase;
        emit(filled + qbase, left < 0 ? 0 : (left > 4 ? 4 : left), c, pr);  // the lane's (up to) four consecutive products
      }
      // every gather of the batch has landed before the next batch (or the caller) goes on: without this a lane
      // whose product lies past the end never reads its register, the load stays "pending" on the loop's back edge,
      // and the compiler parks a full wait at the TOP of the batch loop -- in front of the gathers, where it also
      // waits for the caller's prefetches of the next rows
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("" ::"v"(c[u]));
        if constexpr (WITH_VAL) asm volatile("" ::"v"(b[u]));
      }
    }
    filled += total;
    __syncthreads();
  }
  return filled;
}

// LONG_B: the B rows are expected to be long (the large rows of a power-law product: 733 entries on average in
// the stress case), so a thread's successive products (BLOCK apart) mostly fall into the same A entry as its
// previous one: that entry is tried first and the 6-step search only runs on a miss.
template <typename T, int BLOCK, bool WITH_VAL, bool LONG_B = false, typename Emit>
__device__ __forceinline__ int expand_row(const int64_t *__restrict__ colA, const T *__restrict__ valA,
                                          const int64_t *__restrict__ rowptrB,
                                          const uint32_t *__restrict__ colB, const T *__restrict__ valB,
                                          int64_t as, int64_t ae,
                                          ExpandScratch<typename Traits<T>::acc_t> &sc, Emit emit) {
  using A = typename Traits<T>::acc_t;
#if TSAMD_SPSPMM_OWNER_SCAN
  if constexpr (BLOCK == 64 && !LONG_B)
    return expand_row_wave<T, WITH_VAL>(colA, valA, rowptrB, colB, valB, as, ae, sc, each_product(emit));
#endif
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
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
      if (WITH_VAL && valA != nullptr) av = Traits<T>::to_acc(valA[e]);
    }
    int incl = d;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = lane_read(incl, lane >= off ? lane - off : lane);
      if (lane >= off) incl += o;
    }
    if (tid < 64) {  // every wave holds the same chunk; the first one publishes it
      sc.off[lane] = incl - d;
      sc.bs[lane] = bs;
      if (WITH_VAL) sc.av[lane] = av;
      if (lane == 63) sc.off[64] = incl;
    }
    __syncthreads();
    const int total = sc.off[64];
    // LONG_B: the A entry of the thread's previous product stays in REGISTERS (its product range, the start of its B
    // row, its value): a hit costs no LDS access at all -- the hist / bin kernels of the large rows are bound by the
    // CU's LDS pipe (occupancy 4 -> 8 moved them by 5-18 %, more loads in flight by nothing), and the round-4 form
    // read five LDS words per product even on a hit.
#ifndef TSAMD_SPSPMM_REG_ENTRY
#define TSAMD_SPSPMM_REG_ENTRY 1
#endif
    int c_off = 0, c_end = 0;  // [c_off, c_end): products of the cached entry (empty: nothing cached for this chunk)
    int64_t c_bs = 0;
    A c_av = A(1);
    for (int q0 = tid; q0 < total; q0 += BLOCK * kExpandBatch) {
      int64_t src[kExpandBatch];
      A a[kExpandBatch];
#pragma unroll
      for (int u = 0; u < kExpandBatch; ++u) {
        const int qq = q0 + u * BLOCK;
        const int q = qq < total ? qq : total - 1;
        if constexpr (LONG_B) {
#if TSAMD_SPSPMM_REG_ENTRY
          if (!(c_off <= q && q < c_end)) {
#else
          {  // (A/B builds: search and read LDS for every product)
#endif
            int lo = 0, hi = 64;  // last entry whose offset is <= q (zero-length entries are skipped)
#pragma unroll
            for (int step = 0; step < 6; ++step) {
              const int mid = (lo + hi) >> 1;
              if (sc.off[mid] <= q) lo = mid; else hi = mid;
            }
            c_off = sc.off[lo];
            c_end = sc.off[lo + 1];
            c_bs = sc.bs[lo];
            if (WITH_VAL) c_av = sc.av[lo];
          }
          src[u] = c_bs + (q - c_off);
          a[u] = WITH_VAL ? c_av : A(1);
        } else {
          int lo = 0, hi = 64;  // last entry whose offset is <= q (zero-length entries are skipped)
#pragma unroll
          for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi) >> 1;
            if (sc.off[mid] <= q) lo = mid; else hi = mid;
          }
          src[u] = sc.bs[lo] + (q - sc.off[lo]);
          a[u] = WITH_VAL ? sc.av[lo] : A(1);
        }
      }
      uint32_t c[kExpandBatch];
      A b[kExpandBatch];
#pragma unroll
      for (int u = 0; u < kExpandBatch; ++u) {
        c[u] = colB[src[u]];
        b[u] = (WITH_VAL && valB != nullptr) ? Traits<T>::to_acc(valB[src[u]]) : A(1);
      }
#pragma unroll
      for (int u = 0; u < kExpandBatch; ++u) {
        const int qq = q0 + u * BLOCK;
        if (qq < total) emit(filled + qq, c[u], a[u] * b[u]);
      }
    }
    filled += total;
    __syncthreads();
  }
  return filled;
}

// ---------------------------------------------------------------------------
// symbolic: number of distinct columns among the row's products (LDS hash set)
// ---------------------------------------------------------------------------
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;  // column ids are < 2^32 - 1

// Slot of a column id in a table of 2^LOG_T entries (multiplicative hashing, linear probing behind it -- any function
// is CORRECT, a poor one only probes longer).  v_mul_lo_u32 runs at a quarter of the VALU rate; matrices of at most
// 2^24 columns (template flag of the kernel) take v_mul_u32_u24 instead: the top bits of the low 32 product bits mix every bit
// of a 24-bit id.
template <int LOG_T>
__device__ __forceinline__ uint32_t hash_slot(uint32_t c, bool narrow) {
  const uint32_t m = narrow ? __umul24(c, 0x9E3779u) : c * 0x9E3779B1u;
  return m >> (32 - LOG_T);
}

template <int BLOCK, int LOG_T, bool NARROW>
__global__ __launch_bounds__(BLOCK) void spspmm_symbolic_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const int64_t *__restrict__ rowptrB, const uint32_t *__restrict__ colB,
    const int64_t *__restrict__ prod, const int64_t *__restrict__ rows, int64_t *__restrict__ nnzC) {
  constexpr int kT = 1 << LOG_T;
  __shared__ uint32_t tab[kT];
  __shared__ ExpandScratch<float> sc;
  __shared__ int s_cnt[BLOCK / 64];
  const int tid = (int)threadIdx.x;
  int64_t i, as_i, ae_i;
  if constexpr (BLOCK == 64) {  // small rows: every row in natural order, the others are skipped
    i = blockIdx.x;
    const int64_t p = prod[i];
    as_i = rowptrA[i];  // (requested together with the product count: one round trip, not two)
    ae_i = rowptrA[i + 1];
    if (p == 0 || p > kSmallCap) return;
  } else {
    i = rows[blockIdx.x];
    as_i = rowptrA[i];
    ae_i = rowptrA[i + 1];
  }
  for (int t = tid; t < kT; t += BLOCK) tab[t] = kEmptyKey;
  __syncthreads();
  int fresh = 0;
  expand_row<float, BLOCK, false>(colA, nullptr, rowptrB, colB, nullptr, as_i, ae_i, sc,
                                  [&](int, uint32_t c, float) {
    uint32_t h = hash_slot<LOG_T>(c, NARROW);
    for (;;) {
      const uint32_t old = atomicCAS(&tab[h], kEmptyKey, c);
      if (old == kEmptyKey) {
        ++fresh;
        break;
      }
      if (old == c) break;
      h = (h + 1) & (kT - 1);
    }
  });
  for (int off = 32; off > 0; off >>= 1) fresh += lane_xor(fresh, off);
  if constexpr (BLOCK == 64) {
    if (tid == 0) nnzC[i] = fresh;
  } else {
    if ((tid & 63) == 0) s_cnt[tid >> 6] = fresh;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
#pragma unroll
      for (int w = 0; w < BLOCK / 64; ++w) t += s_cnt[w];
      nnzC[i] = t;
    }
  }
}

// ---------------------------------------------------------------------------
// Persistent one-wave kernels over the small rows, software-pipelined across rows.
//
// A row costs a chain of dependent round trips before its first product can be gathered: prod / rowptrA (uniform)
// -> colA -> rowptrB -> colB.  With one row per workgroup (rounds 1-3) every wave sat through that chain alone:
// after the owner scan took the instruction count down, the SQ counters showed the symbolic kernel 61 % of its
// time parked on a wait with 46 % of the VALU slots used (profiles/r04_sq_counters.md) -- at 8 waves per SIMD the
// device holds ~8 k rows in flight, and 500 k rows x 4 round trips / 8 k is most of the kernel's time.
// Here a wave takes rows blockIdx.x, + gridDim.x, ... and keeps THREE future rows in flight: at the top of every
// iteration it issues rowptrB of the next row (whose colA arrived during the previous iteration), colA (+ valA)
// of the one after (whose rowptrA arrived ...) and prod / rowptrA (/ rowptrC) of the third -- one stage per row,
// all in the same round trip as the current row's own gathers.
// ---------------------------------------------------------------------------
// MEASURED NEGATIVE (round 4, same box, profiles/r04_ab_spspmm_row_pipe.log): config 4 1.48 -> 1.80-1.82 ms (symbolic
// 448 -> 532 us, numeric 844 -> 1085 us under the counter pass).  The per-row time of a wave did not move (7.3 -> 7.6 us):
// what the SQ counters report as "waiting" is the row's LDS traffic and the LDS / VALU pipes shared with the other 7
// waves of the SIMD (VALU 3.4 us + LDS ~2-3 us of a 7.3 us round at 8 waves per SIMD), not the four global round
// trips -- those are already covered by the other waves.  On top, a persistent launch of 28 one-wave workgroups per CU
// does not fit the numeric kernel's 5.9 KB of LDS 28 times (27 do): the 28th waits for a whole wave-lifetime.  Kept
// behind the macro (default OFF) as the record of the experiment.
#ifndef TSAMD_SPSPMM_ROW_PIPE
#define TSAMD_SPSPMM_ROW_PIPE 0
#endif
#ifndef TSAMD_SPSPMM_PIPE_WAVES
#define TSAMD_SPSPMM_PIPE_WAVES 28  // workgroups (= waves) per CU of the persistent launch (LDS: ~5.5-6 KB each of 160 KB)
#endif

#if TSAMD_SPSPMM_ROW_PIPE
template <typename T, bool WITH_VAL>
struct RowPipe {
  using A = typename Traits<T>::acc_t;
  // uniform per row: row id, entries [as, ae), products (RAW: whether the row is a small one is only looked at an
  // iteration after the load was issued -- nothing in issue() may consume what it just asked for), output position
  struct Head {
    int64_t row, as, ae, out, pp;
    __device__ __forceinline__ int p() const { return (pp > 0 && pp <= kSmallCap) ? (int)pp : 0; }
  };
  const int64_t *rowptrA, *colA, *rowptrB, *prod, *rowptrC;
  const T *valA;
  int64_t M, stride;
  int lane;
  Head h0, h1, h2;          // current row, next, the one after
  uint32_t c1 = 0, c2 = 0;  // first 64 column ids of the rows h1 / h2 (per lane)
  A av0 = A(1), av1 = A(1), av2 = A(1);
  int64_t bs0 = 0;
  int d0 = 0;

  __device__ __forceinline__ Head fetch_head(int64_t r) const {  // stage 1: uniform loads
    Head h{r, 0, 0, 0, 0};
    if (r < M) {
      h.pp = prod[r];
      h.as = rowptrA[r];
      h.ae = rowptrA[r + 1];
      if (rowptrC != nullptr) h.out = rowptrC[r];
    }
    return h;
  }
  __device__ __forceinline__ void fetch_cols(const Head &h, uint32_t &c, A &av) const {  // stage 2
    c = 0;
    av = A(1);
    const int64_t e = h.as + lane;
    // (only dwords that stay live until rotate() are asked for: a dead half of a 64-bit load is a register the
    // allocator hands out again at once, and writing it waits for the load -- the prefetch would stall at issue)
    if (h.p() != 0 && e < h.ae) {
      c = *reinterpret_cast<const uint32_t *>(colA + e);  // low half: column ids are < 2^32
      if (WITH_VAL && valA != nullptr) av = Traits<T>::to_acc(valA[e]);
    }
  }
  __device__ __forceinline__ void fetch_brow(const Head &h, uint32_t c, int64_t &bs, uint32_t &be_lo) const {  // stage 3
    bs = 0;
    be_lo = 0;
    if (h.p() != 0 && h.as + lane < h.ae) {
      bs = rowptrB[c];
      be_lo = *reinterpret_cast<const uint32_t *>(rowptrB + (int64_t)c + 1);  // a B row has < 2^31 entries
    }
  }
  __device__ __forceinline__ void start(int64_t first) {
    h0 = fetch_head(first);
    h1 = fetch_head(first + stride);
    h2 = fetch_head(first + 2 * stride);
    uint32_t c0;
    fetch_cols(h0, c0, av0);
    fetch_cols(h1, c1, av1);
    uint32_t be0;
    fetch_brow(h0, c0, bs0, be0);
    d0 = (int)(be0 - (uint32_t)bs0);
  }
  // top of an iteration: one stage for each of the three rows behind the current one (loads only)
  struct Next {
    Head h3;
    int64_t bs1;
    uint32_t be1;
  };
  __device__ __forceinline__ Next issue() {
    Next n;
    fetch_brow(h1, c1, n.bs1, n.be1);
    fetch_cols(h2, c2, av2);
    n.h3 = fetch_head(h2.row + stride);
    return n;
  }
  __device__ __forceinline__ void rotate(const Next &n) {
    h0 = h1;
    bs0 = n.bs1;
    d0 = (int)(n.be1 - (uint32_t)n.bs1);
    av0 = av1;
    h1 = h2;
    c1 = c2;
    av1 = av2;
    h2 = n.h3;
  }
};

__global__ __launch_bounds__(64) void spspmm_symbolic_small_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const int64_t *__restrict__ rowptrB, const uint32_t *__restrict__ colB,
    const int64_t *__restrict__ prod, int64_t M, int64_t *__restrict__ nnzC) {
  constexpr int LOG_T = 10, kT = 1 << LOG_T;
  __shared__ alignas(16) uint32_t tab[kT];
  __shared__ ExpandScratch<float> sc;
  const int lane = (int)threadIdx.x;
  RowPipe<float, false> pipe{rowptrA, colA, rowptrB, prod, nullptr, nullptr, M, (int64_t)gridDim.x, lane};
  pipe.start((int64_t)blockIdx.x);
  while (pipe.h0.row < M) {
    const auto nxt = pipe.issue();
    if (pipe.h0.p() != 0) {  // wave-uniform
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int t = 0; t < kT / 256; ++t)
        *reinterpret_cast<u32x4 *>(tab + 4 * (lane + 64 * t)) = u32x4{kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey};
      __syncthreads();
      int fresh = 0;
      auto insert = [&](int, uint32_t c, float) {
        uint32_t h = hash_slot<LOG_T>(c, false);
        for (;;) {
          const uint32_t old = atomicCAS(&tab[h], kEmptyKey, c);
          if (old == kEmptyKey) {
            ++fresh;
            break;
          }
          if (old == c) break;
          h = (h + 1) & (kT - 1);
        }
      };
      if (pipe.h0.ae - pipe.h0.as <= 64)  // wave-uniform; the prefetched chunk is the whole row
        expand_row_wave<float, false, true>(colA, nullptr, rowptrB, colB, nullptr, pipe.h0.as, pipe.h0.ae, sc,
                                            each_product(insert), pipe.bs0, pipe.d0, 1.0f);
      else
        expand_row_wave<float, false, false>(colA, nullptr, rowptrB, colB, nullptr, pipe.h0.as, pipe.h0.ae, sc,
                                             each_product(insert));
      fresh = (int)wave_scan_add_dpp((uint32_t)fresh);
      if (lane == 63) nnzC[pipe.h0.row] = fresh;
      __syncthreads();
    }
    pipe.rotate(nxt);
  }
}

#endif  // TSAMD_SPSPMM_ROW_PIPE

// ---------------------------------------------------------------------------
// wave-level bitonic sort of 64 * I unique 32-bit keys held in registers, element
// e = lane * I + j.  "Flip + butterfly" form: every compare-exchange leaves the minimum at the
// lower position, so no direction bits are needed; strides below I stay inside a lane
// (v_min / v_max on registers), the others exchange with lane ^ mask through DPP (masks 1, 2, 3,
// 7, 15), ds_swizzle (4, 8, 16, 31) or ds_bpermute (32, 63) -- none of which allocates LDS.
// ---------------------------------------------------------------------------
#ifndef TSAMD_SPSPMM_DPP_XOR8
#define TSAMD_SPSPMM_DPP_XOR8 1
#endif
#ifndef TSAMD_SPSPMM_DPP_XOR4
#define TSAMD_SPSPMM_DPP_XOR4 0
#endif
template <int MASK>
__device__ __forceinline__ uint32_t xor_lane(uint32_t v, int lane) {
  if constexpr (MASK == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  else if constexpr (MASK == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  else if constexpr (MASK == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, true);
  else if constexpr (MASK == 7) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
  else if constexpr (MASK == 15) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
#if TSAMD_SPSPMM_DPP_XOR8
  // lane ^ 8 inside a row of 16 lanes is a rotation by 8: row_ror:8 -- a VALU move instead of an LDS-pipe ds_swizzle
  // (12 of the 32 LDS-pipe exchanges of a 256-key sort; the pipe is shared by every wave of the CU)
  else if constexpr (MASK == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);
#endif
#if TSAMD_SPSPMM_DPP_XOR4
  // lane ^ 4: banks 0 / 2 of a row read four lanes up (row_shl:4), banks 1 / 3 four lanes down (row_shr:4): two DPP
  // moves under bank masks instead of one ds_swizzle
  else if constexpr (MASK == 4) {
    const int up = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(up, (int)v, 0x114, 0xF, 0xA, false);
  }
#endif
  else if constexpr (MASK < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (MASK << 10) | 0x1F);
  else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, (int)v);
}

// One half of a compare-exchange between two lanes: the lower lane keeps min(a, b), the upper one
// max(a, b).  The median of (a, b, 0) is the minimum and the median of (a, b, 2^32 - 1) the maximum, so a
// single v_med3_u32 with a per-lane constant does it (min + max + select otherwise).
__device__ __forceinline__ uint32_t keep_lower_or_upper(uint32_t a, uint32_t b, uint32_t bound) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(bound));
  return r;
}

__device__ __forceinline__ void cmpswap(uint32_t &a, uint32_t &b) {
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}

template <int I>
__device__ __forceinline__ void intra_butterflies(uint32_t (&key)[I], int first_stride) {
#pragma unroll
  for (int s = I / 2; s >= 1; s >>= 1) {
    if (s > first_stride) continue;
#pragma unroll
    for (int j = 0; j < I; ++j)
      if ((j & s) == 0) cmpswap(key[j], key[j | s]);
  }
}

// one merge phase whose blocks span 2^B lanes (k = I * 2^B)
template <int I, int B>
__device__ __forceinline__ void cross_phase(uint32_t (&key)[I], int lane) {
  {  // flip: partner element = e ^ (k - 1): lane ^ (2^B - 1), item I - 1 - j
    const bool lower = ((lane >> (B - 1)) & 1) == 0;
    uint32_t p[I];
#pragma unroll
    for (int j = 0; j < I; ++j) p[j] = xor_lane<(1 << B) - 1>(key[I - 1 - j], lane);
#pragma unroll
    for (int j = 0; j < I; ++j) key[j] = keep_lower_or_upper(key[j], p[j], lower ? 0u : 0xFFFFFFFFu);
  }
  auto butterfly = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if constexpr (t <= B - 2) {
      const bool lower = ((lane >> t) & 1) == 0;
#pragma unroll
      for (int j = 0; j < I; ++j) {
        const uint32_t q = xor_lane<(1 << t)>(key[j], lane);
        key[j] = keep_lower_or_upper(key[j], q, lower ? 0u : 0xFFFFFFFFu);
      }
    }
  };
  butterfly(std::integral_constant<int, 4>{});
  butterfly(std::integral_constant<int, 3>{});
  butterfly(std::integral_constant<int, 2>{});
  butterfly(std::integral_constant<int, 1>{});
  butterfly(std::integral_constant<int, 0>{});
  intra_butterflies<I>(key, I / 2);
}

template <int I>
__device__ __forceinline__ void bitonic_sort_regs(uint32_t (&key)[I], int lane) {
  // merge phases inside a lane: k = 2 .. I
#pragma unroll
  for (int k = 2; k <= I; k <<= 1) {
#pragma unroll
    for (int j = 0; j < I; ++j) {
      const int jj = j ^ (k - 1);
      if (j < jj) cmpswap(key[j], key[jj]);
    }
    intra_butterflies<I>(key, k / 4);
  }
  cross_phase<I, 1>(key, lane);
  cross_phase<I, 2>(key, lane);
  cross_phase<I, 3>(key, lane);
  cross_phase<I, 4>(key, lane);
  cross_phase<I, 5>(key, lane);
  cross_phase<I, 6>(key, lane);
}

template <int I>
__device__ __forceinline__ void sort_lds_keys(uint32_t *skey, int lane) {
  uint32_t key[I];
  if constexpr (I >= 4) {
#pragma unroll
    for (int v = 0; v < I / 4; ++v) {
      const Pack<uint32_t, 4> k4 = *reinterpret_cast<const Pack<uint32_t, 4> *>(skey + lane * I + 4 * v);
#pragma unroll
      for (int j = 0; j < 4; ++j) key[4 * v + j] = k4.v[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < I; ++j) key[j] = skey[lane * I + j];
  }
  bitonic_sort_regs<I>(key, lane);
  __syncthreads();  // all reads of the unsorted keys are done (one wave: orders the LDS traffic)
  if constexpr (I >= 4) {
#pragma unroll
    for (int v = 0; v < I / 4; ++v) {
      Pack<uint32_t, 4> k4;
#pragma unroll
      for (int j = 0; j < 4; ++j) k4.v[j] = key[4 * v + j];
      *reinterpret_cast<Pack<uint32_t, 4> *>(skey + lane * I + 4 * v) = k4;
    }
  } else {
#pragma unroll
    for (int j = 0; j < I; ++j) skey[lane * I + j] = key[j];
  }
}

// Sum equal columns of the sorted row and store it at its final position.
// col_at(idx) / val_at(idx) read entry idx of the sorted row.
template <typename T, int BLOCK, typename ColAt, typename ValAt>
__device__ __forceinline__ void compress_and_store(int p, int64_t out, int64_t *__restrict__ colC,
                                                   T *__restrict__ valC, int *sscan, ColAt col_at,
                                                   ValAt val_at) {
  using A = typename Traits<T>::acc_t;
  const int tid = (int)threadIdx.x;
  int base = 0;
  for (int c0 = 0; c0 < p; c0 += BLOCK) {
    const int idx = c0 + tid;
    uint32_t c = 0;
    bool head = false;
    if (idx < p) {
      c = col_at(idx);
      head = idx == 0 || col_at(idx - 1) != c;
    }
    int tot, pos;
    bool alone = false;  // known without a read: the entry behind this head starts another column
    if constexpr (BLOCK == 64) {  // one wave: positions from the ballot of the heads
      const unsigned long long m = __ballot(head);
      pos = base + __popcll(m & ((1ull << (tid & 63)) - 1ull));
      tot = __popcll(m);
      alone = ((m >> 1) >> (tid & 63)) & 1ull;  // (lane 63: unknown, the loop below looks)
    } else {
      pos = base + block_exclusive_scan_small<BLOCK / 64>(head ? 1 : 0, sscan, &tot);
    }
    if (head) {
      colC[out + pos] = (int64_t)c;
      if (valC != nullptr) {
        A acc = val_at(idx);
        if (!alone)
          for (int q = idx + 1; q < p && col_at(q) == c; ++q) acc += val_at(q);
        valC[out + pos] = Traits<T>::from_acc(acc);
      }
    }
    base += tot;
  }
}

// ---------------------------------------------------------------------------
// numeric, small rows (<= 512 products), column ids below 2^23: one wave per row
// ---------------------------------------------------------------------------
constexpr int kIdxBits = 9;  // log2(kSmallCap)

template <typename T>
__global__ __launch_bounds__(64) void spspmm_numeric_small_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const T *__restrict__ valA, const int64_t *__restrict__ rowptrB,
    const uint32_t *__restrict__ colB, const T *__restrict__ valB,
    const int64_t *__restrict__ prod, const int64_t *__restrict__ rowptrC, int64_t *__restrict__ colC,
    T *__restrict__ valC) {
  using A = typename Traits<T>::acc_t;
  __shared__ alignas(16) uint32_t skey[kSmallCap];
  __shared__ alignas(32) A sval[kSmallCap];
  __shared__ ExpandScratch<A> sc;
  __shared__ int sscan[8];
  const int lane = (int)threadIdx.x;
  const int64_t i = blockIdx.x;
  // everything the row needs that depends on nothing but `i`, requested at once: the product count, the row of A and
  // the output position (which used to be asked for after the sort)
  const int64_t p64 = prod[i], as_i = rowptrA[i], ae_i = rowptrA[i + 1], out_i = rowptrC[i];
  if (p64 == 0 || p64 > kSmallCap) return;
  const int p = (int)p64;
  const bool with_val = valC != nullptr;
  const int items = p <= 64 ? 1 : (p <= 128 ? 2 : (p <= 256 ? 4 : 8));  // keys per lane
  for (int q = p + lane; q < 64 * items; q += 64) skey[q] = kEmptyKey;   // padding sorts last
#if TSAMD_SPSPMM_OWNER_SCAN
  __syncthreads();  // (one wave: the padding above is ordered before the packet stores below)
  // the lane's four consecutive products as ONE 16-byte store of keys (and one packet of values) when their slots
  // are a whole aligned group inside the arrays; slots past the row's end get the padding key again (they are padding:
  // nothing but a later chunk of the same row ever writes them, and that comes later in program order)
  auto put4 = [&](auto wv, int q0, int n, const uint32_t (&c)[4], const A (&v)[4]) __attribute__((always_inline)) {
    if ((q0 & 3) == 0 && q0 + 4 <= kSmallCap) {
      Pack<uint32_t, 4> k4;
#pragma unroll
      for (int u = 0; u < 4; ++u) k4.v[u] = u < n ? ((c[u] << kIdxBits) | (uint32_t)(q0 + u)) : kEmptyKey;
      *reinterpret_cast<Pack<uint32_t, 4> *>(__builtin_assume_aligned(skey + q0, 16)) = k4;
      if constexpr (decltype(wv)::value) {
        Pack<A, 4> v4;
#pragma unroll
        for (int u = 0; u < 4; ++u) v4.v[u] = v[u];
        *reinterpret_cast<Pack<A, 4> *>(__builtin_assume_aligned(sval + q0, 16)) = v4;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < n) {
          skey[q0 + u] = (c[u] << kIdxBits) | (uint32_t)(q0 + u);
          if constexpr (decltype(wv)::value) sval[q0 + u] = v[u];
        }
      }
    }
  };
  if (with_val)
    expand_row_wave<T, true>(colA, valA, rowptrB, colB, valB, as_i, ae_i, sc,
                             [&](int q0, int n, const uint32_t (&c)[4], const A (&v)[4]) { put4(std::true_type{}, q0, n, c, v); });
  else
    expand_row_wave<T, false>(colA, valA, rowptrB, colB, valB, as_i, ae_i, sc,
                              [&](int q0, int n, const uint32_t (&c)[4], const A (&v)[4]) { put4(std::false_type{}, q0, n, c, v); });
#else
  if (with_val) {
    expand_row<T, 64, true>(colA, valA, rowptrB, colB, valB, rowptrA[i], rowptrA[i + 1], sc,
                            [&](int q, uint32_t c, A v) {
      skey[q] = (c << kIdxBits) | (uint32_t)q;
      sval[q] = v;
    });
  } else {
    expand_row<T, 64, false>(colA, valA, rowptrB, colB, valB, rowptrA[i], rowptrA[i + 1], sc,
                             [&](int q, uint32_t c, A) { skey[q] = (c << kIdxBits) | (uint32_t)q; });
  }
#endif
  __syncthreads();
  if (items == 1) sort_lds_keys<1>(skey, lane);
  else if (items == 2) sort_lds_keys<2>(skey, lane);
  else if (items == 4) sort_lds_keys<4>(skey, lane);
  else sort_lds_keys<8>(skey, lane);
  __syncthreads();
  compress_and_store<T, 64>(
      p, out_i, colC, valC, sscan, [&](int idx) { return skey[idx] >> kIdxBits; },
      [&](int idx) { return sval[skey[idx] & (uint32_t)(kSmallCap - 1)]; });
}

#if TSAMD_SPSPMM_ROW_PIPE
// The same kernel, persistent and pipelined over the rows (RowPipe above).
template <typename T>
__global__ __launch_bounds__(64) void spspmm_numeric_small_pipe_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const T *__restrict__ valA, const int64_t *__restrict__ rowptrB,
    const uint32_t *__restrict__ colB, const T *__restrict__ valB,
    const int64_t *__restrict__ prod, const int64_t *__restrict__ rowptrC, int64_t M,
    int64_t *__restrict__ colC, T *__restrict__ valC) {
  using A = typename Traits<T>::acc_t;
  __shared__ alignas(16) uint32_t skey[kSmallCap];
  __shared__ A sval[kSmallCap];
  __shared__ ExpandScratch<A> sc;
  __shared__ int sscan[8];
  const int lane = (int)threadIdx.x;
  const bool with_val = valC != nullptr;
  auto row_body = [&](auto wv, RowPipe<T, decltype(wv)::value> &pipe) __attribute__((always_inline)) {
    constexpr bool kWV = decltype(wv)::value;
    pipe.start((int64_t)blockIdx.x);
    while (pipe.h0.row < M) {
      const auto nxt = pipe.issue();
      const int p = pipe.h0.p();
      if (p != 0) {  // wave-uniform
        const int items = p <= 64 ? 1 : (p <= 128 ? 2 : (p <= 256 ? 4 : 8));  // keys per lane
        for (int q = p + lane; q < 64 * items; q += 64) skey[q] = kEmptyKey;   // padding sorts last
        auto put = [&](int q, uint32_t c, A v) {
          skey[q] = (c << kIdxBits) | (uint32_t)q;
          if constexpr (kWV) sval[q] = v;
        };
        if (pipe.h0.ae - pipe.h0.as <= 64)  // wave-uniform; the prefetched chunk is the whole row
          expand_row_wave<T, kWV, true>(colA, valA, rowptrB, colB, valB, pipe.h0.as, pipe.h0.ae, sc, each_product(put),
                                        pipe.bs0, pipe.d0, pipe.av0);
        else
          expand_row_wave<T, kWV, false>(colA, valA, rowptrB, colB, valB, pipe.h0.as, pipe.h0.ae, sc, each_product(put));
        __syncthreads();
        if (items == 1) sort_lds_keys<1>(skey, lane);
        else if (items == 2) sort_lds_keys<2>(skey, lane);
        else if (items == 4) sort_lds_keys<4>(skey, lane);
        else sort_lds_keys<8>(skey, lane);
        __syncthreads();
        compress_and_store<T, 64>(
            p, pipe.h0.out, colC, valC, sscan, [&](int idx) { return skey[idx] >> kIdxBits; },
            [&](int idx) { return sval[skey[idx] & (uint32_t)(kSmallCap - 1)]; });
        __syncthreads();
      }
      pipe.rotate(nxt);
    }
  };
  if (with_val) {
    RowPipe<T, true> pipe{rowptrA, colA, rowptrB, prod, rowptrC, valA, M, (int64_t)gridDim.x, lane};
    row_body(std::true_type{}, pipe);
  } else {
    RowPipe<T, false> pipe{rowptrA, colA, rowptrB, prod, rowptrC, nullptr, M, (int64_t)gridDim.x, lane};
    row_body(std::false_type{}, pipe);
  }
}

#endif  // TSAMD_SPSPMM_ROW_PIPE

// ---------------------------------------------------------------------------
// numeric, (column, value) pairs sorted in LDS: medium rows (256 threads, bitonic) and small
// rows of matrices with more than 2^23 columns (one wave, stable LSD radix sort)
// ---------------------------------------------------------------------------
template <typename T, int BLOCK, int CAP>
__global__ __launch_bounds__(BLOCK) void spspmm_numeric_pairs_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const T *__restrict__ valA, const int64_t *__restrict__ rowptrB,
    const uint32_t *__restrict__ colB, const T *__restrict__ valB,
    const int64_t *__restrict__ prod, const int64_t *__restrict__ rows,
    const int64_t *__restrict__ rowptrC, int64_t *__restrict__ colC, T *__restrict__ valC, int passes) {
  using A = typename Traits<T>::acc_t;
  __shared__ uint32_t scol_[CAP];
  __shared__ A sval_[CAP];
  __shared__ uint32_t scol2_[BLOCK == 64 ? CAP : 1];  // ping-pong buffers of the wave radix sort
  __shared__ A sval2_[BLOCK == 64 ? CAP : 1];
  __shared__ uint32_t scnt[BLOCK == 64 ? 256 : 1];
  __shared__ ExpandScratch<A> sc;
  __shared__ int sscan[8];
  uint32_t *scol = scol_, *scol2 = scol2_;
  A *sval = sval_, *sval2 = sval2_;
  const int tid = (int)threadIdx.x;
  int64_t i;
  if constexpr (BLOCK == 64) {
    i = blockIdx.x;
    if (prod[i] == 0 || prod[i] > kSmallCap) return;
  } else {
    i = rows[blockIdx.x];
  }
  const int p = (int)prod[i];
  expand_row<T, BLOCK, true>(colA, valA, rowptrB, colB, valB, rowptrA[i], rowptrA[i + 1], sc,
                             [&](int q, uint32_t c, A v) {
    scol[q] = c;
    sval[q] = v;
  });
  if constexpr (BLOCK == 64) {
    wave_radix_sort_lds<A>(scol, sval, scol2, sval2, p, passes, scnt);
  } else {
    int n2 = 2;
    while (n2 < p) n2 <<= 1;
    for (int j = p + tid; j < n2; j += BLOCK) {
      scol[j] = 0xFFFFFFFFu;
      sval[j] = A(0);
    }
    __syncthreads();
    // bitonic sort by column (pairs); equal columns keep no particular order among themselves
    // beyond what the network does, which is a fixed function of their positions
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
  compress_and_store<T, BLOCK>(
      p, rowptrC[i], colC, valC, sscan, [&](int idx) { return scol[idx]; },
      [&](int idx) { return sval[idx]; });
}

// ---------------------------------------------------------------------------
// large rows (more than kMediumCap products): binned dense accumulation.
//
// Power-law operands put most of their products into such rows (A * A^T of an R-MAT graph: a hub row
// has 10^5 .. 10^8 products, many of them on the same few hub columns), where neither a sort nor a
// per-row hash table in LDS fits and where sorting the expansion in HBM costs ~100 bytes of scratch per
// product.  Here the column space is cut into ranges of kRangeCols columns -- one range of fp32 / fp64
// accumulators plus an occupancy bitmap is what LDS holds -- and a row is processed range by range:
//   hist      workgroup per large row: products per (row, range), counted in LDS while expanding
//   (scan)    -> bin offsets: the products of (row, range) get a contiguous segment of a scratch array
//   bin       workgroup per large row: expand again, every product (column inside the range [, value])
//             goes to its segment (LDS cursors); 4 + sizeof(T) bytes of scratch per product
//   classify  bins of at most 512 products | bigger ones (two lists, one atomic per wave)
//   count     (symbolic) small bins: one wave each, occupancy bitmap of the range in LDS, popcount;
//             big bins: persistent workgroups, same bitmap
//   accum     (numeric)  small bins: one wave each, the register sort of the small rows on
//             (column inside the range << 9 | index) keys -- bit-reproducible;
//             big bins: persistent workgroups, atomicAdd into the LDS accumulators, then walk the bitmap
//             in column order, store (column, sum) at the bin's final position and put the touched
//             accumulators back to zero
// Duplicates cost nothing extra, the output comes out sorted by construction, explicit zeros are kept
// (an entry exists when its bit is set).  The sums of a big bin are formed in atomic order: unlike the
// small / medium rows they are not bit-reproducible from run to run (fp32 rounding order).
// Limit: kMaxRanges ranges, i.e. N <= 2^26 (fp32) / 2^25 (fp64) columns for operands that have such rows.
// ---------------------------------------------------------------------------
// Range size / workgroup shape of the dense accumulation, same-box A/B on the R-MAT scale-19 stress product
// (whole op): 2^15 columns x 1024 threads x 1 workgroup per CU 86 ms, 2^14 x 512 x 2: 77 ms,
// 2^13 x 256 x 4: 72 ms, 2^12 x 256 x 8: 75 ms.  Smaller ranges send more bins down the one-wave path and
// keep several bins in flight per CU; below 2^13 the scatter of the bin kernel into more, shorter
// segments costs more than that gains.
#ifndef TSAMD_SPSPMM_LG_RANGE
#define TSAMD_SPSPMM_LG_RANGE 13
#endif
#ifndef TSAMD_SPSPMM_ACCUM_THREADS
#define TSAMD_SPSPMM_ACCUM_THREADS 256
#endif
#ifndef TSAMD_SPSPMM_ACCUM_WGS
#define TSAMD_SPSPMM_ACCUM_WGS 4
#endif
#ifndef TSAMD_SPSPMM_COUNT_WGS
#define TSAMD_SPSPMM_COUNT_WGS 8
#endif
constexpr int kLargeThreads = 256;   // hist / bin kernels: one workgroup per large row
constexpr int kAccumThreads = TSAMD_SPSPMM_ACCUM_THREADS;  // count / accum kernels (persistent, LDS-bound occupancy)
constexpr int kBinBatch = 4;         // bin entries fetched per thread before they are consumed
constexpr int kMaxRanges = 8192;         // LDS counters / cursors of the hist and bin kernels
constexpr int kOffLdsMax = 4096;         // ... up to this many, the bin kernel keeps their segment offsets in LDS as well

template <typename T>
constexpr int kLgRange = sizeof(typename Traits<T>::acc_t) == 4 ? TSAMD_SPSPMM_LG_RANGE : TSAMD_SPSPMM_LG_RANGE - 1;  // fp32 / fp64 columns per range

// Scratch layout of the binned products (round 5).  4-byte values: ONE array of (column, value) pairs -- the bin kernel
// stores 8 bytes per product with one instruction (two 4-byte stores into two arrays wrote 22 GB for 17 GB of payload
// on the stress product: a wave's run of a few products per range is a partial line in BOTH arrays) and the numeric
// kernels fetch a product with one 8-byte load; the symbolic kernels read the columns with stride 2.  8-byte values
// keep the two arrays (a 12-byte pair has no aligned store).
#ifndef TSAMD_SPSPMM_PAIRS
#define TSAMD_SPSPMM_PAIRS 1
#endif
template <typename T>
constexpr bool kPairs = TSAMD_SPSPMM_PAIRS && sizeof(T) == 4;
template <typename T>
__device__ __forceinline__ void bin_load(const uint32_t *__restrict__ bcol, const T *__restrict__ bval, int64_t p, bool want_val,
                                         uint32_t &c, typename Traits<T>::acc_t &v) {
  using A = typename Traits<T>::acc_t;
  if constexpr (kPairs<T>) {
    const uint2 e = reinterpret_cast<const uint2 *>(bcol)[p];
    c = e.x;
    T t;
    __builtin_memcpy(&t, &e.y, 4);
    v = want_val ? Traits<T>::to_acc(t) : A(0);
  } else {
    c = bcol[p];
    v = want_val ? Traits<T>::to_acc(bval[p]) : A(0);
  }
}

// Sub-bins (round 5: reproducible sums).  The four waves of a workgroup take a FIXED subset of a row's products
// (product q of a chunk goes to thread q mod 256), but their cursor atomics interleaved in arrival order, so the ORDER
// of a bin's products in the scratch changed from run to run -- and with it the rounding of every sum formed in that
// order.  Now every (row, range) bin is cut into `sub` = 4 consecutive segments, one per wave (counted separately
// here, reserved separately in the bin kernel): inside a wave the reservations happen in program order and the heads
// of one instruction are served in lane order, so a bin's content is the same sequence every run.  sub = 1 (more than
// kMaxRanges / 4 ranges: the counters would not fit LDS) keeps the shared cursors -- and the run-dependent order.
__global__ __launch_bounds__(kLargeThreads) void spspmm_large_hist_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA,
    const int64_t *__restrict__ rowptrB, const uint32_t *__restrict__ colB,
    const int64_t *__restrict__ rows, int lg_range, int nr, int sub, int64_t *__restrict__ hist) {
  // nr * sub counters, sized at launch (round 5): a static kMaxRanges array (32 KB for the 256 counters a 2^19-column
  // operand uses) held the kernel at 4 workgroups per CU, and at ~250-500 ns per product and thread the expansion is
  // bound by loads in flight (4 per thread), not by bytes -- see large_counter_bytes()
  extern __shared__ int cnt[];
  __shared__ ExpandScratch<float> sc;
  const int tid = (int)threadIdx.x;
  const int64_t i = rows[blockIdx.x];
  const int nc = nr * sub;
  for (int q = tid; q < nc; q += kLargeThreads) cnt[q] = 0;
  __syncthreads();
  // Consecutive products come from one sorted B row, so the 64 lanes of a wave form a few RUNS of equal range:
  // the head of every run adds the run's length with one LDS atomic (all heads in the same instruction, mostly
  // different counters) instead of 64 lanes hitting one counter -- the 2^13-column ranges of a 2^19-column
  // operand are 64 counters, per-product atomics serialise on them.  The active lanes of a step are a prefix of
  // the wave (q < total cuts a suffix), so "the lane below" is active for every active lane but lane 0.
  const int lane = tid & 63;
  const int wsel = sub == 1 ? 0 : (tid >> 6);
  expand_row<float, kLargeThreads, false, true>(colA, nullptr, rowptrB, colB, nullptr, rowptrA[i], rowptrA[i + 1], sc,
                                                [&](int, uint32_t c, float) {
    const int q = (int)(c >> lg_range);
    const int qprev = __builtin_amdgcn_update_dpp(0, q, 0x138, 0xF, 0xF, false);  // wave_shr:1 -- VALU, not the LDS pipe
    const bool head = lane == 0 || q != qprev;
    const unsigned long long hm = __ballot(head), act = __ballot(true);
    if (head) {
      const unsigned long long above = hm & ~((2ull << lane) - 1ull);
      const int next = above ? (int)__builtin_ctzll(above) : 64 - (int)__builtin_clzll(act);
      atomicAdd(&cnt[q * sub + wsel], next - lane);
    }
  });
  __syncthreads();
  for (int q = tid; q < nc; q += kLargeThreads) hist[(int64_t)blockIdx.x * nc + q] = cnt[q];
}

// bin_off[t] = offset of the first sub-bin of bin t (the consumers of the bins do not care about the cut)
__global__ __launch_bounds__(256) void spspmm_bin_offsets_kernel(const int64_t *__restrict__ sub_off, int sub,
                                                                int64_t ntask, int64_t *__restrict__ bin_off) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t <= ntask) bin_off[t] = sub_off[t * sub];
}

template <typename T, bool WITH_VAL>
__global__ __launch_bounds__(kLargeThreads) void spspmm_large_bin_kernel(
    const int64_t *__restrict__ rowptrA, const int64_t *__restrict__ colA, const T *__restrict__ valA,
    const int64_t *__restrict__ rowptrB, const uint32_t *__restrict__ colB, const T *__restrict__ valB,
    const int64_t *__restrict__ rows, int lg_range, int nr, int sub, const int64_t *__restrict__ sub_off,
    uint32_t *__restrict__ bcol, T *__restrict__ bval) {
  using A = typename Traits<T>::acc_t;
  extern __shared__ int cursor[];  // nr * sub cursors, sized at launch (see the hist kernel) | the segments' offsets
  __shared__ ExpandScratch<A> sc;
  const int tid = (int)threadIdx.x;
  const int64_t i = rows[blockIdx.x];
  const int nc = nr * sub;
  // the segment offsets of this row in LDS too: read from global memory inside the emit they were a dependent load
  // between the cursor atomic and the store of every product
  // (up to kOffLdsMax counters: 48 KB; beyond that they stay in global memory)
  const bool off_lds = nc <= kOffLdsMax;
  int64_t *off_s = reinterpret_cast<int64_t *>(cursor + ((nc + 1) & ~1));
  for (int q = tid; q < nc; q += kLargeThreads) {
    cursor[q] = 0;
    if (off_lds) off_s[q] = sub_off[(int64_t)blockIdx.x * nc + q];
  }
  __syncthreads();
  const int64_t *off = off_lds ? off_s : sub_off + (int64_t)blockIdx.x * nc;
  const int lane = tid & 63;
  const int wsel = sub == 1 ? 0 : (tid >> 6);  // this wave's segment of every bin (see the hist kernel)
  expand_row<T, kLargeThreads, WITH_VAL, true>(colA, valA, rowptrB, colB, valB, rowptrA[i], rowptrA[i + 1], sc,
                                               [&](int, uint32_t c, A v) {
    // runs of equal range inside the wave (see the hist kernel): the head of a run reserves the run's slots
    // with ONE returning atomic -- all heads in the same instruction -- and its lanes take consecutive
    // positions, so their stores are contiguous.  (Grouping ALL lanes of equal range instead needs one
    // dependent atomic per group: 11.8 -> 23.3 ms; per-product atomics put 64 lanes on one cursor.)
    const int q = (int)(c >> lg_range);
    const int qprev = __builtin_amdgcn_update_dpp(0, q, 0x138, 0xF, 0xF, false);  // wave_shr:1 -- VALU, not the LDS pipe
    const bool head = lane == 0 || q != qprev;
    const unsigned long long hm = __ballot(head), act = __ballot(true);
    const int leader = 63 - (int)__builtin_clzll(hm & ((2ull << lane) - 1ull));
    int base = 0;
    if (head) {
      const unsigned long long above = hm & ~((2ull << lane) - 1ull);
      const int next = above ? (int)__builtin_ctzll(above) : 64 - (int)__builtin_clzll(act);
      base = atomicAdd(&cursor[q * sub + wsel], next - lane);
    }
    base = lane_read(base, leader);
    const int64_t pos = off[q * sub + wsel] + base + (lane - leader);
    // the FULL column: small bins are merged across ranges (classify), the dense kernels mask
    if constexpr (kPairs<T>) {
      if (WITH_VAL) {
        const T t = Traits<T>::from_acc(v);
        uint2 e;
        e.x = c;
        __builtin_memcpy(&e.y, &t, 4);
        reinterpret_cast<uint2 *>(bcol)[pos] = e;
      } else {
        bcol[2 * pos] = c;
      }
    } else {
      bcol[pos] = c;
      if (WITH_VAL) bval[pos] = Traits<T>::from_acc(v);
    }
  });
}

// The persistent count / accum workgroups draw their bins from a global ticket counter (bins differ by
// orders of magnitude in size, and a fixed stride hands every hub-range bin to the same few workgroups:
// measured 46 -> 111 ms).  A bin costs four dependent round trips before its first product can be loaded
// (ticket -> list entry -> segment offsets / row -> row pointer), so they are taken through a two-deep
// pipeline: while bin k is processed, thread 0 draws the ticket of bin k+2 and fetches the descriptor of
// bin k+1 in three steps spread over the phases of bin k; the finished descriptor (task, segment, output
// position or row) waits in LDS for the next iteration.
#ifndef TSAMD_SPSPMM_TICKET_CHUNK
#define TSAMD_SPSPMM_TICKET_CHUNK 8
#endif
#ifndef TSAMD_SPSPMM_TICKET_CHUNK_ACCUM
#define TSAMD_SPSPMM_TICKET_CHUNK_ACCUM 2
#endif
constexpr int kTicketChunk = TSAMD_SPSPMM_TICKET_CHUNK;             // count kernel (256-thread workgroups)
constexpr int kTicketChunkAccum = TSAMD_SPSPMM_TICKET_CHUNK_ACCUM;  // accumulation (one wave per bin: coarser chunks unbalance it)
struct BinFeed {
  unsigned long long *queue;
  const int64_t *list;
  int64_t n;
  const int64_t *bin_off, *rows, *bin_pref, *rowptrC;  // bin_pref / rowptrC: NULL in the count kernel
  int nr;
  // thread 0 only
  int64_t t_next = 0, t_next2 = 0, task = -1, b0 = 0, b1 = 0, row = 0, pref = 0;
  // Tickets are drawn kTicketChunk at a time: the counter is ONE hot address, ~12 ns per returning atomic and fully
  // serialised (NOTES.md, hardware facts) -- 300 k big bins of the stress product were 3.6 ms of queueing in EACH of the
  // count and the accumulation kernels, whatever the number of workgroups (8 instead of 4 per CU changed nothing).
  int chunk = kTicketChunk;
  int64_t t_chunk = 0;
  int t_left = 0;
  __device__ __forceinline__ int64_t draw() {
    if (t_left == 0) {
      t_chunk = (int64_t)atomicAdd(queue, (unsigned long long)chunk);
      t_left = chunk;
    }
    return t_chunk + (chunk - t_left--);
  }

  __device__ __forceinline__ void step_a() {  // top of an iteration
    if (threadIdx.x != 0) return;
    t_next2 = draw();
    task = t_next < n ? list[t_next] : -1;
  }
  __device__ __forceinline__ void step_b() {  // middle of an iteration
    if (threadIdx.x != 0 || task < 0) return;
    b0 = bin_off[task];
    b1 = bin_off[task + 1];
    const int64_t r = task / nr;
    row = rows[r];
    if (bin_pref != nullptr) pref = bin_pref[task] - bin_pref[r * nr];
  }
  __device__ __forceinline__ void step_c(int64_t *desc) {  // end of an iteration (a barrier follows)
    if (threadIdx.x != 0) return;
    desc[0] = task;
    desc[1] = b0;
    desc[2] = b1;
    desc[3] = task < 0 ? 0 : (rowptrC != nullptr ? rowptrC[row] + pref : row);
    t_next = t_next2;
  }
  __device__ __forceinline__ void start(int64_t *desc) {  // descriptor of the first bin, ticket of the second
    if (threadIdx.x == 0) {
      t_next = draw();
      t_next2 = draw();
      task = t_next < n ? list[t_next] : -1;
    }
    step_b();
    step_c(desc);
    __syncthreads();
  }
};

// symbolic: distinct columns per bin, added up per row
__global__ __launch_bounds__(kAccumThreads) void spspmm_large_count_kernel(
    const int64_t *__restrict__ rows, int nr, const int64_t *__restrict__ big, const int64_t *__restrict__ n_big,
    const int64_t *__restrict__ bin_off, const uint32_t *__restrict__ bcol, int bstride, int range_words,
    int64_t *__restrict__ bin_cnt, unsigned long long *__restrict__ nnzC, unsigned long long *queue) {
  __shared__ uint32_t bits[(1 << 15) / 32];
  __shared__ int s_part[kAccumThreads / 64];
  const int tid = (int)threadIdx.x;
  __shared__ int64_t s_desc[2][4];
  for (int w = tid; w < range_words; w += kAccumThreads) bits[w] = 0;
  BinFeed feed{queue, big, *n_big, bin_off, rows, nullptr, nullptr, nr};
  feed.start(s_desc[0]);
  for (int cur = 0;; cur ^= 1) {
    const int64_t task = s_desc[cur][0];
    if (task < 0) break;
    const int64_t b0 = s_desc[cur][1], b1 = s_desc[cur][2], crow = s_desc[cur][3];
    feed.step_a();
    for (int64_t p0 = b0 + tid; p0 < b1; p0 += (int64_t)kAccumThreads * kBinBatch) {
      uint32_t c[kBinBatch];
#pragma unroll
      for (int u = 0; u < kBinBatch; ++u) {
        const int64_t p = p0 + (int64_t)u * kAccumThreads;
        c[u] = bcol[(p < b1 ? p : b1 - 1) * bstride] & (uint32_t)(range_words * 32 - 1);  // a repeated entry sets the same bit again
      }
#pragma unroll
      for (int u = 0; u < kBinBatch; ++u) atomicOr(&bits[c[u] >> 5], 1u << (c[u] & 31u));
    }
    __syncthreads();
    feed.step_b();
    int n = 0;
    for (int w = tid; w < range_words; w += kAccumThreads) {
      n += __popc(bits[w]);
      bits[w] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) n += lane_xor(n, off);
    if ((tid & 63) == 0) s_part[tid >> 6] = n;
    feed.step_c(s_desc[cur ^ 1]);
    __syncthreads();  // (also: the bitmap is clear before the next bin sets bits)
    if (tid == 0) {
      int tot = 0;
#pragma unroll
      for (int w = 0; w < kAccumThreads / 64; ++w) tot += s_part[w];
      bin_cnt[task] = tot;
      atomicAdd(&nnzC[crow], (unsigned long long)tot);
    }
  }
}

// numeric: accumulate a bin in LDS, emit it in column order at its final position
template <typename T>
__global__ __launch_bounds__(kAccumThreads) void spspmm_large_accum_kernel(
    const int64_t *__restrict__ rows, int nr, const int64_t *__restrict__ big, const int64_t *__restrict__ n_big,
    const int64_t *__restrict__ bin_off, const uint32_t *__restrict__ bcol, const T *__restrict__ bval,
    const int64_t *__restrict__ bin_pref,
    const int64_t *__restrict__ rowptrC, int64_t *__restrict__ colC, T *__restrict__ valC,
    unsigned long long *queue) {
  using A = typename Traits<T>::acc_t;
  constexpr int kCols = 1 << kLgRange<T>;
  constexpr int kWords = kCols / 32;  // 1024 (fp32) or 512 (fp64): at most one word per thread
  static_assert(kWords <= kAccumThreads, "one bitmap word per thread");
  __shared__ A acc[kCols];
  __shared__ uint32_t bits[kWords];
  __shared__ int sscan[kAccumThreads / 64];
  __shared__ int wpre[kWords];
  const int tid = (int)threadIdx.x;
  for (int c = tid; c < kCols; c += kAccumThreads) acc[c] = A(0);
  __shared__ int64_t s_desc[2][4];
  for (int w = tid; w < kWords; w += kAccumThreads) bits[w] = 0;
  BinFeed feed{queue, big, *n_big, bin_off, rows, bin_pref, rowptrC, nr};
  feed.chunk = kTicketChunkAccum;
  feed.start(s_desc[0]);
  for (int cur = 0;; cur ^= 1) {
    const int64_t task = s_desc[cur][0];
    if (task < 0) break;
    const int64_t b0 = s_desc[cur][1], b1 = s_desc[cur][2], out0 = s_desc[cur][3];
    feed.step_a();
    for (int64_t p0 = b0 + tid; p0 < b1; p0 += (int64_t)kAccumThreads * kBinBatch) {
      uint32_t c[kBinBatch];
      A v[kBinBatch];
#pragma unroll
      for (int u = 0; u < kBinBatch; ++u) {
        const int64_t p = p0 + (int64_t)u * kAccumThreads;
        const bool ok = p < b1;
        bin_load<T>(bcol, bval, ok ? p : b1 - 1, ok && valC != nullptr, c[u], v[u]);  // a repeat adds zero
        c[u] &= (uint32_t)(kCols - 1);
      }
#pragma unroll
      for (int u = 0; u < kBinBatch; ++u) {
        atomicOr(&bits[c[u] >> 5], 1u << (c[u] & 31u));
        if (valC != nullptr) atomicAdd(&acc[c[u]], v[u]);
      }
    }
    __syncthreads();
    feed.step_b();
    const int64_t q = task % nr;
    // bitmap word t -> exclusive prefix of the set bits (thread t owns word t)
    const uint32_t wd = tid < kWords ? bits[tid] : 0u;
    int tot;
    const int pre = block_exclusive_scan_small<kAccumThreads / 64>(__popc(wd), sscan, &tot);
    if (tid < kWords) wpre[tid] = pre;
    __syncthreads();
    // lane = OUTPUT position, so that the stores are coalesced (a thread that walks its own word writes
    // 64 different cache lines per store instruction: measured 2x slower for the whole kernel): find the
    // word that holds the o-th set bit (binary search over the prefixes), then the bit inside the word
    const int64_t col0 = q << kLgRange<T>;
    for (int o = tid; o < tot; o += kAccumThreads) {
      int lo = 0, hi = kWords;  // last word whose prefix is <= o
#pragma unroll
      for (int step = 0; step < 10; ++step) {
        const int mid = (lo + hi) >> 1;
        if (mid < kWords && wpre[mid] <= o) lo = mid; else hi = mid;
      }
      // (the LAST word with prefix <= o is the one that holds position o: empty words with the same
      // prefix lie before it, every later word has a larger prefix)
      const uint32_t word = bits[lo];
      int k = o - wpre[lo];
      int bit = 0;
#pragma unroll
      for (int sft = 16; sft >= 1; sft >>= 1) {
        const int c = __popc((word >> bit) & ((1u << sft) - 1u));
        if (k >= c) {
          k -= c;
          bit += sft;
        }
      }
      const int idx = lo * 32 + bit;
      colC[out0 + o] = col0 + idx;
      if (valC != nullptr) {
        valC[out0 + o] = Traits<T>::from_acc(acc[idx]);
        acc[idx] = A(0);
      }
    }
    feed.step_c(s_desc[cur ^ 1]);
    __syncthreads();
    if (wd) bits[tid] = 0;
    __syncthreads();  // the bitmap is clear before the next bin sets bits
  }
}

// numeric, big bins, REPRODUCIBLE form (round 5, the default when the bins were cut into per-wave segments): ONE wave per
// bin.  The bin's products are added in their scratch order -- ascending positions, 64 per LDS instruction: the
// instructions of one wave execute in program order and the lanes of one ds_add_f32 that hit the same accumulator are
// served in a fixed order -- so every sum is formed in the same order every run (the 256-thread kernel above lets four
// waves race on the accumulators).  A bin of 10^5 products keeps one wave busy for ~0.2 ms: the persistent
// grid draws bins from the ticket counter, so the others keep going.
#ifndef TSAMD_SPSPMM_WAVE_BATCH
#define TSAMD_SPSPMM_WAVE_BATCH 8
#endif
template <typename T>
__global__ __launch_bounds__(64) void spspmm_large_accum_wave_kernel(
    const int64_t *__restrict__ rows, int nr, const int64_t *__restrict__ big, const int64_t *__restrict__ n_big,
    const int64_t *__restrict__ bin_off, const uint32_t *__restrict__ bcol, const T *__restrict__ bval,
    const int64_t *__restrict__ bin_pref,
    const int64_t *__restrict__ rowptrC, int64_t *__restrict__ colC, T *__restrict__ valC,
    unsigned long long *queue) {
  using A = typename Traits<T>::acc_t;
  constexpr int kCols = 1 << kLgRange<T>;
  constexpr int kWords = kCols / 32;
  constexpr int kU = TSAMD_SPSPMM_WAVE_BATCH;
  static_assert(kWords % 64 == 0, "whole words per lane");
  __shared__ A acc[kCols];
  __shared__ uint32_t bits[kWords];
  __shared__ uint16_t stage[2048];
  __shared__ int64_t s_desc[2][4];
  const int lane = (int)threadIdx.x;
  for (int c = lane; c < kCols; c += 64) acc[c] = A(0);
  for (int w = lane; w < kWords; w += 64) bits[w] = 0;
  BinFeed feed{queue, big, *n_big, bin_off, rows, bin_pref, rowptrC, nr};
  feed.chunk = kTicketChunkAccum;
  feed.start(s_desc[0]);
  for (int cur = 0;; cur ^= 1) {
    const int64_t task = s_desc[cur][0];
    if (task < 0) break;
    const int64_t b0 = s_desc[cur][1], b1 = s_desc[cur][2], out0 = s_desc[cur][3];
    feed.step_a();
    for (int64_t p0 = b0; p0 < b1; p0 += 64 * kU) {
      uint32_t c[kU];
      A v[kU];
      bool ok[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t p = p0 + u * 64 + lane;
        ok[u] = p < b1;
        bin_load<T>(bcol, bval, ok[u] ? p : b1 - 1, ok[u] && valC != nullptr, c[u], v[u]);
        c[u] &= (uint32_t)(kCols - 1);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {  // instruction u holds positions p0 + 64 u .. + 63: ascending order overall
        if (ok[u]) {
          atomicOr(&bits[c[u] >> 5], 1u << (c[u] & 31u));
          if (valC != nullptr) atomicAdd(&acc[c[u]], v[u]);
        }
      }
    }
    __syncthreads();
    feed.step_b();
    // emit in column order, 64 bitmap words (2048 columns) at a time: lane L owns word L of the chunk, the DPP scan
    // gives the position of its first set bit among the chunk's, every lane drops the column offsets of ITS set bits
    // into a staging array at those positions (LDS scatter), and the chunk's entries are then stored with lane =
    // output position (coalesced) -- no search per output (the 256-thread kernel's 10-step binary search over the word
    // prefixes + 5-step bit select cost more than the accumulation itself once 64 lanes had to do it alone).
    const int64_t col0 = (task % nr) << kLgRange<T>;
    int emitted = 0;
#pragma unroll 1
    for (int ch = 0; ch < kWords / 64; ++ch) {
      uint32_t word = bits[ch * 64 + lane];
      const int pc = __popc(word);
      const uint32_t incl = wave_scan_add_dpp((uint32_t)pc);
      const int tot_ch = (int)lane_read(incl, 63);
      if (tot_ch == 0) continue;  // (wave-uniform)
      int pos = (int)incl - pc;
      if (word) bits[ch * 64 + lane] = 0;  // the bitmap is clear again before the next bin sets bits
      while (word) {
        const int bit = __builtin_ctz(word);
        word &= word - 1u;
        stage[pos++] = (uint16_t)(lane * 32 + bit);
      }
      __syncthreads();
      for (int o = lane; o < tot_ch; o += 64) {
        const int idx = ch * 2048 + (int)stage[o];
        colC[out0 + emitted + o] = col0 + idx;
        if (valC != nullptr) {
          valC[out0 + emitted + o] = Traits<T>::from_acc(acc[idx]);
          acc[idx] = A(0);
        }
      }
      emitted += tot_ch;
      __syncthreads();
    }
    feed.step_c(s_desc[cur ^ 1]);
    __syncthreads();
  }
}

// Bins by size.  Big bins (more than kSmallBinCap products) go to the persistent workgroups above.  The others
// are taken by one WAVE each (many in flight per CU) -- and CONSECUTIVE small bins of a row are merged into one
// group of at most kSmallBinCap products first: their products are contiguous in the scratch and the sort-based
// wave path does not care how many column ranges its keys span.  Most bins of a power-law product hold ~100
// products and the bins of a uniform product with a few thousand products per row ~60: a wave per such bin runs
// at 50 % lane utilisation at best and pays its descriptor loads per bin (uniform product: 4.9 M bins -> 0.4 M
// groups).  lists[0 .. ntask) = small groups, encoded first task | number of merged bins << 40;
// lists[ntask .. 2 ntask) = big bins; counts[0] / counts[1] their numbers.  bin_cnt of empty bins and of the
// bins merged into a group behind its first one is set to 0 here (the group's count lands on its first bin).
// One thread per large row, two passes over its nr bins; one atomic per workgroup and list.
// Merging stops at kGroupCap products.  Same-box (R-MAT stress / uniform product): (small-bin cap, group cap) =
// (1024, 1024): 43.1 / 7.6 ms; (1024, 512): 44.1 / 7.6 ms; (512, 512): 48.0 / 6.6 ms -- the uniform product gains from the
// smaller LDS footprint of the 512-key kernels, not from shorter sorts; the R-MAT product wants its bins of
// 513..1024 products off the persistent workgroups.
#ifndef TSAMD_SPSPMM_GROUP_CAP
#define TSAMD_SPSPMM_GROUP_CAP 1024
#endif
constexpr int kGroupCap = TSAMD_SPSPMM_GROUP_CAP < kSmallBinCap ? TSAMD_SPSPMM_GROUP_CAP : kSmallBinCap;
constexpr int kGroupShift = 40;
constexpr int kMaxGroupBins = 256;  // (column span of a group) << kBinIdxBits must fit 32 bits: 2^(13+8+10) = 2^31

__global__ __launch_bounds__(256) void spspmm_large_classify_kernel(const int64_t *__restrict__ bin_off,
                                                                   int64_t n_rows, int nr, int64_t ntask,
                                                                   int64_t *__restrict__ lists,
                                                                   unsigned long long *counts,
                                                                   int64_t *__restrict__ bin_cnt) {
  const int lane = (int)(threadIdx.x & 63), wid = (int)(threadIdx.x >> 6);
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t t0 = r * nr;
  // pass 1: how many groups / big bins does this row emit
  int n_small = 0, n_big = 0;
  if (r < n_rows) {
    int64_t cur = 0;
    int span = 0;
    for (int q = 0; q < nr; ++q) {
      const int64_t n = bin_off[t0 + q + 1] - bin_off[t0 + q];
      if (n > kSmallBinCap) {
        if (cur > 0) ++n_small;
        cur = 0;
        span = 0;
        ++n_big;
        continue;
      }
      if (cur > 0 && (cur + n > kGroupCap || span >= kMaxGroupBins)) {
        ++n_small;
        cur = 0;
        span = 0;
      }
      if (n > 0 || cur > 0) {
        cur += n;
        span += 1;
      }
    }
    if (cur > 0) ++n_small;
  }
  // block-level exclusive positions, one atomic per workgroup and list
  __shared__ int s_tot[2][4];
  __shared__ unsigned long long s_base[2];
  int inc_s = n_small, inc_b = n_big;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int a = lane_read(inc_s, lane >= off ? lane - off : lane), b = lane_read(inc_b, lane >= off ? lane - off : lane);
    if (lane >= off) {
      inc_s += a;
      inc_b += b;
    }
  }
  if (lane == 63) {
    s_tot[0][wid] = inc_s;
    s_tot[1][wid] = inc_b;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int tot = s_tot[threadIdx.x][0] + s_tot[threadIdx.x][1] + s_tot[threadIdx.x][2] + s_tot[threadIdx.x][3];
    s_base[threadIdx.x] = tot ? atomicAdd(&counts[threadIdx.x], (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  if (r >= n_rows) return;
  int64_t pos_s = (int64_t)s_base[0] + (inc_s - n_small), pos_b = ntask + (int64_t)s_base[1] + (inc_b - n_big);
  for (int w = 0; w < wid; ++w) {
    pos_s += s_tot[0][w];
    pos_b += s_tot[1][w];
  }
  // pass 2: the same walk, writing the entries
  int64_t cur = 0, first = -1;
  int span = 0;
  auto flush = [&]() {
    if (cur > 0) lists[pos_s++] = first | ((int64_t)span << kGroupShift);
    cur = 0;
    span = 0;
    first = -1;
  };
  for (int q = 0; q < nr; ++q) {
    const int64_t t = t0 + q;
    const int64_t n = bin_off[t + 1] - bin_off[t];
    if (n > kSmallBinCap) {
      flush();
      lists[pos_b++] = t;
      continue;
    }
    bin_cnt[t] = 0;  // empty, or merged behind the first bin of its group (which the count kernel overwrites)
    if (cur > 0 && (cur + n > kGroupCap || span >= kMaxGroupBins)) flush();
    if (n > 0 || cur > 0) {
      if (cur == 0) first = t;
      cur += n;
      span += 1;
    }
  }
  flush();
}

constexpr int kSmallBinWaves = 8192;  // resident waves of the small-bin kernels (grid-stride over the list)

// The descriptor of a small group is a chain of two dependent round trips (list entry -> the two bin offsets) in front
// of its first product.  The grid-stride kernels below take it through a two-deep pipeline in registers: while group
// t is processed, the entry of group t + 2 * grid and the offsets of group t + grid are on their way.
struct GroupFeed {
  const int64_t *small, *bin_off;
  int64_t ns, stride;
  int64_t e_nxt = -1, e_nn = -1;   // list entries of the next two groups of this wave (-1: none)
  int64_t task = 0, b0 = 0, b1 = 0;  // current group
  int nb = 0;
  int64_t n_task = 0, n_b0 = 0, n_b1 = 0;
  int n_nb = 0;
  __device__ __forceinline__ static void decode(int64_t entry, int64_t &task, int &nb) {
    task = entry & (((int64_t)1 << 40) - 1);
    nb = (int)(entry >> 40);
  }
  __device__ __forceinline__ bool start(int64_t t) {  // -> false: nothing for this wave
    if (t >= ns) return false;
    const int64_t e = small[t];
    e_nxt = t + stride < ns ? small[t + stride] : -1;
    decode(e, task, nb);
    b0 = bin_off[task];
    b1 = bin_off[task + nb];
    return true;
  }
  __device__ __forceinline__ void prefetch(int64_t t) {  // issue the loads of the groups behind group t
    e_nn = t + 2 * stride < ns ? small[t + 2 * stride] : -1;
    if (e_nxt >= 0) {
      decode(e_nxt, n_task, n_nb);
      n_b0 = bin_off[n_task];
      n_b1 = bin_off[n_task + n_nb];
    }
  }
  __device__ __forceinline__ void advance() {
    task = n_task;
    nb = n_nb;
    b0 = n_b0;
    b1 = n_b1;
    e_nxt = e_nn;
  }
};

// symbolic, small groups: distinct columns among the group's products -- an LDS hash set (the columns of a
// merged group span several ranges, a bitmap of them would not fit), one wave per group
constexpr int kGroupLogT = kSmallBinCap <= 512 ? 10 : (kSmallBinCap <= 1024 ? 11 : 12);

__global__ __launch_bounds__(64) void spspmm_smallbin_count_kernel(
    const int64_t *__restrict__ rows, int nr, const int64_t *__restrict__ small,
    const unsigned long long *__restrict__ n_small, const int64_t *__restrict__ bin_off,
    const uint32_t *__restrict__ bcol, int bstride, int64_t *__restrict__ bin_cnt, unsigned long long *__restrict__ nnzC) {
  constexpr int kT = 1 << kGroupLogT;
  __shared__ uint32_t tab[kT];
  const int lane = (int)threadIdx.x;
  for (int w = lane; w < kT; w += 64) tab[w] = kEmptyKey;
  const int64_t ns = (int64_t)*n_small;
  static_assert(kGroupShift == 40, "GroupFeed::decode");
  GroupFeed feed{small, bin_off, ns, (int64_t)gridDim.x};
  if (!feed.start(blockIdx.x)) return;
  for (int64_t t = blockIdx.x; t < ns; t += gridDim.x, feed.advance()) {
    feed.prefetch(t);
    const int64_t task = feed.task;
    const int64_t b0 = feed.b0;
    const int n = (int)(feed.b1 - b0);
    __syncthreads();  // (one wave: orders the table resets of the previous group)
    int fresh = 0;
    constexpr int kCU = 8;  // columns per lane in flight (a group of 1024 products: two round trips instead of four)
    for (int q0 = 0; q0 < n; q0 += 64 * kCU) {
      uint32_t c[kCU];
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const int q = q0 + u * 64 + lane;
        c[u] = bcol[(b0 + (q < n ? q : n - 1)) * bstride];
        if (q >= n) c[u] = kEmptyKey;
      }
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        if (c[u] == kEmptyKey) continue;
        uint32_t h = (c[u] * 0x9E3779B1u) >> (32 - kGroupLogT);
        for (;;) {
          const uint32_t old = atomicCAS(&tab[h], kEmptyKey, c[u]);
          if (old == kEmptyKey) {
            ++fresh;
            break;
          }
          if (old == c[u]) break;
          h = (h + 1) & (kT - 1);
        }
      }
    }
    for (int off = 32; off > 0; off >>= 1) fresh += lane_xor(fresh, off);
    __syncthreads();
    for (int w = lane; w < kT; w += 64) tab[w] = kEmptyKey;
    if (lane == 0) {
      bin_cnt[task] = fresh;
      atomicAdd(&nnzC[rows[task / nr]], (unsigned long long)fresh);
    }
  }
}

// numeric, small bins: the register sort of the small rows on (column inside the range << 9 | index) keys
template <typename T>
__global__ __launch_bounds__(64) void spspmm_smallbin_accum_kernel(
    const int64_t *__restrict__ rows, int nr, const int64_t *__restrict__ small,
    const unsigned long long *__restrict__ n_small, const int64_t *__restrict__ bin_off,
    const uint32_t *__restrict__ bcol, const T *__restrict__ bval, const int64_t *__restrict__ bin_pref,
    const int64_t *__restrict__ rowptrC, int64_t *__restrict__ colC, T *__restrict__ valC) {
  using A = typename Traits<T>::acc_t;
  __shared__ alignas(16) uint32_t skey[kSmallBinCap];
  __shared__ A sval[kSmallBinCap];
  __shared__ int sscan[8];
  const int lane = (int)threadIdx.x;
  const int64_t ns = (int64_t)*n_small;
  GroupFeed feed{small, bin_off, ns, (int64_t)gridDim.x};
  if (!feed.start(blockIdx.x)) return;
  for (int64_t t = blockIdx.x; t < ns; t += gridDim.x, feed.advance()) {
    feed.prefetch(t);
    const int64_t task = feed.task;
    const int64_t b0 = feed.b0;
    const int p = (int)(feed.b1 - b0);
    const uint32_t col_base = (uint32_t)((task % nr) << kLgRange<T>);
    const int items = p <= 64 ? 1 : (p <= 128 ? 2 : (p <= 256 ? 4 : (p <= 512 ? 8 : (p <= 1024 ? 16 : 32))));
    // where the group's output goes: asked for NOW (three dependent loads that used to start after the sort)
    const int64_t r = task / nr;
    const int64_t out_row = rows[r];
    const int64_t pref_t = bin_pref[task], pref_r = bin_pref[r * nr];
    // the group's products, four per lane in flight (one load per loop iteration was one round trip per 64 products:
    // 16 in a row for a full group)
    for (int q0 = lane; q0 < 64 * items; q0 += 64 * 4) {
      uint32_t cq[4];
      A vq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + 64 * u;
        bin_load<T>(bcol, bval, b0 + (q < p ? q : p - 1), valC != nullptr, cq[u], vq[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + 64 * u;
        if (q >= 64 * items) continue;
        uint32_t k = kEmptyKey;
        if (q < p) {
          k = ((cq[u] - col_base) << kBinIdxBits) | (uint32_t)q;
          if (valC != nullptr) sval[q] = vq[u];
        }
        skey[q] = k;
      }
    }
    const int64_t row_c = rowptrC[out_row];  // (on its way while the group is sorted)
    __syncthreads();
    if (items == 1) sort_lds_keys<1>(skey, lane);
    else if (items == 2) sort_lds_keys<2>(skey, lane);
    else if (items == 4) sort_lds_keys<4>(skey, lane);
    else if (items == 8) sort_lds_keys<8>(skey, lane);
    else if (items == 16) sort_lds_keys<16>(skey, lane);
    else if constexpr (kSmallBinCap > 1024) sort_lds_keys<32>(skey, lane);
    __syncthreads();
    const int64_t out0 = row_c + (pref_t - pref_r);
    const uint32_t col0 = col_base;
    compress_and_store<T, 64>(
        p, out0, colC, valC, sscan, [&](int idx) { return col0 + (skey[idx] >> kBinIdxBits); },
        [&](int idx) { return sval[skey[idx] & (uint32_t)(kSmallBinCap - 1)]; });
    __syncthreads();
  }
}

#ifndef TSAMD_SPSPMM_SUBBINS
#define TSAMD_SPSPMM_SUBBINS 1  // 0: shared cursors (round-4 behaviour: bin order and big-bin sums run-dependent), for A/B builds
#endif
#ifndef TSAMD_SPSPMM_WAVE_ACCUM
#define TSAMD_SPSPMM_WAVE_ACCUM 1  // 0: the 256-thread accumulation (sums in arrival order), for A/B builds
#endif
struct LargeWs {
  int64_t *hist;      // [n_large * nr * sub + 1] products per (bin, wave segment), scanned in place -> segment offsets
  int64_t *bin_off;   // [n_large * nr + 1] offset of every bin = of its first segment
  int sub;            // wave segments per bin: 4 (reproducible bin order) or 1 (too many ranges for the LDS counters)
  int64_t *bin_cnt;   // [n_large * nr + 1] distinct columns per bin, scanned at numeric time
  uint32_t *bcol;     // [P_large] column inside its range
  void *bval;         // [P_large] value (numeric stage)
  unsigned long long *queue;  // ticket counters of the persistent kernels
  int64_t *lists;             // [2 * ntask] small bins | big bins (spspmm_large_classify_kernel)
  unsigned long long *counts; // [2] their numbers
  void *scan_ws;
  int nr, lg_range;
  int64_t ntask;
};

// One layout for both value types of a call sequence: the range is the one of the value type.
size_t carve_large(void *base, int64_t n_large, int64_t P_large, int64_t N, size_t esize, LargeWs *w) {
  char *p = reinterpret_cast<char *>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) -> void * {
    void *r = p ? p + off : nullptr;
    off += align_up(bytes > 0 ? bytes : 1, 256);
    return r;
  };
  LargeWs l;
  l.lg_range = esize == 8 ? TSAMD_SPSPMM_LG_RANGE - 1 : TSAMD_SPSPMM_LG_RANGE;
  l.nr = (int)((N + ((int64_t)1 << l.lg_range) - 1) >> l.lg_range);
  if (l.nr < 1) l.nr = 1;
  l.ntask = n_large * (int64_t)l.nr;
  static const bool subbins = [] {  // TSAMD_SPSPMM_SUBBINS=0 in the environment: the round-4 shared cursors (A/B runs)
    const char *e = exp_env("TSAMD_SPSPMM_SUBBINS");
    return e ? e[0] != '0' : (TSAMD_SPSPMM_SUBBINS != 0);
  }();
  l.sub = (subbins && l.nr <= kMaxRanges / (kLargeThreads / 64)) ? kLargeThreads / 64 : 1;
  l.hist = (int64_t *)take(8 * (size_t)(l.ntask * l.sub + 1));
  l.bin_off = (int64_t *)take(8 * (size_t)(l.ntask + 1));
  l.bin_cnt = (int64_t *)take(8 * (size_t)(l.ntask + 1));
  if (TSAMD_SPSPMM_PAIRS && esize == 4) {  // (column, value) pairs in one array
    l.bcol = (uint32_t *)take(8 * (size_t)P_large);
    l.bval = nullptr;
  } else {
    l.bcol = (uint32_t *)take(4 * (size_t)P_large);
    l.bval = take(esize * (size_t)P_large);
  }
  l.queue = (unsigned long long *)take(64);
  l.lists = (int64_t *)take(16 * (size_t)l.ntask);
  l.counts = (unsigned long long *)take(64);
  l.scan_ws = take(scan_workspace_bytes(l.ntask * l.sub + 1));
  if (w) *w = l;
  return off;
}

#if TSAMD_SPSPMM_ROW_PIPE
unsigned int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  return (unsigned int)cus;
}

// one-wave workgroups of the pipelined small-row kernels: as many as the device holds at once, never more than rows
unsigned int pipe_blocks(int64_t M) {
  const int64_t cap = (int64_t)device_cus() * TSAMD_SPSPMM_PIPE_WAVES;
  return (unsigned int)(M < cap ? (M > 0 ? M : 1) : cap);
}
#endif

unsigned int persistent_blocks();
// one-wave workgroups of the reproducible accumulation: as many as the LDS of the device holds at once
template <typename T>
unsigned int persistent_wave_blocks() {
  const size_t lds = sizeof(typename Traits<T>::acc_t) * ((size_t)1 << kLgRange<T>) + 4 * (((size_t)1 << kLgRange<T>) / 32) + 4096 + 128;
  const unsigned int per_cu = (unsigned int)((size_t)160 * 1024 / lds);
  return persistent_blocks() / TSAMD_SPSPMM_ACCUM_WGS * (per_cu < 1 ? 1 : per_cu);
}

unsigned int persistent_blocks() {
  static int cus = 0;  // the LDS footprint allows TSAMD_SPSPMM_ACCUM_WGS workgroups per CU
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  return (unsigned int)cus * TSAMD_SPSPMM_ACCUM_WGS;
}

// dynamic LDS of the hist / bin kernels: their nr * sub counters (TSAMD_SPSPMM_STATIC_COUNTERS=1 in the environment asks
// for the round-4 footprint of kMaxRanges counters: same-box A/B runs of the occupancy effect)
static size_t large_counter_bytes(int nr, int sub) {
  static const bool fat = [] {
    const char *e = exp_env("TSAMD_SPSPMM_STATIC_COUNTERS");
    return e != nullptr && e[0] == '1';
  }();
  const size_t n = fat ? (size_t)kMaxRanges : (size_t)nr * (size_t)sub;
  // (the bin kernel keeps an int64 offset beside every cursor when there are at most kOffLdsMax of them; the hist kernel
  // uses the first part only)
  const size_t off_bytes = n <= (size_t)kOffLdsMax ? n * sizeof(int64_t) : 0;
  return (((n + 1) & ~(size_t)1) * sizeof(int) + off_bytes + 255) / 256 * 256;
}

// valA / valB given (either may be NULL): the products are binned WITH their values, so that the numeric
// stage does not have to expand the large rows a third time.
template <typename T>
int symbolic_large(const int64_t *rowptrA, const int64_t *colA, const void *valA, const int64_t *rowptrB,
                   const uint32_t *colB, const void *valB, bool with_values, const int64_t *rows,
                   int64_t n_large, int64_t P_large, int64_t N, int64_t *nnzC, void *workspace,
                   hipStream_t stream) {
  constexpr size_t esize = sizeof(T);
  LargeWs w;
  carve_large(workspace, n_large, P_large, N, esize, &w);
  if (w.nr > kMaxRanges) return TSAMD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(spspmm_large_hist_kernel, dim3((unsigned int)n_large), dim3(kLargeThreads),
                     large_counter_bytes(w.nr, w.sub), stream,
                     rowptrA, colA, rowptrB, colB, rows, w.lg_range, w.nr, w.sub, w.hist);
  TSAMD_LAUNCH_CHECK();
  TSAMD_HIP_TRY(hipMemsetAsync(w.hist + w.ntask * w.sub, 0, 8, stream));
  int st = exclusive_scan_i64(w.hist, w.hist, w.ntask * w.sub + 1, nullptr, w.scan_ws, stream);
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL(spspmm_bin_offsets_kernel, dim3((unsigned int)ceil_div(w.ntask + 1, 256)), dim3(256), 0, stream,
                     (const int64_t *)w.hist, w.sub, w.ntask, w.bin_off);
  TSAMD_LAUNCH_CHECK();
  if (with_values)
    hipLaunchKernelGGL((spspmm_large_bin_kernel<T, true>), dim3((unsigned int)n_large), dim3(kLargeThreads), large_counter_bytes(w.nr, w.sub),
                       stream, rowptrA, colA, reinterpret_cast<const T *>(valA), rowptrB, colB,
                       reinterpret_cast<const T *>(valB), rows, w.lg_range, w.nr, w.sub, (const int64_t *)w.hist,
                       w.bcol, reinterpret_cast<T *>(w.bval));
  else
    hipLaunchKernelGGL((spspmm_large_bin_kernel<T, false>), dim3((unsigned int)n_large), dim3(kLargeThreads), large_counter_bytes(w.nr, w.sub),
                       stream, rowptrA, colA, (const T *)nullptr, rowptrB, colB, (const T *)nullptr, rows,
                       w.lg_range, w.nr, w.sub, (const int64_t *)w.hist, w.bcol, (T *)nullptr);
  TSAMD_LAUNCH_CHECK();
  TSAMD_HIP_TRY(hipMemsetAsync(w.queue, 0, 64, stream));
  TSAMD_HIP_TRY(hipMemsetAsync(w.counts, 0, 64, stream));
  hipLaunchKernelGGL(spspmm_large_classify_kernel, dim3((unsigned int)ceil_div(n_large, 256)), dim3(256), 0, stream,
                     (const int64_t *)w.bin_off, n_large, w.nr, w.ntask, w.lists, w.counts, w.bin_cnt);
  TSAMD_LAUNCH_CHECK();
  const unsigned int small_grid = (unsigned int)(w.ntask < kSmallBinWaves ? w.ntask : kSmallBinWaves);
  hipLaunchKernelGGL(spspmm_smallbin_count_kernel, dim3(small_grid), dim3(64), 0, stream, rows, w.nr,
                     (const int64_t *)w.lists, (const unsigned long long *)w.counts, (const int64_t *)w.bin_off,
                     (const uint32_t *)w.bcol, kPairs<T> ? 2 : 1, w.bin_cnt,
                     reinterpret_cast<unsigned long long *>(nnzC));
  TSAMD_LAUNCH_CHECK();
  // (the count kernel's footprint is its 4 KB bitmap: 8 persistent workgroups per CU instead of the accumulation's 4 --
  // a bin costs ~8 us of dependent round trips and barriers whatever its size, so bins in flight are what it needs:
  // SQ counters, profiles/r05_sq_counters.md: VALU pipe 0.12, waiting 0.92)
  hipLaunchKernelGGL(spspmm_large_count_kernel, dim3(persistent_blocks() / TSAMD_SPSPMM_ACCUM_WGS * TSAMD_SPSPMM_COUNT_WGS), dim3(kAccumThreads), 0, stream, rows,
                     w.nr, (const int64_t *)(w.lists + w.ntask), (const int64_t *)(w.counts + 1),
                     (const int64_t *)w.bin_off, (const uint32_t *)w.bcol, kPairs<T> ? 2 : 1, (1 << w.lg_range) / 32, w.bin_cnt,
                     reinterpret_cast<unsigned long long *>(nnzC), w.queue);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

template <typename T>
int numeric_large(const int64_t *rowptrA, const int64_t *colA, const void *valA, const int64_t *rowptrB,
                  const uint32_t *colB, const void *valB, const int64_t *rows, int64_t n_large,
                  int64_t P_large, int64_t N, const int64_t *rowptrC, int64_t *colC, void *valC,
                  bool values_binned, void *workspace, hipStream_t stream) {
  LargeWs w;
  carve_large(workspace, n_large, P_large, N, sizeof(T), &w);
  if (w.nr > kMaxRanges) return TSAMD_ERR_UNSUPPORTED;
  T *bv = reinterpret_cast<T *>(w.bval);
  if (valC != nullptr && !values_binned) {  // the symbolic stage binned the columns only; the values follow the same offsets
    hipLaunchKernelGGL((spspmm_large_bin_kernel<T, true>), dim3((unsigned int)n_large), dim3(kLargeThreads), large_counter_bytes(w.nr, w.sub),
                       stream, rowptrA, colA, reinterpret_cast<const T *>(valA), rowptrB, colB,
                       reinterpret_cast<const T *>(valB), rows, w.lg_range, w.nr, w.sub, (const int64_t *)w.hist,
                       w.bcol, bv);
    TSAMD_LAUNCH_CHECK();
  }
  TSAMD_HIP_TRY(hipMemsetAsync(w.bin_cnt + w.ntask, 0, 8, stream));
  int st = exclusive_scan_i64(w.bin_cnt, w.bin_cnt, w.ntask + 1, nullptr, w.scan_ws, stream);
  if (st != TSAMD_OK) return st;
  TSAMD_HIP_TRY(hipMemsetAsync(w.queue, 0, 64, stream));
  // (the lists of small / big bins were left in the workspace by the symbolic stage)
  const unsigned int small_grid = (unsigned int)(w.ntask < kSmallBinWaves ? w.ntask : kSmallBinWaves);
  hipLaunchKernelGGL((spspmm_smallbin_accum_kernel<T>), dim3(small_grid), dim3(64), 0, stream, rows, w.nr,
                     (const int64_t *)w.lists, (const unsigned long long *)w.counts, (const int64_t *)w.bin_off,
                     (const uint32_t *)w.bcol, (const T *)bv, (const int64_t *)w.bin_cnt, rowptrC, colC,
                     reinterpret_cast<T *>(valC));
  TSAMD_LAUNCH_CHECK();
  if (w.sub > 1 && TSAMD_SPSPMM_WAVE_ACCUM)  // bins in a reproducible order: sum them in that order, one wave per bin
    hipLaunchKernelGGL((spspmm_large_accum_wave_kernel<T>), dim3(persistent_wave_blocks<T>()), dim3(64), 0, stream,
                       rows, w.nr, (const int64_t *)(w.lists + w.ntask), (const int64_t *)(w.counts + 1),
                       (const int64_t *)w.bin_off, (const uint32_t *)w.bcol, (const T *)bv, (const int64_t *)w.bin_cnt,
                       rowptrC, colC, reinterpret_cast<T *>(valC), w.queue);
  else
    hipLaunchKernelGGL((spspmm_large_accum_kernel<T>), dim3(persistent_blocks()), dim3(kAccumThreads), 0, stream,
                       rows, w.nr, (const int64_t *)(w.lists + w.ntask), (const int64_t *)(w.counts + 1),
                       (const int64_t *)w.bin_off, (const uint32_t *)w.bcol, (const T *)bv, (const int64_t *)w.bin_cnt,
                       rowptrC, colC, reinterpret_cast<T *>(valC), w.queue);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

template <typename T>
int numeric_rows(const int64_t *rowptrA, const int64_t *colA, const void *valA, const int64_t *rowptrB,
                 const uint32_t *colB, const void *valB, const int64_t *prod, const int64_t *bins,
                 int64_t M, int64_t N, int64_t n_medium, const int64_t *rowptrC, int64_t *colC, void *valC,
                 hipStream_t stream) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < N) ++bits;
  const int passes = (bits + 7) / 8;  // 8-bit radix passes over the column ids
  const T *va = reinterpret_cast<const T *>(valA);
  const T *vb = reinterpret_cast<const T *>(valB);
  T *vc = reinterpret_cast<T *>(valC);
  {  // small rows: all M rows in natural order, the kernel skips the others
    if (bits + kIdxBits <= 32) {
#if TSAMD_SPSPMM_ROW_PIPE
      hipLaunchKernelGGL((spspmm_numeric_small_pipe_kernel<T>), dim3(pipe_blocks(M)), dim3(64), 0, stream,
                         rowptrA, colA, va, rowptrB, colB, vb, prod, rowptrC, M, colC, vc);
#else
      hipLaunchKernelGGL((spspmm_numeric_small_kernel<T>), dim3((unsigned int)M), dim3(64), 0, stream,
                         rowptrA, colA, va, rowptrB, colB, vb, prod, rowptrC, colC, vc);
#endif
    } else
      hipLaunchKernelGGL((spspmm_numeric_pairs_kernel<T, 64, kSmallCap>), dim3((unsigned int)M), dim3(64),
                         0, stream, rowptrA, colA, va, rowptrB, colB, vb, prod, bins, rowptrC, colC, vc,
                         passes);
    TSAMD_LAUNCH_CHECK();
  }
  if (n_medium > 0) {
    hipLaunchKernelGGL((spspmm_numeric_pairs_kernel<T, 256, kMediumCap>), dim3((unsigned int)n_medium),
                       dim3(256), 0, stream, rowptrA, colA, va, rowptrB, colB, vb, prod, bins, rowptrC,
                       colC, vc, passes);
    TSAMD_LAUNCH_CHECK();
  }
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

extern "C" int tsamd_spspmm_plan(const int64_t *rowptrA, const int64_t *colA,
                                 const int64_t *rowptrB, const int64_t *colB64, int64_t nnzB, int64_t M,
                                 int64_t *prod, int64_t *bins, uint32_t *colB32, int64_t *stats,
                                 void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || !stats || nnzB < 0 || (nnzB > 0 && (!colB64 || !colB32))) return TSAMD_ERR_INVALID;
  if (nnzB > 0) {
    const int64_t blocks = ceil_div(nnzB, 256 * 8);
    hipLaunchKernelGGL(spspmm_narrow_cols_kernel, dim3((unsigned int)(blocks < 65536 ? blocks : 65536)), dim3(256),
                       0, stream, colB64, nnzB, colB32);
    TSAMD_LAUNCH_CHECK();
  }
  if (M > 0 && (!rowptrA || !rowptrB || !bins || !prod)) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(stats, 0, 8 * sizeof(int64_t), stream));
  if (M == 0) return TSAMD_OK;
  hipLaunchKernelGGL(spspmm_count_kernel, dim3((unsigned int)ceil_div(M * kCountLanes, 256)), dim3(256),
                     0, stream, rowptrA, colA, rowptrB, M, prod);
  TSAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(spspmm_bin_kernel, dim3((unsigned int)ceil_div(M, 256)), dim3(256), 0, stream,
                     (const int64_t *)prod, M, bins, reinterpret_cast<unsigned long long *>(stats));
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_spspmm_workspace_bytes(int dtype, int64_t n_large, int64_t P_large, int64_t N) {
  if (n_large <= 0) return 0;
  return carve_large(nullptr, n_large, P_large, N, dtype == TSAMD_F64 ? 8 : 4, nullptr);
}

extern "C" int tsamd_spspmm_symbolic(int dtype, const int64_t *rowptrA, const int64_t *colA,
                                     const void *valA, const int64_t *rowptrB, const uint32_t *colB,
                                     const void *valB, int bin_values, int64_t M, int64_t N,
                                     const int64_t *prod, const int64_t *bins,
                                     int64_t n_medium, int64_t n_large, int64_t P_large,
                                     int64_t *nnzC, void *workspace, size_t workspace_bytes,
                                     void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || N < 0 || N >= ((int64_t)1 << 32) - 1 || M >= ((int64_t)1 << 31)) return TSAMD_ERR_UNSUPPORTED;
  if (M == 0) return TSAMD_OK;
  // colA / colB may be NULL for operands without entries (every row then has zero products)
  if (!nnzC || !rowptrA || !rowptrB || !prod || !bins) return TSAMD_ERR_INVALID;
  if (n_medium < 0 || n_large < 0 || n_medium + n_large > M) return TSAMD_ERR_INVALID;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64) return TSAMD_ERR_UNSUPPORTED;
  if (n_large > 0 && (!workspace || workspace_bytes < tsamd_spspmm_workspace_bytes(dtype, n_large, P_large, N)))
    return TSAMD_ERR_WORKSPACE;
  TSAMD_HIP_TRY(hipMemsetAsync(nnzC, 0, sizeof(int64_t) * (size_t)M, stream));
  const bool narrow_cols = N <= ((int64_t)1 << 24);  // hash_slot: 24-bit multiply
#if TSAMD_SPSPMM_ROW_PIPE
  hipLaunchKernelGGL(spspmm_symbolic_small_kernel, dim3(pipe_blocks(M)), dim3(64), 0, stream, rowptrA, colA,
                     rowptrB, colB, prod, M, nnzC);
#else
  if (narrow_cols)
    hipLaunchKernelGGL((spspmm_symbolic_kernel<64, 10, true>), dim3((unsigned int)M), dim3(64), 0, stream, rowptrA,
                       colA, rowptrB, colB, prod, bins, nnzC);
  else
    hipLaunchKernelGGL((spspmm_symbolic_kernel<64, 10, false>), dim3((unsigned int)M), dim3(64), 0, stream, rowptrA,
                       colA, rowptrB, colB, prod, bins, nnzC);
#endif
  TSAMD_LAUNCH_CHECK();
  if (n_medium > 0) {
    if (narrow_cols)
      hipLaunchKernelGGL((spspmm_symbolic_kernel<256, kMediumLogT, true>), dim3((unsigned int)n_medium), dim3(256), 0,
                         stream, rowptrA, colA, rowptrB, colB, prod, bins, nnzC);
    else
      hipLaunchKernelGGL((spspmm_symbolic_kernel<256, kMediumLogT, false>), dim3((unsigned int)n_medium), dim3(256), 0,
                         stream, rowptrA, colA, rowptrB, colB, prod, bins, nnzC);
    TSAMD_LAUNCH_CHECK();
  }
  if (n_large > 0) {
    if (dtype == TSAMD_F64)
      return symbolic_large<double>(rowptrA, colA, valA, rowptrB, colB, valB, bin_values != 0, bins + M, n_large,
                                    P_large, N, nnzC, workspace, stream);
    return symbolic_large<float>(rowptrA, colA, valA, rowptrB, colB, valB, bin_values != 0, bins + M, n_large,
                                 P_large, N, nnzC, workspace, stream);
  }
  return TSAMD_OK;
}

extern "C" int tsamd_spspmm_numeric(int dtype, const int64_t *rowptrA, const int64_t *colA,
                                    const void *valA, const int64_t *rowptrB, const uint32_t *colB,
                                    const void *valB, int64_t M, int64_t N, const int64_t *prod,
                                    const int64_t *bins, int64_t n_medium, int64_t n_large,
                                    int64_t P_large, const int64_t *rowptrC, int64_t *colC,
                                    void *valC, int values_binned, void *workspace,
                                    size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64) return TSAMD_ERR_UNSUPPORTED;
  if (M < 0 || N < 0 || N >= ((int64_t)1 << 32) - 1 || M >= ((int64_t)1 << 31)) return TSAMD_ERR_UNSUPPORTED;
  if (M == 0) return TSAMD_OK;
  if (!rowptrA || !rowptrB || !prod || !bins || !rowptrC) return TSAMD_ERR_INVALID;
  if (n_medium < 0 || n_large < 0 || n_medium + n_large > M) return TSAMD_ERR_INVALID;
  if (n_large > 0 && (!workspace || workspace_bytes < tsamd_spspmm_workspace_bytes(dtype, n_large, P_large, N)))
    return TSAMD_ERR_WORKSPACE;
  int st;
  if (dtype == TSAMD_F32)
    st = numeric_rows<float>(rowptrA, colA, valA, rowptrB, colB, valB, prod, bins, M, N, n_medium, rowptrC,
                             colC, valC, stream);
  else
    st = numeric_rows<double>(rowptrA, colA, valA, rowptrB, colB, valB, prod, bins, M, N, n_medium, rowptrC,
                              colC, valC, stream);
  if (st != TSAMD_OK || n_large == 0) return st;
  if (dtype == TSAMD_F32)
    return numeric_large<float>(rowptrA, colA, valA, rowptrB, colB, valB, bins + M, n_large, P_large, N,
                                rowptrC, colC, valC, values_binned != 0, workspace, stream);
  return numeric_large<double>(rowptrA, colA, valA, rowptrB, colB, valB, bins + M, n_large, P_large, N,
                               rowptrC, colC, valC, values_binned != 0, workspace, stream);
}
