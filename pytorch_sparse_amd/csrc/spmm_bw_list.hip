// min / max backward, grad_mat by WINNER LISTS (round 4).  Part of tsamd_spmm_minmax_bw_csc (include/tsamd.h);
// replaces the ATen composition of SPMMMin/Max::backward for grad_mat (csrc/spmm.cpp:204-242, 264-302 of the
// reference: masked_fill / index_select / gather / scatter_add_).
//
// The round-3 pull (winner bit masks + the merge-path SpMM with a per-(entry, feature) predicate) gathers one
// whole grad_out row per entry although an entry wins K / deg of the K features of its row (5 % at configs[2]:
// 9.2 GB moved for 2.15 GB priced, at the streaming ceiling for what it moves).  Here the products themselves
// are compacted:
//   1. winlist_kernel (entry-balanced over the CSR: a wave owns 64 consecutive entries, like the round-3 record
//      kernel): for every row that intersects the chunk the winners of all features are read once (arg_out
//      row, L2-hot for hub rows), the chunk's entries collect their win masks in LDS, and the products
//          v = round_T(value[e] * grad_out[b, m, k])       (the reference rounds the product to the element type)
//      are written as (k, v) pairs into the row's K-slot segment of `pairs`, grouped by winning entry in
//      entry order -- every (b, m, k) with a winner exactly once, M K pairs at most.  Per entry a 16-byte
//      record (row, column, offset of its list in the row's segment, length).
//   2. listpull_kernel (position-balanced over the CSC: a wave owns 256 consecutive positions of the
//      column-major order): entry e = csr2csc[p] -> its record (one random 16-byte read) -> its list (one or two
//      64-byte segments instead of the 4 + 1 of a grad_out row and a mask record) -> fp32 / fp64 adds into a
//      per-wave LDS tile of the columns in flight -> every column that lies inside the wave's positions is
//      written once, rounded once; a column cut by a wave boundary leaves carry records (head / tail, as the
//      SpMM's merge kernel does) that
//   3. listfix_kernel folds in position order.
// Deterministic (fixed partition, fixed order of the LDS adds), no global atomics, grad_mat needs no memset
// beyond the one for columns without entries.
#if defined(TSAMD_EXPERIMENTS)  // measured slower than the mask route (profiles/r04_minmax_bw_routes.md): experiment builds only
#include "common.h"
#include "spmm_internal.h"

#include <type_traits>

namespace tsamd {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kTileWords = 1024;   // accumulator tile of the pull kernel per wave (elements of acc_t)
constexpr int kSubChunks = 4;      // 64-position sub-chunks per wave of the pull kernel

struct ListRec {  // one per (batch, CSR entry)
  uint32_t row, col;
  uint32_t off, cnt;  // the entry's pairs sit at pairs[(b * M + row) * K + off .. + cnt)
};
static_assert(sizeof(ListRec) == 16, "16-byte records");

// (feature id, product) pairs: 4 bytes for f16 / bf16, 8 for fp32, 16 for fp64
template <typename T>
struct PairOf {
  using type = uint32_t;
  __device__ static inline type pack(uint32_t k, T v) {
    uint16_t b;
    __builtin_memcpy(&b, &v, 2);
    return (k << 16) | (uint32_t)b;
  }
  __device__ static inline uint32_t key(type p) { return p >> 16; }
  __device__ static inline typename Traits<T>::acc_t val(type p) {
    T v;
    const uint16_t b = (uint16_t)(p & 0xFFFFu);
    __builtin_memcpy(&v, &b, 2);
    return Traits<T>::to_acc(v);
  }
};
template <>
struct PairOf<float> {
  using type = unsigned long long;
  __device__ static inline type pack(uint32_t k, float v) {
    uint32_t b;
    __builtin_memcpy(&b, &v, 4);
    return ((unsigned long long)k << 32) | b;
  }
  __device__ static inline uint32_t key(type p) { return (uint32_t)(p >> 32); }
  __device__ static inline float val(type p) {
    const uint32_t b = (uint32_t)p;
    float v;
    __builtin_memcpy(&v, &b, 4);
    return v;
  }
};
struct PairF64 {
  unsigned long long k;
  double v;
};
template <>
struct PairOf<double> {
  using type = PairF64;
  __device__ static inline type pack(uint32_t k, double v) { return PairF64{k, v}; }
  __device__ static inline uint32_t key(const type &p) { return (uint32_t)p.k; }
  __device__ static inline double val(const type &p) { return p.v; }
};

// ---------------------------------------------------------------------------
// 1. winner lists
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void winlist_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ col, const T *__restrict__ value,
    const int64_t *__restrict__ arg_out, const T *__restrict__ grad_out, ListRec *__restrict__ rec,
    typename PairOf<T>::type *__restrict__ pairs, int64_t B, int64_t M, uint32_t K, int64_t E) {
  using A = typename Traits<T>::acc_t;
  using PT = PairOf<T>;
  __shared__ uint32_t mask_[kWavesPerBlock][kWave * 4];   // win masks of the pass: 128 features per entry
  __shared__ uint32_t base_[kWavesPerBlock][kWave * 4];   // position of the first pair of (entry, mask word)
  __shared__ A w_[kWavesPerBlock][kWave];
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t *mask = mask_[wib];
  uint32_t *pbase = base_[wib];
  const int64_t e0 = ((int64_t)blockIdx.x * kWavesPerBlock + wib) * kWave;
  if (e0 >= E) return;
  const int n = (int)(E - e0 < kWave ? E - e0 : kWave);
  const uint32_t ntiles = (K + 63u) >> 6;
  const uint32_t npass = (ntiles + 1u) >> 1;  // 2 feature tiles (4 mask words) per pass
  const bool mine = lane < n;
  const uint32_t m_l = mine ? (uint32_t)row[e0 + lane] : 0xFFFFFFFFu;
  const uint32_t c_l = mine ? (uint32_t)col[e0 + lane] : 0u;
  const uint32_t m_prev = lane_read(m_l, lane > 0 ? lane - 1 : 0);
  const unsigned long long heads = __ballot(mine && (lane == 0 || m_l != m_prev));
  // the first row of the chunk may have started in an earlier chunk: its winners below e0 come first in the
  // row's segment
  const bool first_cut = e0 > 0 && (uint32_t)row[e0 - 1] == (uint32_t)__builtin_amdgcn_readfirstlane((int)m_l);
  A wv = A(1);
  if (mine && value != nullptr) wv = Traits<T>::to_acc(value[e0 + lane]);
  w_[wib][lane] = wv;
  const uint32_t bit = 1u << (lane & 31), half = (uint32_t)lane >> 5;
  // head lane of this lane's row (for the segmented prefix below)
  const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  const int head_lane = below ? 63 - __builtin_clzll(below) : 0;

  for (int64_t b = 0; b < B; ++b) {
    const int64_t *a_b = arg_out + (uint64_t)b * M * K;
    const T *g_b = grad_out + (uint64_t)b * M * K;
    typename PT::type *p_b = pairs + (uint64_t)b * M * K;
    uint32_t cnt_l = 0, before0 = 0, off_l = 0, filled_l = 0;
    // phase 0 counts every entry's winners over all passes; phase 1 writes the pairs (one pass: the masks of
    // phase 0 are still in LDS and are not built again)
    for (int phase = 0; phase < 2; ++phase) {
      for (uint32_t ps = 0; ps < npass; ++ps) {
        const uint32_t t0 = ps * 2u;
        const uint32_t k0 = t0 * 64u + (uint32_t)lane, k1 = k0 + 64u;
        if (!(phase == 1 && npass == 1)) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          *reinterpret_cast<u32x4 *>(mask + lane * 4) = u32x4{0u, 0u, 0u, 0u};
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          unsigned long long todo = heads;
          bool first = true;
          while (todo != 0) {  // two rows per step: their loads are independent
            const int p0 = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const bool two = todo != 0;
            const int p1 = two ? (int)__builtin_ctzll(todo) : p0;
            if (two) todo &= todo - 1;
            const uint64_t ra = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m_l, p0) * K;
            const uint64_t rb = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m_l, p1) * K;
            int64_t a00 = -1, a01 = -1, a10 = -1, a11 = -1;
            if (k0 < K) a00 = a_b[ra + k0];
            if (k1 < K) a01 = a_b[ra + k1];
            if (two && k0 < K) a10 = a_b[rb + k0];
            if (two && k1 < K) a11 = a_b[rb + k1];
            const int64_t r00 = a00 - e0, r01 = a01 - e0, r10 = a10 - e0, r11 = a11 - e0;
            if (a00 >= 0 && r00 >= 0 && r00 < n) atomicOr(mask + (uint32_t)r00 * 4 + half, bit);
            if (a01 >= 0 && r01 >= 0 && r01 < n) atomicOr(mask + (uint32_t)r01 * 4 + 2 + half, bit);
            if (a10 >= 0 && r10 >= 0 && r10 < n) atomicOr(mask + (uint32_t)r10 * 4 + half, bit);
            if (a11 >= 0 && r11 >= 0 && r11 < n) atomicOr(mask + (uint32_t)r11 * 4 + 2 + half, bit);
            if (phase == 0 && first && first_cut) {  // winners of the cut first row that precede the chunk
              before0 += (uint32_t)__popcll(__ballot(a00 >= 0 && a00 < e0)) + (uint32_t)__popcll(__ballot(a01 >= 0 && a01 < e0));
            }
            first = false;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        const uint32_t w0 = mask[lane * 4], w1 = mask[lane * 4 + 1], w2 = mask[lane * 4 + 2], w3 = mask[lane * 4 + 3];
        const uint32_t c0 = (uint32_t)__builtin_popcount(w0), c1 = (uint32_t)__builtin_popcount(w1),
                       c2 = (uint32_t)__builtin_popcount(w2), c3 = (uint32_t)__builtin_popcount(w3);
        if (phase == 0) {
          cnt_l += c0 + c1 + c2 + c3;
          continue;
        }
        // ---- phase 1: the pairs of this pass ----
        {
          const uint32_t s = off_l + filled_l;
          pbase[lane * 4] = s;
          pbase[lane * 4 + 1] = s + c0;
          pbase[lane * 4 + 2] = s + c0 + c1;
          pbase[lane * 4 + 3] = s + c0 + c1 + c2;
          filled_l += c0 + c1 + c2 + c3;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned long long todo = heads;
        while (todo != 0) {  // row by row (two per step), lane = feature: balanced whatever the entries win
          const int p0 = (int)__builtin_ctzll(todo);
          todo &= todo - 1;
          const bool two = todo != 0;
          const int p1 = two ? (int)__builtin_ctzll(todo) : p0;
          if (two) todo &= todo - 1;
          uint64_t rr[2];
          rr[0] = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m_l, p0) * K;
          rr[1] = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m_l, p1) * K;
          int64_t a[2][2];
          T gg[2][2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const uint32_t k = u == 0 ? k0 : k1;
              const bool on = k < K && (q == 0 || two);
              a[q][u] = on ? a_b[rr[q] + k] : -1;
              gg[q][u] = g_b[rr[q] + (on ? k : 0u)];
            }
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const uint32_t k = u == 0 ? k0 : k1;
              const int64_t r = a[q][u] - e0;
              if (a[q][u] < 0 || r < 0 || r >= n) continue;
              const uint32_t word = (uint32_t)u * 2u + half;
              const uint32_t mw = mask[(uint32_t)r * 4 + word];
              const uint32_t pos = pbase[(uint32_t)r * 4 + word] + (uint32_t)__builtin_popcount(mw & (bit - 1u));
              const A prod = w_[wib][r] * Traits<T>::to_acc(gg[q][u]);
              p_b[rr[q] + pos] = PT::pack(k, Traits<T>::from_acc(prod));
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (phase == 0) {
        // offset of the entry's list in its row's segment: winners of earlier entries of the same row
        uint32_t inc = cnt_l;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = lane_read(inc, lane >= off ? lane - off : lane);
          if (lane >= off) inc += o;
        }
        const uint32_t ex = inc - cnt_l;
        const uint32_t ex_head = lane_read(ex, head_lane);
        off_l = ex - ex_head + (head_lane == 0 ? before0 : 0u);
        if (mine) rec[(uint64_t)b * (uint64_t)E + (uint64_t)(e0 + lane)] = ListRec{m_l, c_l, off_l, cnt_l};
      }
    }
  }
}

// ---------------------------------------------------------------------------
// 2. pull over the CSC order
// ---------------------------------------------------------------------------
struct PullCarry {
  int64_t *head_col, *tail_col;  // [nwaves]  column whose partial is in head_val / tail_val, or -1
  void *head_val, *tail_val;     // [B][nwaves][K] acc_t
  int64_t nwaves;
};

template <typename T>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void listpull_kernel(
    const int64_t *__restrict__ colptr, const int64_t *__restrict__ perm, const ListRec *__restrict__ rec,
    const typename PairOf<T>::type *__restrict__ pairs, T *__restrict__ gmat, int64_t B, int64_t M, int64_t N,
    uint32_t K, int64_t E, PullCarry cy) {
  using A = typename Traits<T>::acc_t;
  using PT = PairOf<T>;
  using PR = typename PT::type;
  __shared__ A tile_[kWavesPerBlock][kTileWords];
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  A *tile = tile_[wib];
  const int64_t q = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t P0 = q * (kSubChunks * kWave);
  if (P0 >= E) return;
  const int64_t P1 = P0 + kSubChunks * kWave < E ? P0 + kSubChunks * kWave : E;
  const uint32_t R = kTileWords / K;  // columns in flight (the caller guarantees K <= kTileWords)
  A *head_val = reinterpret_cast<A *>(cy.head_val), *tail_val = reinterpret_cast<A *>(cy.tail_val);
  // the wave is a chain of dependent round trips (position -> entry -> record -> pairs): everything that does not
  // depend on an earlier load is requested up front, for all sub-chunks at once
  int64_t e_l[kSubChunks];
#pragma unroll
  for (int sc = 0; sc < kSubChunks; ++sc) {
    const int64_t p = P0 + (int64_t)sc * kWave + lane;
    e_l[sc] = p < P1 ? perm[p] : -1;
  }

  for (int64_t b = 0; b < B; ++b) {
    const ListRec *rec_b = rec + (uint64_t)b * (uint64_t)E;
    const PR *p_b = pairs + (uint64_t)b * M * K;
    T *g_b = gmat + (uint64_t)b * N * K;
    ListRec r_sc[kSubChunks];
#pragma unroll
    for (int sc = 0; sc < kSubChunks; ++sc) {
      r_sc[sc] = ListRec{0u, 0u, 0u, 0u};
      if (e_l[sc] >= 0) r_sc[sc] = rec_b[e_l[sc]];
    }
    // only the column of the wave's first position can have started before it
    const int64_t first_col = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)r_sc[0].col);
    const bool first_started_here = colptr[first_col] >= P0;
    int64_t end_sc[kSubChunks];  // first position behind the last column of each sub-chunk
#pragma unroll
    for (int sc = 0; sc < kSubChunks; ++sc) {
      const int64_t p0 = P0 + (int64_t)sc * kWave;
      const int n = p0 < P1 ? (int)(P1 - p0 < kWave ? P1 - p0 : kWave) : 0;
      end_sc[sc] = 0;
      if (n > 0) end_sc[sc] = colptr[(int64_t)(uint32_t)__builtin_amdgcn_readlane((int)r_sc[sc].col, n - 1) + 1];
    }
    int64_t open_col = -1;   // column whose partial sits in tile row 0 across sub-chunks
    int64_t hcol = -1;
#pragma unroll
    for (int sc = 0; sc < kSubChunks; ++sc) {
      const int64_t p0 = P0 + (int64_t)sc * kWave;
      if (p0 >= P1) break;
      const int n = (int)(P1 - p0 < kWave ? P1 - p0 : kWave);
      const ListRec r_l = r_sc[sc];
      const int64_t col_lo = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)r_l.col);
      const int64_t col_hi = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)r_l.col, n - 1);
      const int64_t end_hi = end_sc[sc];
      // the DISTINCT columns of the sub-chunk, densely numbered (a window over column ids would walk through the
      // empty columns in between: thousands of them in the sparse tail of a power-law matrix)
      const uint32_t col_prev = lane_read(r_l.col, lane > 0 ? lane - 1 : 0);
      const unsigned long long heads = __ballot(lane < n && (lane == 0 || r_l.col != col_prev));
      const int nd = (int)__popcll(heads);
      const int rank_l = (int)__popcll(heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))) - 1;
      unsigned long long hm = heads;  // consumed by the flush, one head per column
      (void)col_hi;
      for (int wb = 0; wb < nd; wb += (int)R) {
        const bool keep0 = wb == 0 && open_col == col_lo;  // row 0 carries the open column's partial
        for (uint32_t i = (keep0 ? K : 0u) + (uint32_t)lane; i < R * K; i += kWave) tile[i] = A(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 16 lanes per entry (an entry wins K / deg features: 6.4 on average at configs[2], more than 16 for 2 % of
        // them), 4 entries per step, 8 steps = 32 entries per round trip: two round trips per sub-chunk
        constexpr int kLpe = 16, kSteps = 8;
        const int g4 = lane >> 4, sub16 = lane & (kLpe - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          PR pr[kSteps];
          uint32_t jn[kSteps], drow[kSteps];  // list length (0: not in this window) and tile row of the step's entry
#pragma unroll
          for (int st = 0; st < kSteps; ++st) {
            const int j = h * 32 + st * 4 + g4;
            const int src = j < n ? j : 0;
            const uint32_t jm = lane_read(r_l.row, src);
            const uint32_t jo = lane_read(r_l.off, src);
            jn[st] = lane_read(r_l.cnt, src);
            const int d = lane_read(rank_l, src) - wb;
            if (!(j < n && d >= 0 && d < (int)R)) jn[st] = 0;
            drow[st] = (uint32_t)d * K;
            if ((uint32_t)sub16 < jn[st]) pr[st] = p_b[(uint64_t)jm * K + jo + (uint32_t)sub16];
          }
#pragma unroll
          for (int st = 0; st < kSteps; ++st) {
            if ((uint32_t)sub16 < jn[st]) atomicAdd(tile + drow[st] + PT::key(pr[st]), PT::val(pr[st]));
          }
        }
        // the rest of the lists of entries that win more than 16 features (the entries of short rows: up to K) is read
        // by the WHOLE wave, 2 x 64 pairs per round trip
        {
          const int dl = rank_l - wb;
          unsigned long long longs = __ballot(lane < n && r_l.cnt > (uint32_t)kLpe && dl >= 0 && dl < (int)R);
          while (longs != 0) {
            const int j = (int)__builtin_ctzll(longs);
            longs &= longs - 1;
            const uint32_t jm = (uint32_t)__builtin_amdgcn_readlane((int)r_l.row, j);
            const int jd = __builtin_amdgcn_readlane(rank_l, j) - wb;
            const uint32_t jo = (uint32_t)__builtin_amdgcn_readlane((int)r_l.off, j);
            const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)r_l.cnt, j);
            const PR *lp = p_b + (uint64_t)jm * K + jo;
            A *trow = tile + (uint32_t)jd * K;
            for (uint32_t i = (uint32_t)kLpe + (uint32_t)lane; i < cn; i += 2u * kWave) {
              const PR p2 = lp[i];
              const bool two = i + kWave < cn;
              const PR p3 = lp[two ? i + kWave : i];
              atomicAdd(trow + PT::key(p2), PT::val(p2));
              if (two) atomicAdd(trow + PT::key(p3), PT::val(p3));
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // flush the window's columns
        const int wlast = wb + (int)R - 1 < nd - 1 ? wb + (int)R - 1 : nd - 1;
        for (int rr = wb; rr <= wlast; ++rr) {
          const int hj = (int)__builtin_ctzll(hm);  // head lane of the rr-th distinct column
          hm &= hm - 1;
          const int64_t c = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)r_l.col, hj);
          const A *trow = tile + (uint32_t)(rr - wb) * K;
          const bool complete = rr < nd - 1 || end_hi <= p0 + n;  // no position of c behind this sub-chunk
          if (complete) {
            if (c != first_col || first_started_here) {
              for (uint32_t k = (uint32_t)lane; k < K; k += kWave) g_b[(uint64_t)c * K + k] = Traits<T>::from_acc(trow[k]);
            } else {  // the head of a column that earlier waves started: the fix-up finishes it
              for (uint32_t k = (uint32_t)lane; k < K; k += kWave)
                head_val[((uint64_t)b * cy.nwaves + (uint64_t)q) * K + k] = trow[k];
              hcol = c;
            }
            if (open_col == c) open_col = -1;
          } else {  // c == col_hi and it continues: keep its partial in row 0 for the next sub-chunk
            if (rr != wb) {
              for (uint32_t k = (uint32_t)lane; k < K; k += kWave) tile[k] = trow[k];
            }
            open_col = c;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    // what is still open continues behind this wave's positions
    int64_t tcol = -1;
    if (open_col >= 0) {
      for (uint32_t k = (uint32_t)lane; k < K; k += kWave)
        tail_val[((uint64_t)b * cy.nwaves + (uint64_t)q) * K + k] = tile[k];
      tcol = open_col;
    }
    if (b == 0 && lane == 0) {
      cy.head_col[q] = hcol;
      cy.tail_col[q] = tcol;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// 3. a column that ends in wave q's positions and started earlier: head(q) + tail(q-1) + tail(q-2) + ...
template <typename T>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void listfix_kernel(T *__restrict__ gmat, int64_t B, int64_t N,
                                                                       uint32_t K, PullCarry cy) {
  using A = typename Traits<T>::acc_t;
  const int lane = (int)(threadIdx.x & 63);
  const int wib = (int)(threadIdx.x >> 6);
  const int64_t q = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  if (q >= cy.nwaves) return;
  const int64_t c = cy.head_col[q];
  if (c < 0) return;
  int64_t run = 0;
  while (q - 1 - run >= 0 && cy.tail_col[q - 1 - run] == c) ++run;
  const A *head_val = reinterpret_cast<const A *>(cy.head_val), *tail_val = reinterpret_cast<const A *>(cy.tail_val);
  for (int64_t b = 0; b < B; ++b) {
    for (uint32_t k = (uint32_t)lane; k < K; k += kWave) {
      // position order: the earliest wave's tail first, this wave's head last
      A acc = A(0);
      int64_t i = run;
      constexpr int kFold = 8;  // a hub column is cut into hundreds of pieces: 8 independent loads per round trip
      for (; i >= kFold; i -= kFold) {
        A v[kFold];
#pragma unroll
        for (int f = 0; f < kFold; ++f) v[f] = tail_val[((uint64_t)b * cy.nwaves + (uint64_t)(q - i + f)) * K + k];
#pragma unroll
        for (int f = 0; f < kFold; ++f) acc += v[f];
      }
      for (; i >= 1; --i) acc += tail_val[((uint64_t)b * cy.nwaves + (uint64_t)(q - i)) * K + k];
      acc += head_val[((uint64_t)b * cy.nwaves + (uint64_t)q) * K + k];
      gmat[((uint64_t)b * N + (uint64_t)c) * K + k] = Traits<T>::from_acc(acc);
    }
  }
}

struct ListWs {
  ListRec *rec;
  void *pairs;
  PullCarry cy;
};

template <typename T>
size_t carve_lists(void *base, int64_t B, int64_t M, int64_t K, int64_t E, ListWs *ws) {
  char *p = reinterpret_cast<char *>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) -> void * {
    void *r = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return r;
  };
  using A = typename Traits<T>::acc_t;
  ListWs w;
  const int64_t nwaves = ceil_div(E > 0 ? E : 1, kSubChunks * kWave);
  w.rec = reinterpret_cast<ListRec *>(take(sizeof(ListRec) * (size_t)(B * E)));
  w.pairs = take(sizeof(typename PairOf<T>::type) * (size_t)(B * M * K));
  w.cy.nwaves = nwaves;
  w.cy.head_col = reinterpret_cast<int64_t *>(take(sizeof(int64_t) * (size_t)nwaves));
  w.cy.tail_col = reinterpret_cast<int64_t *>(take(sizeof(int64_t) * (size_t)nwaves));
  w.cy.head_val = take(sizeof(A) * (size_t)(B * nwaves * K));
  w.cy.tail_val = take(sizeof(A) * (size_t)(B * nwaves * K));
  if (ws) *ws = w;
  return off;
}

}  // namespace

bool minmax_bw_lists_supported(int dtype, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E) {
  (void)B;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16) return false;
  return K >= 1 && K <= kTileWords && K < 65536 && M < ((int64_t)1 << 32) && N < ((int64_t)1 << 32) &&
         E < ((int64_t)1 << 32);
}

size_t minmax_bw_lists_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E) {
  (void)N;
  switch (dtype) {
    case TSAMD_F32: return carve_lists<float>(nullptr, B, M, K, E, nullptr);
    case TSAMD_F64: return carve_lists<double>(nullptr, B, M, K, E, nullptr);
    case TSAMD_F16: return carve_lists<f16_t>(nullptr, B, M, K, E, nullptr);
    case TSAMD_BF16: return carve_lists<bf16_t>(nullptr, B, M, K, E, nullptr);
    default: return 0;
  }
}

template <typename T>
static int run_lists(const int64_t *row, const int64_t *col, const void *value, const void *grad_out,
                     const int64_t *arg_out, const int64_t *colptr, const int64_t *csr2csc, void *grad_mat, int64_t B,
                     int64_t M, int64_t N, int64_t K, int64_t E, void *workspace, hipStream_t stream) {
  ListWs ws;
  carve_lists<T>(workspace, B, M, K, E, &ws);
  // columns without entries are never visited by the pull: zero first (the visited ones are overwritten)
  TSAMD_HIP_TRY(hipMemsetAsync(grad_mat, 0, sizeof(T) * (size_t)(B * N * K), stream));
  const unsigned int blocks1 = (unsigned int)ceil_div(ceil_div(E, kWave), kWavesPerBlock);
  hipLaunchKernelGGL((winlist_kernel<T>), dim3(blocks1), dim3(kWavesPerBlock * kWave), 0, stream, row, col,
                     reinterpret_cast<const T *>(value), arg_out, reinterpret_cast<const T *>(grad_out), ws.rec,
                     reinterpret_cast<typename PairOf<T>::type *>(ws.pairs), B, M, (uint32_t)K, E);
  TSAMD_LAUNCH_CHECK();
  const unsigned int blocks2 = (unsigned int)ceil_div(ws.cy.nwaves, kWavesPerBlock);
  hipLaunchKernelGGL((listpull_kernel<T>), dim3(blocks2), dim3(kWavesPerBlock * kWave), 0, stream, colptr, csr2csc,
                     ws.rec, reinterpret_cast<const typename PairOf<T>::type *>(ws.pairs),
                     reinterpret_cast<T *>(grad_mat), B, M, N, (uint32_t)K, E, ws.cy);
  TSAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL((listfix_kernel<T>), dim3(blocks2), dim3(kWavesPerBlock * kWave), 0, stream,
                     reinterpret_cast<T *>(grad_mat), B, N, (uint32_t)K, ws.cy);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

int minmax_bw_lists(int dtype, const int64_t *row, const int64_t *col, const void *value, const void *grad_out,
                    const int64_t *arg_out, const int64_t *colptr, const int64_t *csr2csc, void *grad_mat, int64_t B,
                    int64_t M, int64_t N, int64_t K, int64_t E, void *workspace, hipStream_t stream) {
  switch (dtype) {
    case TSAMD_F32:
      return run_lists<float>(row, col, value, grad_out, arg_out, colptr, csr2csc, grad_mat, B, M, N, K, E, workspace, stream);
    case TSAMD_F64:
      return run_lists<double>(row, col, value, grad_out, arg_out, colptr, csr2csc, grad_mat, B, M, N, K, E, workspace, stream);
    case TSAMD_F16:
      return run_lists<f16_t>(row, col, value, grad_out, arg_out, colptr, csr2csc, grad_mat, B, M, N, K, E, workspace, stream);
    case TSAMD_BF16:
      return run_lists<bf16_t>(row, col, value, grad_out, arg_out, colptr, csr2csc, grad_mat, B, M, N, K, E, workspace, stream);
    default: return TSAMD_ERR_UNSUPPORTED;
  }
}

}  // namespace tsamd
#endif  // TSAMD_EXPERIMENTS
