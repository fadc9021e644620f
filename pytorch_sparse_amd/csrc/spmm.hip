// CSR SpMM forward for gfx950 (MI355X), wave64 row-split.
//
// Replaces spmm_cuda / spmm_cpu of the reference (csrc/cuda/spmm_cuda.cu:92-155,
// csrc/cpu/spmm_cpu.cpp:8-101).  The arithmetic contract (init values, strict
// compares, first-occurrence ties, empty-row handling, mean divisor) follows
// csrc/cpu/reducer.h:43-84.
//
// Mapping ("SpMM kernel"):
//   * one wavefront owns one output row (b, m);
//   * the 64 lanes are split into G = 64 / LPR groups of LPR lanes, each lane
//     holding VEC consecutive features (16 bytes when the row pitch allows it),
//     so one vector-memory instruction gathers G different rows of `mat`, each
//     as one contiguous LPR*16-byte read (F=128 fp32: 2 rows x 512 B);
//   * a row's (col, value) pairs are read once, 64 per coalesced load, and
//     handed to the groups with ds_bpermute (no LDS allocation, no re-reads
//     per 32-column tile as in the reference kernel);
//   * U gathers are issued back to back before the first use (U*G rows in
//     flight per wave);
//   * groups are combined with a bpermute butterfly; MIN/MAX carry
//     (value, edge id) and break ties towards the smaller edge id;
//   * rows longer than kLongRow edges are not processed by their wave: they are
//     appended to a work list, cut into kChunk-edge pieces that are spread over
//     the whole chip by a second kernel, and merged in edge order by a third
//     (deterministic, no atomics on the data path).
#include "common.h"

namespace tsamd {
namespace {

constexpr int RED_ADD = 0;  // sum and mean
constexpr int RED_MIN = 1;
constexpr int RED_MAX = 2;

constexpr int kUnroll = 4;          // gathers in flight per group
constexpr int kWavesPerBlock = 4;   // 256-thread workgroups
constexpr int kLongRow = 512;       // rows above this go to the long-row path
constexpr int kChunk = 512;         // edges per long-row work item
constexpr int64_t kNoArg = 0x7fffffffffffffffLL;

struct LongRec {  // one long row
  int64_t vrow;   // b * M + m
  int64_t first;  // first work item
  int64_t nchunk;
};
struct LongItem {  // one kChunk-edge piece of a long row
  int64_t vrow;
  int64_t chunk;
};

struct Workspace {
  unsigned int *counters;  // [0] = #items, [1] = #long rows
  LongRec *recs;
  LongItem *items;
  void *part_val;     // [items][K] acc_t
  int64_t *part_arg;  // [items][K] (min/max only)
  int64_t max_recs, max_items;
};

// Accumulate edges [eb, ee) of one row into val/arg.  All 64 lanes stay
// active; lanes whose feature slot is out of range load from slot 0 and are
// masked at the store.
template <typename T, int VEC, int RED>
__device__ __forceinline__ void accumulate_range(
    int64_t eb, int64_t ee, const int64_t *__restrict__ col,
    const T *__restrict__ value, const T *__restrict__ matk, uint32_t K, int lane,
    int lgG, int g, typename Traits<T>::acc_t (&val)[VEC], int64_t (&arg)[VEC]) {
  using A = typename Traits<T>::acc_t;
  using P = Pack<T, VEC>;
  for (int64_t base = eb; base < ee; base += kWave) {
    const int64_t rem = ee - base;
    const int n = rem < kWave ? (int)rem : kWave;
    uint32_t c_l = 0;
    A w_l = A(1);
    if (lane < n) {
      c_l = (uint32_t)col[base + lane];
      if (value != nullptr) w_l = Traits<T>::to_acc(value[base + lane]);
    }
    const int nsteps = (n + (1 << lgG) - 1) >> lgG;
    for (int s = 0; s < nsteps; s += kUnroll) {
      P x[kUnroll];
      A w[kUnroll];
      int idx[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        idx[u] = ((s + u) << lgG) + g;
        const int src = idx[u] < n ? idx[u] : n - 1;
        const uint32_t c = lane_read(c_l, src);
        w[u] = lane_read(w_l, src);
        x[u] = *reinterpret_cast<const P *>(matk + (uint64_t)c * K);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const bool ok = idx[u] < n;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const A xv = Traits<T>::to_acc(x[u].v[j]);
          if constexpr (RED == RED_ADD) {
            const A p = w[u] * xv;
            val[j] += ok ? p : A(0);
          } else {
            const A p = Traits<T>::round_acc(w[u] * xv);
            const bool better = RED == RED_MIN ? (p < val[j]) : (p > val[j]);
            if (ok && better) {
              val[j] = p;
              arg[j] = base + idx[u];
            }
          }
        }
      }
    }
  }
}

// Butterfly over the G groups; afterwards every lane holds the row result.
template <typename A, int VEC, int RED>
__device__ __forceinline__ void reduce_groups(int lgG, A (&val)[VEC], int64_t (&arg)[VEC]) {
  for (int off = 32; off >= (64 >> lgG); off >>= 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const A o = lane_xor(val[j], off);
      if constexpr (RED == RED_ADD) {
        val[j] += o;
      } else {
        const int64_t oa = lane_xor(arg[j], off);
        const bool better = RED == RED_MIN ? (o < val[j]) : (o > val[j]);
        if (better || (o == val[j] && oa < arg[j])) {
          val[j] = o;
          arg[j] = oa;
        }
      }
    }
  }
}

template <typename T, int VEC, int RED>
__device__ __forceinline__ void init_acc(typename Traits<T>::acc_t (&val)[VEC],
                                         int64_t (&arg)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if constexpr (RED == RED_ADD) val[j] = 0;
    else if constexpr (RED == RED_MIN) val[j] = Traits<T>::max_init();
    else val[j] = Traits<T>::lowest_init();
    arg[j] = kNoArg;
  }
}

// Final write of one row (reducer.h:69-83).
template <typename T, int VEC, int RED>
__device__ __forceinline__ void write_row(T *__restrict__ outk, int64_t *__restrict__ argk,
                                          typename Traits<T>::acc_t (&val)[VEC],
                                          int64_t (&arg)[VEC], int64_t deg, bool mean,
                                          int64_t E) {
  using A = typename Traits<T>::acc_t;
  Pack<T, VEC> o;
  if constexpr (RED == RED_ADD) {
    if (mean) {
      const A d = (A)(deg > 0 ? deg : 1);
#pragma unroll
      for (int j = 0; j < VEC; ++j) val[j] = val[j] / d;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = Traits<T>::from_acc(val[j]);
    *reinterpret_cast<Pack<T, VEC> *>(outk) = o;
  } else {
    Pack<int64_t, VEC> a;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (deg > 0) {
        o.v[j] = Traits<T>::from_acc(val[j]);
        a.v[j] = arg[j] == kNoArg ? E : arg[j];
      } else {
        o.v[j] = Traits<T>::from_acc(A(0));
        a.v[j] = E;
      }
    }
    *reinterpret_cast<Pack<T, VEC> *>(outk) = o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) argk[j] = a.v[j];
  }
}

// --------------------------------------------------------------------------
// main kernel: one wave per (row, feature tile)
// --------------------------------------------------------------------------
template <typename T, int VEC, int RED>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_rows_kernel(
    const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
    const T *__restrict__ value, const T *__restrict__ mat, T *__restrict__ out,
    int64_t *__restrict__ arg_out, int64_t BM, int64_t M, int64_t N, uint32_t K,
    int64_t E, int lgG, bool mean, Workspace ws) {
  using A = typename Traits<T>::acc_t;
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t vrow = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  if (vrow >= BM) return;
  const int64_t b = vrow / M;
  const int64_t m = vrow - b * M;
  const int64_t e0 = rowptr[m];
  const int64_t e1 = rowptr[m + 1];
  const int64_t deg = e1 - e0;

  if (deg > kLongRow) {
    if (blockIdx.y == 0) {
      const int64_t nch = (deg + kChunk - 1) / kChunk;
      unsigned int first = 0, r = 0;
      if (lane == 0) {
        first = atomicAdd(&ws.counters[0], (unsigned int)nch);
        r = atomicAdd(&ws.counters[1], 1u);
      }
      first = __builtin_amdgcn_readfirstlane(first);
      r = __builtin_amdgcn_readfirstlane(r);
      if (lane == 0) ws.recs[r] = LongRec{vrow, (int64_t)first, nch};
      for (int64_t c = lane; c < nch; c += kWave) ws.items[first + c] = LongItem{vrow, c};
    }
    return;
  }

  const int lpr = 64 >> lgG;
  const int g = lane >> (6 - lgG);
  const int kl = lane & (lpr - 1);
  const uint32_t k0 = (blockIdx.y * 64u + (uint32_t)kl) * VEC;
  const bool kok = k0 < K;
  const T *matk = mat + (uint64_t)b * N * K + (kok ? k0 : 0u);

  A val[VEC];
  int64_t arg[VEC];
  init_acc<T, VEC, RED>(val, arg);
  accumulate_range<T, VEC, RED>(e0, e1, col, value, matk, K, lane, lgG, g, val, arg);
  reduce_groups<A, VEC, RED>(lgG, val, arg);
  if (g == 0 && kok) {
    const uint64_t o = (uint64_t)vrow * K + k0;
    write_row<T, VEC, RED>(out + o, arg_out ? arg_out + o : nullptr, val, arg, deg, mean, E);
  }
}

// --------------------------------------------------------------------------
// long rows, pass 1: one wave per kChunk-edge work item, all feature tiles
// --------------------------------------------------------------------------
template <typename T, int VEC, int RED>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_long_chunks_kernel(
    const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
    const T *__restrict__ value, const T *__restrict__ mat, int64_t M, int64_t N,
    uint32_t K, int lgG, Workspace ws) {
  using A = typename Traits<T>::acc_t;
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t nitems = ws.counters[0];
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int lpr = 64 >> lgG;
  const int g = lane >> (6 - lgG);
  const int kl = lane & (lpr - 1);
  const uint32_t ktiles = (K + 64u * VEC - 1) / (64u * VEC);
  A *part_val = reinterpret_cast<A *>(ws.part_val);
  for (int64_t it = (int64_t)blockIdx.x * kWavesPerBlock + wib; it < nitems; it += nwaves) {
    const LongItem item = ws.items[it];
    const int64_t b = item.vrow / M;
    const int64_t m = item.vrow - b * M;
    const int64_t e0 = rowptr[m] + item.chunk * kChunk;
    const int64_t e1r = rowptr[m + 1];
    const int64_t e1 = e0 + kChunk < e1r ? e0 + kChunk : e1r;
    for (uint32_t kt = 0; kt < ktiles; ++kt) {
      const uint32_t k0 = (kt * 64u + (uint32_t)kl) * VEC;
      const bool kok = k0 < K;
      const T *matk = mat + (uint64_t)b * N * K + (kok ? k0 : 0u);
      A val[VEC];
      int64_t arg[VEC];
      init_acc<T, VEC, RED>(val, arg);
      accumulate_range<T, VEC, RED>(e0, e1, col, value, matk, K, lane, lgG, g, val, arg);
      reduce_groups<A, VEC, RED>(lgG, val, arg);
      if (g == 0 && kok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          part_val[(uint64_t)it * K + k0 + j] = val[j];
          if constexpr (RED != RED_ADD) ws.part_arg[(uint64_t)it * K + k0 + j] = arg[j];
        }
      }
    }
  }
}

// --------------------------------------------------------------------------
// long rows, pass 2: one wave per long row merges its pieces in edge order
// --------------------------------------------------------------------------
template <typename T, int RED>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_long_merge_kernel(
    const int64_t *__restrict__ rowptr, T *__restrict__ out, int64_t *__restrict__ arg_out,
    int64_t M, uint32_t K, int64_t E, bool mean, Workspace ws) {
  using A = typename Traits<T>::acc_t;
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t nrecs = ws.counters[1];
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const A *part_val = reinterpret_cast<const A *>(ws.part_val);
  for (int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + wib; r < nrecs; r += nwaves) {
    const LongRec rec = ws.recs[r];
    const int64_t m = rec.vrow % M;
    const int64_t deg = rowptr[m + 1] - rowptr[m];
    for (uint32_t k = lane; k < K; k += kWave) {
      A val[1];
      int64_t arg[1];
      init_acc<T, 1, RED>(val, arg);
      for (int64_t c = 0; c < rec.nchunk; ++c) {
        const uint64_t p = (uint64_t)(rec.first + c) * K + k;
        const A o = part_val[p];
        if constexpr (RED == RED_ADD) {
          val[0] += o;
        } else {
          const int64_t oa = ws.part_arg[p];
          const bool better = RED == RED_MIN ? (o < val[0]) : (o > val[0]);
          // pieces arrive in edge order: a strict compare keeps the first winner
          if (better) {
            val[0] = o;
            arg[0] = oa;
          }
        }
      }
      const uint64_t o = (uint64_t)rec.vrow * K + k;
      write_row<T, 1, RED>(out + o, arg_out ? arg_out + o : nullptr, val, arg, deg, mean, E);
    }
  }
}

int ilog2_ceil(uint32_t x) {
  int l = 0;
  while ((1u << l) < x) ++l;
  return l;
}

Workspace carve_workspace(void *base, int dtype, int reduce, int64_t B, int64_t K, int64_t E) {
  Workspace ws;
  const int64_t BE = B * E;
  ws.max_recs = BE / (kLongRow + 1) + 1;
  ws.max_items = BE / kChunk + ws.max_recs + 1;
  char *p = reinterpret_cast<char *>(base);
  size_t off = 0;
  ws.counters = reinterpret_cast<unsigned int *>(p + off);
  off += 256;
  ws.recs = reinterpret_cast<LongRec *>(p + off);
  off += align_up(sizeof(LongRec) * ws.max_recs, 256);
  ws.items = reinterpret_cast<LongItem *>(p + off);
  off += align_up(sizeof(LongItem) * ws.max_items, 256);
  ws.part_val = p + off;
  off += align_up(acc_size(dtype) * (size_t)ws.max_items * K, 256);
  ws.part_arg = reinterpret_cast<int64_t *>(p + off);
  if (reduce == TSAMD_MIN || reduce == TSAMD_MAX)
    off += align_up(sizeof(int64_t) * (size_t)ws.max_items * K, 256);
  (void)off;
  return ws;
}

size_t workspace_bytes(int dtype, int reduce, int64_t B, int64_t K, int64_t E) {
  const int64_t BE = B * E;
  const int64_t max_recs = BE / (kLongRow + 1) + 1;
  const int64_t max_items = BE / kChunk + max_recs + 1;
  size_t off = 256;
  off += align_up(sizeof(LongRec) * max_recs, 256);
  off += align_up(sizeof(LongItem) * max_items, 256);
  off += align_up(acc_size(dtype) * (size_t)max_items * K, 256);
  if (reduce == TSAMD_MIN || reduce == TSAMD_MAX)
    off += align_up(sizeof(int64_t) * (size_t)max_items * K, 256);
  return off;
}

template <typename T, int VEC, int RED>
int launch_spmm(const int64_t *rowptr, const int64_t *col, const T *value, const T *mat,
                T *out, int64_t *arg_out, int64_t B, int64_t M, int64_t N, int64_t K,
                int64_t E, bool mean, Workspace ws, hipStream_t stream) {
  const int64_t BM = B * M;
  const uint32_t slots = (uint32_t)((K + VEC - 1) / VEC);  // feature packets per row
  const uint32_t lpr = slots >= 64 ? 64u : (1u << ilog2_ceil(slots));
  const int lgG = 6 - ilog2_ceil(lpr);
  const uint32_t ktiles = (slots + 63) / 64;

  TSAMD_HIP_TRY(hipMemsetAsync(ws.counters, 0, 2 * sizeof(unsigned int), stream));
  {
    dim3 grid((unsigned int)ceil_div(BM, kWavesPerBlock), ktiles, 1);
    hipLaunchKernelGGL((spmm_rows_kernel<T, VEC, RED>), grid, dim3(kWavesPerBlock * kWave), 0,
                       stream, rowptr, col, value, mat, out, arg_out, BM, M, N, (uint32_t)K, E,
                       lgG, mean, ws);
    TSAMD_LAUNCH_CHECK();
  }
  if (E > kLongRow) {  // a row can only be long if the matrix has that many entries
    const unsigned int nblk = 2048;
    hipLaunchKernelGGL((spmm_long_chunks_kernel<T, VEC, RED>), dim3(nblk),
                       dim3(kWavesPerBlock * kWave), 0, stream, rowptr, col, value, mat, M, N,
                       (uint32_t)K, lgG, ws);
    TSAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL((spmm_long_merge_kernel<T, RED>), dim3(256), dim3(kWavesPerBlock * kWave),
                       0, stream, rowptr, out, arg_out, M, (uint32_t)K, E, mean, ws);
    TSAMD_LAUNCH_CHECK();
  }
  return TSAMD_OK;
}

template <typename T>
int dispatch_spmm(int reduce, bool vec_ok, const int64_t *rowptr, const int64_t *col,
                  const void *value, const void *mat, void *out, int64_t *arg_out, int64_t B,
                  int64_t M, int64_t N, int64_t K, int64_t E, Workspace ws, hipStream_t stream) {
  constexpr int kVec = 16 / (int)sizeof(T);
  const T *v = reinterpret_cast<const T *>(value);
  const T *x = reinterpret_cast<const T *>(mat);
  T *o = reinterpret_cast<T *>(out);
  const bool mean = reduce == TSAMD_MEAN;
#define TSAMD_SPMM_GO(VEC, RED) \
  return launch_spmm<T, VEC, RED>(rowptr, col, v, x, o, arg_out, B, M, N, K, E, mean, ws, stream)
  if (vec_ok) {
    if (reduce == TSAMD_MIN) TSAMD_SPMM_GO(kVec, RED_MIN);
    if (reduce == TSAMD_MAX) TSAMD_SPMM_GO(kVec, RED_MAX);
    TSAMD_SPMM_GO(kVec, RED_ADD);
  } else {
    if (reduce == TSAMD_MIN) TSAMD_SPMM_GO(1, RED_MIN);
    if (reduce == TSAMD_MAX) TSAMD_SPMM_GO(1, RED_MAX);
    TSAMD_SPMM_GO(1, RED_ADD);
  }
#undef TSAMD_SPMM_GO
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" size_t tsamd_spmm_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M,
                                             int64_t K, int64_t E) {
  (void)M;
  if (dtype_size(dtype) == 0 || B < 0 || K < 0 || E < 0) return 0;
  return workspace_bytes(dtype, reduce, B, K, E);
}

extern "C" int tsamd_spmm(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                          const void *value, const void *mat, void *out, int64_t *arg_out,
                          int64_t B, int64_t M, int64_t N, int64_t K, int64_t E, void *workspace,
                          size_t workspace_bytes_given, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (reduce < TSAMD_SUM || reduce > TSAMD_MAX) return TSAMD_ERR_UNSUPPORTED;
  if (dtype_size(dtype) == 0) return TSAMD_ERR_UNSUPPORTED;
  if (N >= (int64_t)1 << 32 || K >= (int64_t)1 << 31) return TSAMD_ERR_UNSUPPORTED;
  const bool minmax = reduce == TSAMD_MIN || reduce == TSAMD_MAX;
  if (B * M * K == 0) return TSAMD_OK;  // nothing to write
  if (!rowptr || !out || (E > 0 && (!col || !mat)) || (minmax && !arg_out))
    return TSAMD_ERR_INVALID;
  const size_t need = workspace_bytes(dtype, reduce, B, K, E);
  if (!workspace || workspace_bytes_given < need) return TSAMD_ERR_WORKSPACE;
  Workspace ws = carve_workspace(workspace, dtype, reduce, B, K, E);
  const size_t es = dtype_size(dtype);
  const bool vec_ok = (K * es) % 16 == 0 && ((uintptr_t)mat % 16 == 0) &&
                      ((uintptr_t)out % 16 == 0);

  return TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    return dispatch_spmm<scalar_t>(reduce, vec_ok, rowptr, col, value, mat, out, arg_out, B, M, N,
                                   K, E, ws, stream);
  });
}
