// Backward kernels of CSR SpMM for gfx950 (MI355X).
//
//  * tsamd_spmm_value_bw  -- gradient of SUM/MEAN SpMM w.r.t. the sparse values, an SDDMM
//    over the pattern.  Replaces spmm_value_bw_cuda / spmm_value_bw_cpu of the reference
//    (csrc/cuda/spmm_cuda.cu:157-237, csrc/cpu/spmm_cpu.cpp:103-152).
//  * tsamd_spmm_minmax_bw -- backward of MIN/MAX SpMM.  Replaces the ATen composition
//    (masked_fill / index_select / gather / scatter_add_) in SPMMMin/SPMMMax::backward
//    (csrc/spmm.cpp:204-242, 264-302) with one fused pass.
#include "common.h"

#include <type_traits>

namespace tsamd {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kUnroll = 2;

// ---------------------------------------------------------------------------
// value gradient: edge-parallel (perfectly balanced whatever the degrees).
// A wave owns 64 consecutive edges; (row, col) ids are read once, coalesced.
// G = 64/LPR groups of LPR lanes x VEC features each take one edge per step:
// two 16-byte gathers per lane (mat[col], grad[row]), a VEC-wide dot, then a
// butterfly over the LPR lanes.  Consecutive edges share `row`, so the grad
// row is served by L1/L2 after its first touch.  Results are moved to the lane
// that owns the edge and stored with one coalesced write per window.
// ---------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_value_bw_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ rowptr,
    const int64_t *__restrict__ col, const T *__restrict__ mat, const T *__restrict__ grad,
    T *__restrict__ out, int64_t B, int64_t M, int64_t N, uint32_t K, int64_t E, int lgG,
    bool mean) {
  using A = typename Traits<T>::acc_t;
  using P = Pack<T, VEC>;
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wib) * kWave;
  if (base >= E) return;
  const int64_t rem = E - base;
  const int n = rem < kWave ? (int)rem : kWave;

  // this lane's edge
  uint32_t c_l = 0, r_l = 0;
  int64_t deg_l = 1;
  if (lane < n) {
    const int64_t e = base + lane;
    c_l = (uint32_t)col[e];
    int64_t r;
    if (row != nullptr) {
      r = row[e];
    } else {  // last r with rowptr[r] <= e
      int64_t lo = 0, hi = M;
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid;
      }
      r = lo;
    }
    r_l = (uint32_t)r;
    if (mean) {
      deg_l = rowptr[r + 1] - rowptr[r];
      if (deg_l < 1) deg_l = 1;
    }
  }

  const int G = 1 << lgG;
  const int lpr = 64 >> lgG;
  const int g = lane >> (6 - lgG);
  const int kl = lane & (lpr - 1);
  const uint32_t slots = (K + VEC - 1) / VEC;
  const int nsteps = (n + G - 1) >> lgG;

  A mine = A(0);  // result of the edge owned by this lane
  for (int s = 0; s < nsteps; s += kUnroll) {
    A acc[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int idx = ((s + u) << lgG) + g;
      const int src = idx < n ? idx : n - 1;
      const uint32_t c = lane_read(c_l, src);
      const uint32_t r = lane_read(r_l, src);
      acc[u] = A(0);
      for (int64_t b = 0; b < B; ++b) {
        const T *mrow = mat + ((uint64_t)b * N + c) * K;
        const T *grow = grad + ((uint64_t)b * M + r) * K;
        for (uint32_t sl = kl; sl < slots; sl += lpr) {
          const P x = *reinterpret_cast<const P *>(mrow + (uint64_t)sl * VEC);
          const P y = *reinterpret_cast<const P *>(grow + (uint64_t)sl * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            acc[u] += Traits<T>::to_acc(x.v[j]) * Traits<T>::to_acc(y.v[j]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      A v = acc[u];
      for (int off = lpr >> 1; off > 0; off >>= 1) v += lane_xor(v, off);
      // slot (s+u)*G + g' was computed by group g'; its owner lane fetches it
      const int owner_step = lane >> lgG;          // step in which this lane's edge is handled
      const int owner_group = lane & (G - 1);
      const A got = lane_read(v, owner_group << (6 - lgG));
      if (owner_step == s + u) mine = got;
    }
  }
  if (lane < n) {
    if (mean) mine = mine / (A)deg_l;
    out[base + lane] = Traits<T>::from_acc(mine);
  }
}

// ---------------------------------------------------------------------------
// min/max backward: one thread per output element (b, m, k).
// ---------------------------------------------------------------------------
template <typename ACC>
__device__ inline ACC wave_sum(ACC v) {
  for (int off = 32; off > 0; off >>= 1) v += lane_xor(v, off);
  return v;
}

// One thread per output element (b, m, k).  grad_mat targets are distinct within a wave
// (same row, consecutive k), so they go out as plain atomics.  grad_value targets repeat a lot
// (the same neighbour usually wins many features of a row): equal targets are summed inside
// the wave first, one atomic per distinct edge, instead of up to 64 same-address atomics.
template <typename T, typename ACC>
__global__ void spmm_minmax_bw_kernel(const int64_t *__restrict__ col, const T *__restrict__ value,
                                      const T *__restrict__ mat, const T *__restrict__ grad_out,
                                      const int64_t *__restrict__ arg_out, ACC *__restrict__ gval,
                                      ACC *__restrict__ gmat, int64_t M, int64_t N, int64_t K,
                                      int64_t E, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  int64_t a = E;
  if (i < total) a = arg_out[i];
  const bool valid = a != E;  // empty row / no winner: masked out (spmm.cpp:210)
  ACC contrib = ACC(0);
  if (valid) {
    const int64_t k = i % K;
    const int64_t b = i / (M * K);
    const int64_t c = col[a];
    const ACC g = (ACC)Traits<T>::to_acc(grad_out[i]);
    const uint64_t xoff = ((uint64_t)b * N + c) * K + k;
    if (gval != nullptr) contrib = (ACC)Traits<T>::to_acc(mat[xoff]) * g;
    if (gmat != nullptr) {
      const ACC v = value != nullptr ? (ACC)Traits<T>::to_acc(value[a]) : ACC(1);
      atomicAdd(&gmat[xoff], v * g);
    }
  }
  if (gval == nullptr) return;
  unsigned long long todo = __ballot(valid);
  while (todo) {  // wave-uniform loop over the distinct targets
    const int leader = __ffsll((long long)todo) - 1;
    const int64_t a0 = lane_read(a, leader);
    const bool mine = valid && a == a0;
    const unsigned long long same = __ballot(mine);
    ACC s = contrib;
    if (__popcll(same) > 1) s = wave_sum<ACC>(mine ? contrib : ACC(0));
    if (lane == leader) atomicAdd(&gval[a0], s);
    todo &= ~same;
  }
}

template <typename T>
__global__ void narrow_from_f32_kernel(const float *__restrict__ src, T *__restrict__ dst,
                                       int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = Traits<T>::from_acc(src[i]);
}

int ilog2_ceil(uint32_t x) {
  int l = 0;
  while ((1u << l) < x) ++l;
  return l;
}

template <typename T, int VEC>
int launch_value_bw(const int64_t *row, const int64_t *rowptr, const int64_t *col, const T *mat,
                    const T *grad, T *out, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                    bool mean, hipStream_t stream) {
  const uint32_t slots = (uint32_t)((K + VEC - 1) / VEC);
  const uint32_t lpr = slots >= 64 ? 64u : (1u << ilog2_ceil(slots));
  const int lgG = 6 - ilog2_ceil(lpr);
  const unsigned int blocks = (unsigned int)ceil_div(ceil_div(E, kWave), kWavesPerBlock);
  hipLaunchKernelGGL((spmm_value_bw_kernel<T, VEC>), dim3(blocks), dim3(kWavesPerBlock * kWave), 0,
                     stream, row, rowptr, col, mat, grad, out, B, M, N, (uint32_t)K, E, lgG, mean);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

template <typename T>
int dispatch_value_bw(bool vec_ok, const int64_t *row, const int64_t *rowptr, const int64_t *col,
                      const void *mat, const void *grad, void *out, int64_t B, int64_t M,
                      int64_t N, int64_t K, int64_t E, bool mean, hipStream_t stream) {
  constexpr int kVec = 16 / (int)sizeof(T);
  const T *x = reinterpret_cast<const T *>(mat);
  const T *g = reinterpret_cast<const T *>(grad);
  T *o = reinterpret_cast<T *>(out);
  if (vec_ok) return launch_value_bw<T, kVec>(row, rowptr, col, x, g, o, B, M, N, K, E, mean, stream);
  return launch_value_bw<T, 1>(row, rowptr, col, x, g, o, B, M, N, K, E, mean, stream);
}

template <typename T, typename ACC>
int launch_minmax_bw(const int64_t *col, const void *value, const void *mat, const void *grad_out,
                     const int64_t *arg_out, ACC *gval, ACC *gmat, int64_t B, int64_t M, int64_t N,
                     int64_t K, int64_t E, hipStream_t stream) {
  const int64_t total = B * M * K;
  if (total == 0) return TSAMD_OK;
  hipLaunchKernelGGL((spmm_minmax_bw_kernel<T, ACC>), dim3((unsigned int)ceil_div(total, 256)),
                     dim3(256), 0, stream, col, reinterpret_cast<const T *>(value),
                     reinterpret_cast<const T *>(mat), reinterpret_cast<const T *>(grad_out),
                     arg_out, gval, gmat, M, N, K, E, total);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

template <typename T>
int narrow_out(const float *src, void *dst, int64_t n, hipStream_t stream) {
  if (n == 0) return TSAMD_OK;
  hipLaunchKernelGGL((narrow_from_f32_kernel<T>), dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0,
                     stream, src, reinterpret_cast<T *>(dst), n);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" int tsamd_spmm_value_bw(int dtype, int reduce, const int64_t *row,
                                   const int64_t *rowptr, const int64_t *col, const void *mat,
                                   const void *grad, void *out, int64_t B, int64_t M, int64_t N,
                                   int64_t K, int64_t E, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (reduce != TSAMD_SUM && reduce != TSAMD_MEAN) return TSAMD_ERR_UNSUPPORTED;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16)
    return TSAMD_ERR_UNSUPPORTED;
  if (N >= (int64_t)1 << 32 || M >= (int64_t)1 << 32 || K >= (int64_t)1 << 31)
    return TSAMD_ERR_UNSUPPORTED;
  if (E == 0) return TSAMD_OK;
  if (!rowptr || !col || !out || (B * K > 0 && (!mat || !grad))) return TSAMD_ERR_INVALID;
  const size_t es = dtype_size(dtype);
  const bool vec_ok = K > 0 && (K * es) % 16 == 0 && ((uintptr_t)mat % 16 == 0) &&
                      ((uintptr_t)grad % 16 == 0);
  const bool mean = reduce == TSAMD_MEAN;
  return TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    if constexpr (std::is_integral<scalar_t>::value) {
      return (int)TSAMD_ERR_UNSUPPORTED;
    } else {
      return dispatch_value_bw<scalar_t>(vec_ok, row, rowptr, col, mat, grad, out, B, M, N, K, E,
                                         mean, stream);
    }
  });
}

extern "C" size_t tsamd_spmm_minmax_bw_workspace_bytes(int dtype, int64_t B, int64_t N, int64_t K,
                                                       int64_t E) {
  if (dtype == TSAMD_F16 || dtype == TSAMD_BF16)
    return align_up(sizeof(float) * (size_t)E, 256) + align_up(sizeof(float) * (size_t)(B * N * K), 256);
  return 0;
}

extern "C" int tsamd_spmm_minmax_bw(int dtype, const int64_t *col, const void *value,
                                    const void *mat, const void *grad_out, const int64_t *arg_out,
                                    void *grad_value, void *grad_mat, int64_t B, int64_t M,
                                    int64_t N, int64_t K, int64_t E, void *workspace,
                                    size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16)
    return TSAMD_ERR_UNSUPPORTED;
  const int64_t total = B * M * K;
  if (total > 0 && (!col || !mat || !grad_out || !arg_out)) return TSAMD_ERR_INVALID;
  const size_t es = dtype_size(dtype);
  const size_t nmat = (size_t)(B * N * K);
  if (dtype == TSAMD_F32 || dtype == TSAMD_F64) {
    if (grad_value) TSAMD_HIP_TRY(hipMemsetAsync(grad_value, 0, es * (size_t)E, stream));
    if (grad_mat) TSAMD_HIP_TRY(hipMemsetAsync(grad_mat, 0, es * nmat, stream));
    if (dtype == TSAMD_F32)
      return launch_minmax_bw<float, float>(col, value, mat, grad_out, arg_out,
                                            reinterpret_cast<float *>(grad_value),
                                            reinterpret_cast<float *>(grad_mat), B, M, N, K, E, stream);
    return launch_minmax_bw<double, double>(col, value, mat, grad_out, arg_out,
                                            reinterpret_cast<double *>(grad_value),
                                            reinterpret_cast<double *>(grad_mat), B, M, N, K, E, stream);
  }
  // narrow types: accumulate in an fp32 workspace, round once
  const size_t need = tsamd_spmm_minmax_bw_workspace_bytes(dtype, B, N, K, E);
  if (!workspace || workspace_bytes < need) return TSAMD_ERR_WORKSPACE;
  float *wv = reinterpret_cast<float *>(workspace);
  float *wm = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) +
                                        align_up(sizeof(float) * (size_t)E, 256));
  TSAMD_HIP_TRY(hipMemsetAsync(workspace, 0, need, stream));
  int st;
  if (dtype == TSAMD_F16) {
    st = launch_minmax_bw<f16_t, float>(col, value, mat, grad_out, arg_out, grad_value ? wv : nullptr,
                                        grad_mat ? wm : nullptr, B, M, N, K, E, stream);
    if (st == TSAMD_OK && grad_value) st = narrow_out<f16_t>(wv, grad_value, E, stream);
    if (st == TSAMD_OK && grad_mat) st = narrow_out<f16_t>(wm, grad_mat, (int64_t)nmat, stream);
  } else {
    st = launch_minmax_bw<bf16_t, float>(col, value, mat, grad_out, arg_out, grad_value ? wv : nullptr,
                                         grad_mat ? wm : nullptr, B, M, N, K, E, stream);
    if (st == TSAMD_OK && grad_value) st = narrow_out<bf16_t>(wv, grad_value, E, stream);
    if (st == TSAMD_OK && grad_mat) st = narrow_out<bf16_t>(wm, grad_mat, (int64_t)nmat, stream);
  }
  return st;
}
