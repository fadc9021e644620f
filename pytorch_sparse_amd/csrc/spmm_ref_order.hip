// SpMM in the REFERENCE'S ORDER OF OPERATIONS -- a verification mode, not a fast path.
//
// The product kernels (csrc/spmm.hip) cut rows between waves and add partial sums in a tree: their fp32 sums differ
// from the reference CPU kernel's in the last bits (inside 1e-5 * sum|terms|, tests/ and bench.py count it), because
// that kernel (csrc/cpu/spmm_cpu.cpp:61-87 with csrc/cpu/reducer.h:43-84) walks a row's entries one after the other,
// multiplies, rounds, adds, rounds.  This kernel does exactly that -- one thread per (batch, row, feature), entries in
// CSR order, the multiply and the add as two separately rounded operations in the element type (no fused multiply-add;
// f16 / bf16 round after each, as c10::Half arithmetic does), the mean divided by the count converted to the element
// type -- so that every reduction of every dtype is BIT-IDENTICAL to the reference on the same inputs.
// Selected per process with tsamd_spmm_reference_order(1) (include/tsamd.h); tests/test_spmm_gpu.py and the parity leg
// of bench.py compare it with the compiled reference bit for bit.
#include "common.h"
#include "spmm_internal.h"

#include <atomic>
#include <type_traits>

namespace tsamd {
namespace {

std::atomic<int> g_reference_order{0};

template <typename T, int REDUCE, typename ArgT>
__global__ __launch_bounds__(256) void spmm_reference_order_kernel(
    const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col, const T *__restrict__ value,
    const T *__restrict__ mat, T *__restrict__ out, ArgT *__restrict__ arg_out, int64_t B, int64_t M, int64_t N,
    int64_t K, int64_t E) {
#pragma clang fp contract(off)
  using A = typename Traits<T>::acc_t;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * M * K) return;
  const int64_t k = t % K, m = (t / K) % M, b = t / (K * M);
  const int64_t row_start = rowptr[m], row_end = rowptr[m + 1];
  A acc = REDUCE == TSAMD_MIN ? Traits<T>::max_init() : (REDUCE == TSAMD_MAX ? Traits<T>::lowest_init() : (A)0);
  int64_t arg = E;  // (the reference pre-sets arg_out to E and leaves it where nothing beats the initial value)
  const T *x = mat + (size_t)b * N * K + k;
  for (int64_t e = row_start; e < row_end; ++e) {
    A v = Traits<T>::to_acc(x[(size_t)col[e] * K]);
    if (value != nullptr) v = Traits<T>::round_acc(Traits<T>::to_acc(value[e]) * v);
    if (REDUCE == TSAMD_SUM || REDUCE == TSAMD_MEAN) {
      acc = Traits<T>::round_acc(acc + v);
    } else if ((REDUCE == TSAMD_MIN && v < acc) || (REDUCE == TSAMD_MAX && v > acc)) {
      acc = v;
      arg = e;
    }
  }
  const int64_t count = row_end - row_start;
  if (REDUCE == TSAMD_MEAN) {
    if constexpr (std::is_floating_point<A>::value) {
      const A d = Traits<T>::round_acc((A)(count > 0 ? count : 1));  // (scalar_t)count, reducer.h:74
      acc = acc / d;
    } else {
      acc = mean_of<T>(acc, count);
    }
  }
  if (REDUCE == TSAMD_MIN || REDUCE == TSAMD_MAX) {
    if (count > 0) {
      out[t] = Traits<T>::from_acc(acc);
      arg_out[t] = (ArgT)arg;
    } else {
      out[t] = Traits<T>::from_acc((A)0);
      arg_out[t] = (ArgT)E;
    }
  } else {
    out[t] = Traits<T>::from_acc(acc);
  }
}

template <typename T, typename ArgT>
int launch(int reduce, const int64_t *rowptr, const int64_t *col, const T *value, const T *mat, T *out, ArgT *arg_out,
           int64_t B, int64_t M, int64_t N, int64_t K, int64_t E, hipStream_t stream) {
  const int64_t total = B * M * K;
  const dim3 grid((unsigned int)ceil_div(total, 256)), block(256);
  switch (reduce) {
    case TSAMD_SUM:
      hipLaunchKernelGGL((spmm_reference_order_kernel<T, TSAMD_SUM, ArgT>), grid, block, 0, stream, rowptr, col, value, mat,
                         out, arg_out, B, M, N, K, E);
      break;
    case TSAMD_MEAN:
      hipLaunchKernelGGL((spmm_reference_order_kernel<T, TSAMD_MEAN, ArgT>), grid, block, 0, stream, rowptr, col, value, mat,
                         out, arg_out, B, M, N, K, E);
      break;
    case TSAMD_MIN:
      hipLaunchKernelGGL((spmm_reference_order_kernel<T, TSAMD_MIN, ArgT>), grid, block, 0, stream, rowptr, col, value, mat,
                         out, arg_out, B, M, N, K, E);
      break;
    default:
      hipLaunchKernelGGL((spmm_reference_order_kernel<T, TSAMD_MAX, ArgT>), grid, block, 0, stream, rowptr, col, value, mat,
                         out, arg_out, B, M, N, K, E);
  }
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

}  // namespace

bool spmm_reference_order_on() { return g_reference_order.load(std::memory_order_relaxed) != 0; }

int spmm_reference_order_run(int dtype, int reduce, const int64_t *rowptr, const int64_t *col, const void *value,
                             const void *mat, void *out, void *arg_out, bool arg32, int64_t B, int64_t M, int64_t N,
                             int64_t K, int64_t E, hipStream_t stream) {
  if (B * M * K >= ((int64_t)1 << 31) * 256) return TSAMD_ERR_UNSUPPORTED;
  return TSAMD_DISPATCH_DTYPE_ALL(dtype, [&]() -> int {
    const scalar_t *v = reinterpret_cast<const scalar_t *>(value), *x = reinterpret_cast<const scalar_t *>(mat);
    scalar_t *o = reinterpret_cast<scalar_t *>(out);
    if (arg32) return launch<scalar_t, int32_t>(reduce, rowptr, col, v, x, o, reinterpret_cast<int32_t *>(arg_out), B, M, N, K, E, stream);
    return launch<scalar_t, int64_t>(reduce, rowptr, col, v, x, o, reinterpret_cast<int64_t *>(arg_out), B, M, N, K, E, stream);
  });
}

}  // namespace tsamd

extern "C" int tsamd_spmm_reference_order(int set) {
  if (set == 0 || set == 1) tsamd::g_reference_order.store(set, std::memory_order_relaxed);
  return tsamd::g_reference_order.load(std::memory_order_relaxed);
}
