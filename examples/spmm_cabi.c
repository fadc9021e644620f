/* Calling the hot path through the C-ABI alone: no torch, no C++ -- plain C, the HIP runtime for
 * device memory, and include/tsamd.h.  This is what a binding of the reference's dispatcher
 * (csrc/spmm.cpp:22-35) or any foreign-function interface (ctypes, cgo, JNI ...) does.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/spmm_cabi.c -Iinclude -I/opt/rocm/include \
 *       -Lpytorch_sparse_amd/lib -ltsamd -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/pytorch_sparse_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/spmm_cabi && /tmp/spmm_cabi
 *
 * Multiplies the README matrix of the reference,
 *     [[1,0,2,0],[0,0,4,3],[0,5,0,0]] (stored with a duplicate-free COO) times X = [[1,4],[2,5],[4,3],[3,6]]
 * and checks sum / max against the values worked out by hand. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsamd.h"

#define HIP_OK(x)                                                       \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
      return 2;                                                         \
    }                                                                   \
  } while (0)

static void *to_device(const void *src, size_t bytes) {
  void *d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return NULL;
  if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  enum { M = 3, N = 4, K = 2, E = 5 };
  const int64_t rowptr[M + 1] = {0, 2, 4, 5};
  const int64_t col[E] = {0, 2, 2, 3, 1};
  const float value[E] = {1, 2, 4, 3, 5};
  const float x[N * K] = {1, 4, 2, 5, 4, 3, 3, 6};
  const float want_sum[M * K] = {9, 10, 25, 30, 10, 25};
  const float want_max[M * K] = {8, 6, 16, 18, 10, 25};
  const int64_t want_arg[M * K] = {1, 1, 2, 3, 4, 4};

  void *d_rowptr = to_device(rowptr, sizeof rowptr), *d_col = to_device(col, sizeof col);
  void *d_value = to_device(value, sizeof value), *d_x = to_device(x, sizeof x);
  void *d_out = NULL, *d_arg = NULL, *d_ws = NULL;
  if (!d_rowptr || !d_col || !d_value || !d_x) {
    fprintf(stderr, "no HIP device / out of memory\n");
    return 2;
  }
  HIP_OK(hipMalloc(&d_out, sizeof want_sum));
  HIP_OK(hipMalloc(&d_arg, sizeof want_arg));

  float out[M * K];
  int64_t arg[M * K];
  int fails = 0;
  for (int reduce = TSAMD_SUM; reduce <= TSAMD_MAX; reduce += TSAMD_MAX - TSAMD_SUM) {
    const size_t ws_bytes = tsamd_spmm_workspace_bytes(TSAMD_F32, reduce, 1, M, N, K, E);
    HIP_OK(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 1));
    const int st = tsamd_spmm(TSAMD_F32, reduce, (const int64_t *)d_rowptr, (const int64_t *)d_col, d_value, d_x,
                              d_out, reduce == TSAMD_MAX ? (int64_t *)d_arg : NULL, 1, M, N, K, E, d_ws,
                              ws_bytes, /*stream=*/NULL);
    if (st != TSAMD_OK) {
      fprintf(stderr, "tsamd_spmm: %s (hip error %d)\n", tsamd_status_string(st), tsamd_last_hip_error());
      return 1;
    }
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
    const float *want = reduce == TSAMD_SUM ? want_sum : want_max;
    for (int i = 0; i < M * K; ++i) fails += out[i] != want[i];
    if (reduce == TSAMD_MAX) {
      HIP_OK(hipMemcpy(arg, d_arg, sizeof arg, hipMemcpyDeviceToHost));
      for (int i = 0; i < M * K; ++i) fails += arg[i] != want_arg[i];
    }
    printf("%s: [%g %g | %g %g | %g %g]\n", reduce == TSAMD_SUM ? "sum" : "max", out[0], out[1], out[2],
           out[3], out[4], out[5]);
    HIP_OK(hipFree(d_ws));
  }
  printf(fails ? "MISMATCH (%d)\n" : "C-ABI example OK (%d mismatches), HIP_VERSION of the library: %lld\n", fails,
         (long long)tsamd_hip_version());
  return fails != 0;
}
