// Shared helpers of the torch operator glue (ops_spmm.cpp, ops_storage.cpp, ops_sample.cpp): argument
// checks, dtype / reduce codes, workspaces, the current HIP stream.  Host-only C++ (g++), links torch.
#pragma once

#include <ATen/Context.h>
#include <ATen/hip/HIPContext.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/script.h>
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <mutex>
#include <torch/torch.h>

#include "tsamd.h"

namespace tsamd_ops {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using OptTensor = std::optional<Tensor>;

inline int dtype_code(const Tensor &t) {
  switch (t.scalar_type()) {
    case at::kFloat: return TSAMD_F32;
    case at::kDouble: return TSAMD_F64;
    case at::kHalf: return TSAMD_F16;
    case at::kBFloat16: return TSAMD_BF16;
    case at::kInt: return TSAMD_I32;
    case at::kLong: return TSAMD_I64;
    case at::kByte: return TSAMD_U8;
    case at::kChar: return TSAMD_I8;
    case at::kShort: return TSAMD_I16;
    default:
      TORCH_CHECK(false, "pytorch_sparse_amd: unsupported dtype ", t.scalar_type(),
                  " (supported: float32, float64, float16, bfloat16, int32, int64; uint8, int8, int16 in "
                  "the SpMM forward)");
  }
}

inline void check_status(int st, const char *what) {
  if (st == TSAMD_OK) return;
  if (st == TSAMD_ERR_HIP)
    TORCH_CHECK(false, what, " failed: HIP runtime error ", tsamd_last_hip_error());
  TORCH_CHECK(false, what, " failed: ", tsamd_status_string(st));
}

inline void check_gpu(const Tensor &t, const char *name) {
  TORCH_CHECK(t.device().is_cuda(), name,
              " must be a GPU (HIP) tensor: pytorch_sparse_amd has no CPU implementation");
}

inline void *current_stream(const Tensor &t) {
  return reinterpret_cast<void *>(c10::hip::getCurrentHIPStream(t.get_device()).stream());
}

inline Tensor workspace(size_t bytes, const Tensor &like) {
  return torch::empty({(int64_t)(bytes > 256 ? bytes : 256)},
                      like.options().dtype(torch::kUInt8).requires_grad(false));
}


inline const void *ptr_or_null(const OptTensor &t) { return t.has_value() ? t.value().data_ptr() : nullptr; }

inline int reduce_code(const std::string &r) {
  if (r == "sum" || r == "add") return TSAMD_SUM;
  if (r == "mean") return TSAMD_MEAN;
  if (r == "min") return TSAMD_MIN;
  if (r == "max") return TSAMD_MAX;
  TORCH_CHECK(false, "unknown reduce '", r, "'");
}

inline void check_index(const Tensor &t, const char *name) {
  check_gpu(t, name);
  TORCH_CHECK(t.scalar_type() == at::kLong && t.dim() == 1, name, " must be a 1-D int64 tensor");
}

inline bool needs_grad(const Tensor &t) { return torch::autograd::any_variable_requires_grad({t}); }

// defined in ops_storage.cpp, used by the samplers as well
std::tuple<Tensor, Tensor, Tensor, Tensor> select_segments(Tensor ptr, Tensor ind, Tensor idx, bool want_seg,
                                                           bool want_ind);

}  // namespace tsamd_ops
