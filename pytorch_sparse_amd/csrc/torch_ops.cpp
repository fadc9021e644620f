// torch operator boundary of the MI355X sparse-matmul hot path.
//
// Registers the reference's operator names with the reference's schemas
// (csrc/spmm.cpp:344-348, csrc/convert.cpp:46-48, csrc/version.cpp:40-41 of
// rusty1s/pytorch_sparse) on top of the C-ABI in include/tsamd.h:
//
//   torch_sparse::spmm_sum (Tensor? row, Tensor rowptr, Tensor col, Tensor? value,
//                           Tensor? colptr, Tensor? csr2csc, Tensor mat) -> Tensor
//   torch_sparse::spmm_mean(Tensor? row, Tensor rowptr, Tensor col, Tensor? value,
//                           Tensor? rowcount, Tensor? colptr, Tensor? csr2csc, Tensor mat) -> Tensor
//   torch_sparse::spmm_min / spmm_max(Tensor rowptr, Tensor col, Tensor? value, Tensor mat)
//                                                                      -> (Tensor, Tensor)
//   torch_sparse::ind2ptr(Tensor ind, int M) / ptr2ind(Tensor ptr, int E) -> Tensor
//   torch_sparse::cuda_version() -> int
//
// Like the reference, autograd lives inside the op (torch::autograd::Function) and the
// kernels run on the current stream without synchronising.  Unlike the reference there is
// no CPU branch: tensors must live on the GPU, anything else raises.
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

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using OptTensor = std::optional<Tensor>;

int dtype_code(const Tensor &t) {
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

void check_status(int st, const char *what) {
  if (st == TSAMD_OK) return;
  if (st == TSAMD_ERR_HIP)
    TORCH_CHECK(false, what, " failed: HIP runtime error ", tsamd_last_hip_error());
  TORCH_CHECK(false, what, " failed: ", tsamd_status_string(st));
}

void check_gpu(const Tensor &t, const char *name) {
  TORCH_CHECK(t.device().is_cuda(), name,
              " must be a GPU (HIP) tensor: pytorch_sparse_amd has no CPU implementation");
}

void *current_stream(const Tensor &t) {
  return reinterpret_cast<void *>(c10::hip::getCurrentHIPStream(t.get_device()).stream());
}

Tensor workspace(size_t bytes, const Tensor &like) {
  return torch::empty({(int64_t)(bytes > 256 ? bytes : 256)},
                      like.options().dtype(torch::kUInt8).requires_grad(false));
}

void check_index(const Tensor &t, const char *name);

const void *ptr_or_null(const OptTensor &t) { return t.has_value() ? t.value().data_ptr() : nullptr; }

int reduce_code(const std::string &r) {
  if (r == "sum" || r == "add") return TSAMD_SUM;
  if (r == "mean") return TSAMD_MEAN;
  if (r == "min") return TSAMD_MIN;
  if (r == "max") return TSAMD_MAX;
  TORCH_CHECK(false, "unknown reduce '", r, "'");
}

// ---- operand cache (include/tsamd.h: tsamd_spmm_cached) ------------------------------------------------
// One entry: the relabelled copy of the last dense operand that needed one.  A call may reuse it when the
// operand is provably the same tensor contents as far as torch can tell -- same storage object (held weakly:
// while it is alive its address cannot be handed to another tensor), same data pointer, same version
// counter, same shape / dtype / device / stream, same sparse pattern (col pointer and length) and reduction
// class -- and the kernel side re-checks a sampled fingerprint on the device.  TSAMD_OPERAND_CACHE=0 or
// torch.ops.tsamd.operand_cache(False) turns it off; inference tensors (no version counter) never use it.
struct OperandCache {
  std::mutex mu;
  bool enabled = true;
  c10::weak_intrusive_ptr<c10::StorageImpl> storage{c10::weak_intrusive_ptr<c10::StorageImpl>(
      c10::make_intrusive<c10::StorageImpl>(c10::StorageImpl::use_byte_size_t(), 0, c10::DataPtr(), nullptr, false))};
  const void *ptr = nullptr, *col_ptr = nullptr;
  uint32_t version = 0;
  std::vector<int64_t> sizes;
  int dtype = -1, red_class = -1, device = -1;
  int64_t E = -1;
  void *stream = nullptr;
  Tensor buf;
  int64_t hits = 0, fills = 0;
};

OperandCache &operand_cache_state() {
  static OperandCache c;
  static bool init = [] {
    const char *env = getenv("TSAMD_OPERAND_CACHE");
    if (env != nullptr && env[0] == '0') c.enabled = false;
    return true;
  }();
  (void)init;
  return c;
}

// torch.ops.tsamd.operand_cache(enable) -> [was enabled, hits, fills]; drops the cached copy
std::vector<int64_t> operand_cache_ctl(bool enable) {
  OperandCache &c = operand_cache_state();
  std::lock_guard<std::mutex> lock(c.mu);
  std::vector<int64_t> r = {c.enabled ? 1 : 0, c.hits, c.fills};
  c.enabled = enable;
  c.buf = Tensor();
  c.ptr = nullptr;
  return r;
}

// Forward launch: mirrors the argument checks of spmm_cpu.cpp:12-24 / spmm_cuda.cu:96-109.
std::tuple<Tensor, OptTensor> spmm_fw(const Tensor &rowptr, const Tensor &col,
                                      const OptTensor &opt_value, Tensor mat,
                                      const std::string &reduce, const OptTensor &opt_perm = std::nullopt) {
  check_gpu(rowptr, "rowptr");
  check_gpu(col, "col");
  if (opt_value.has_value()) check_gpu(opt_value.value(), "value");
  check_gpu(mat, "mat");
  TORCH_CHECK(rowptr.dim() == 1 && col.dim() == 1, "Input mismatch");
  TORCH_CHECK(rowptr.scalar_type() == at::kLong && col.scalar_type() == at::kLong,
              "rowptr and col must be int64");
  if (opt_value.has_value()) {
    TORCH_CHECK(opt_value.value().dim() == 1, "Input mismatch");
    TORCH_CHECK(opt_value.value().size(0) == col.size(0), "Input mismatch");
    TORCH_CHECK(opt_value.value().scalar_type() == mat.scalar_type(), "expected scalar type ",
                mat.scalar_type(), " but found ", opt_value.value().scalar_type());
  }
  TORCH_CHECK(mat.dim() >= 2, "Input mismatch");
  c10::hip::HIPGuard guard(rowptr.get_device());

  mat = mat.contiguous();
  Tensor rp = rowptr.contiguous(), c = col.contiguous();
  OptTensor value = opt_value.has_value() ? OptTensor(opt_value.value().contiguous()) : std::nullopt;
  auto sizes = mat.sizes().vec();
  const int64_t M = rp.numel() - 1, E = c.numel();
  const int64_t N = mat.size(-2), K = mat.size(-1);
  const int64_t B = (N * K) > 0 ? mat.numel() / (N * K) : 1;
  sizes[mat.dim() - 2] = M;
  Tensor out = torch::empty(sizes, mat.options().requires_grad(false));
  const int red = reduce_code(reduce);
  OptTensor arg_out = std::nullopt;
  int64_t *arg_ptr = nullptr;
  if (red == TSAMD_MIN || red == TSAMD_MAX) {
    arg_out = torch::empty(sizes, rp.options());
    arg_ptr = arg_out.value().data_ptr<int64_t>();
  }
  const int dt = dtype_code(mat);
  const size_t need = tsamd_spmm_workspace_bytes(dt, red, B, M, N, K, E);
  Tensor ws = workspace(need, mat);
  if (opt_perm.has_value()) {  // entries through a permutation (the CSC view in the backward)
    check_index(opt_perm.value(), "perm");
    TORCH_CHECK(opt_perm.value().numel() == E, "Input mismatch");
    Tensor perm = opt_perm.value().contiguous();
    check_status(tsamd_spmm_permuted(dt, red, rp.data_ptr<int64_t>(), c.data_ptr<int64_t>(),
                                     ptr_or_null(value), perm.data_ptr<int64_t>(), mat.data_ptr(),
                                     out.data_ptr(), arg_ptr, B, M, N, K, E, ws.data_ptr(),
                                     (size_t)ws.numel(), current_stream(mat)),
                 "tsamd_spmm_permuted");
    return std::make_tuple(out, arg_out);
  }
  check_status(tsamd_spmm(dt, red, rp.data_ptr<int64_t>(), c.data_ptr<int64_t>(),
                          ptr_or_null(value), mat.data_ptr(), out.data_ptr(), arg_ptr, B, M, N, K,
                          E, ws.data_ptr(), (size_t)ws.numel(), current_stream(mat)),
               "tsamd_spmm");
  return std::make_tuple(out, arg_out);
}

// the forward of the registered ops: tsamd_spmm, or tsamd_spmm_cached when this product copies its operand
std::tuple<Tensor, OptTensor> spmm_fw_cached(const Tensor &rowptr, const Tensor &col, const OptTensor &opt_value,
                                             const Tensor &mat_in, const std::string &reduce) {
  OperandCache &oc = operand_cache_state();
  const int red = reduce_code(reduce);
  bool eligible = oc.enabled && mat_in.defined() && mat_in.device().is_cuda() && mat_in.dim() >= 2 &&
                  mat_in.is_contiguous() && !mat_in.is_inference() && rowptr.device().is_cuda() &&
                  rowptr.dim() == 1 && col.dim() == 1 && col.is_contiguous() && rowptr.is_contiguous() &&
                  (reinterpret_cast<uintptr_t>(mat_in.data_ptr()) % 16) == 0;
  size_t cache_bytes = 0;
  int64_t B = 1, M = 0, N = 0, K = 0, E = 0;
  int dt = -1;
  if (eligible) {
    switch (mat_in.scalar_type()) {
      case at::kFloat: case at::kDouble: case at::kHalf: case at::kBFloat16: case at::kInt: case at::kLong:
        dt = dtype_code(mat_in);
        break;
      default: eligible = false;
    }
  }
  if (eligible) {
    M = rowptr.numel() - 1;
    E = col.numel();
    N = mat_in.size(-2);
    K = mat_in.size(-1);
    B = (N * K) > 0 ? mat_in.numel() / (N * K) : 1;
    cache_bytes = tsamd_spmm_operand_cache_bytes(dt, red, B, M, N, K, E);
  }
  if (!eligible || cache_bytes == 0) return spmm_fw(rowptr, col, opt_value, mat_in, reduce);

  // same checks as spmm_fw
  check_gpu(col, "col");
  if (opt_value.has_value()) check_gpu(opt_value.value(), "value");
  TORCH_CHECK(rowptr.scalar_type() == at::kLong && col.scalar_type() == at::kLong, "rowptr and col must be int64");
  if (opt_value.has_value()) {
    TORCH_CHECK(opt_value.value().dim() == 1, "Input mismatch");
    TORCH_CHECK(opt_value.value().size(0) == col.size(0), "Input mismatch");
    TORCH_CHECK(opt_value.value().scalar_type() == mat_in.scalar_type(), "expected scalar type ",
                mat_in.scalar_type(), " but found ", opt_value.value().scalar_type());
  }
  c10::hip::HIPGuard guard(rowptr.get_device());
  OptTensor value = opt_value.has_value() ? OptTensor(opt_value.value().contiguous()) : std::nullopt;
  auto sizes = mat_in.sizes().vec();
  sizes[mat_in.dim() - 2] = M;
  Tensor out = torch::empty(sizes, mat_in.options().requires_grad(false));
  OptTensor arg_out = std::nullopt;
  int64_t *arg_ptr = nullptr;
  if (red == TSAMD_MIN || red == TSAMD_MAX) {
    arg_out = torch::empty(sizes, rowptr.options());
    arg_ptr = arg_out.value().data_ptr<int64_t>();
  }
  void *stream = current_stream(mat_in);
  c10::StorageImpl *simpl = mat_in.storage().unsafeGetStorageImpl();
  const uint32_t version = mat_in.unsafeGetTensorImpl()->version_counter().current_version();
  const int red_class = (red == TSAMD_MIN || red == TSAMD_MAX) ? 1 : 0;

  std::lock_guard<std::mutex> lock(oc.mu);
  bool valid = false;
  if (oc.buf.defined() && oc.ptr == mat_in.data_ptr() && oc.version == version && oc.dtype == dt &&
      oc.red_class == red_class && oc.device == mat_in.get_device() && oc.stream == stream &&
      oc.col_ptr == col.data_ptr() && oc.E == E && oc.sizes == mat_in.sizes().vec() &&
      (size_t)oc.buf.numel() >= cache_bytes) {
    auto locked = oc.storage.lock();  // the storage the copy was made from is still alive and is this one
    valid = locked && locked.get() == simpl;
  }
  if (!valid) {
    if (!oc.buf.defined() || (size_t)oc.buf.numel() < cache_bytes || oc.buf.get_device() != mat_in.get_device()) {
      oc.buf = Tensor();  // release before the new allocation
      oc.buf = workspace(cache_bytes, mat_in);
    }
    oc.storage = c10::weak_intrusive_ptr<c10::StorageImpl>(
        c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(simpl));
    oc.ptr = mat_in.data_ptr();
    oc.version = version;
    oc.dtype = dt;
    oc.red_class = red_class;
    oc.device = mat_in.get_device();
    oc.stream = stream;
    oc.col_ptr = col.data_ptr();
    oc.E = E;
    oc.sizes = mat_in.sizes().vec();
    ++oc.fills;
  } else {
    ++oc.hits;
  }
  Tensor ws = workspace(tsamd_spmm_cached_workspace_bytes(dt, red, B, M, N, K, E), mat_in);
  check_status(tsamd_spmm_cached(dt, red, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(), ptr_or_null(value),
                                 mat_in.data_ptr(), out.data_ptr(), arg_ptr, B, M, N, K, E, ws.data_ptr(),
                                 (size_t)ws.numel(), oc.buf.data_ptr(), (size_t)oc.buf.numel(), valid ? 1 : 0, stream),
               "tsamd_spmm_cached");
  return std::make_tuple(out, arg_out);
}

Tensor spmm_value_bw(const Tensor &row, const Tensor &rowptr, const Tensor &col, Tensor mat,
                     Tensor grad, const std::string &reduce) {
  check_gpu(rowptr, "rowptr");
  check_gpu(mat, "mat");
  check_gpu(grad, "grad");
  c10::hip::HIPGuard guard(rowptr.get_device());
  mat = mat.contiguous();
  grad = grad.contiguous();
  check_index(rowptr, "rowptr");
  check_index(col, "col");
  check_index(row, "row");
  Tensor rp = rowptr.contiguous(), c = col.contiguous(), r = row.contiguous();
  const int64_t M = grad.size(-2), N = mat.size(-2), K = mat.size(-1), E = col.numel();
  const int64_t B = (N * K) > 0 ? mat.numel() / (N * K) : 1;
  Tensor out = torch::empty({E}, grad.options().requires_grad(false));
  check_status(tsamd_spmm_value_bw(dtype_code(mat), reduce_code(reduce), r.data_ptr<int64_t>(),
                                   rp.data_ptr<int64_t>(), c.data_ptr<int64_t>(),
                                   mat.data_ptr(), grad.data_ptr(), out.data_ptr(), B, M, N, K, E,
                                   current_stream(mat)),
               "tsamd_spmm_value_bw");
  return out;
}

bool needs_grad(const Tensor &t) { return torch::autograd::any_variable_requires_grad({t}); }

// ---- sum / mean ---------------------------------------------------------------------------
// One Function serves both: `mean` selects the divisor handling in both directions.
// Saved tensors: row, rowptr, col, value, rowcount, colptr, csr2csc, mat.
class SpmmAddFunction : public torch::autograd::Function<SpmmAddFunction> {
 public:
  static variable_list forward(AutogradContext *ctx, OptTensor opt_row, Tensor rowptr, Tensor col,
                               Tensor value, OptTensor opt_rowcount, OptTensor opt_colptr,
                               OptTensor opt_csr2csc, Tensor mat, bool has_value, bool mean) {
    if (has_value && needs_grad(value)) TORCH_CHECK(opt_row.has_value(), "Argument `row` is missing");
    if (needs_grad(mat)) {
      TORCH_CHECK(opt_row.has_value(), "Argument `row` is missing");
      if (mean) TORCH_CHECK(opt_rowcount.has_value(), "Argument `rowcount` is missing");
      TORCH_CHECK(opt_colptr.has_value(), "Argument `colptr` is missing");
      TORCH_CHECK(opt_csr2csc.has_value(), "Argument `csr2csc` is missing");
    }
    OptTensor v = has_value ? OptTensor(value) : std::nullopt;
    Tensor out = std::get<0>(spmm_fw_cached(rowptr, col, v, mat, mean ? "mean" : "sum"));
    ctx->saved_data["has_value"] = has_value;
    ctx->saved_data["mean"] = mean;
    // absent optionals are parked as `col` (any tensor will do; they are never read then)
    ctx->save_for_backward({opt_row.value_or(col), rowptr, col, value, opt_rowcount.value_or(col),
                            opt_colptr.value_or(col), opt_csr2csc.value_or(col), mat});
    return {out};
  }

  static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
    const bool has_value = ctx->saved_data["has_value"].toBool();
    const bool mean = ctx->saved_data["mean"].toBool();
    Tensor grad_out = grad_outs[0];
    auto s = ctx->get_saved_variables();
    Tensor row = s[0], rowptr = s[1], col = s[2], value = s[3], rowcount = s[4], colptr = s[5],
           csr2csc = s[6], mat = s[7];

    Tensor grad_value, grad_mat;
    if (has_value && needs_grad(value))
      grad_value = spmm_value_bw(row, rowptr, col, mat, grad_out, mean ? "mean" : "sum");

    if (needs_grad(mat)) {
      // grad_mat = A^T * grad_out: the CSC arrays are the CSR of A^T; per-edge weights are
      // value (sum) or value / max(deg(row), 1) (mean) in CSC order.
      if (!mean) {
        // sum: the kernel reads (row, value) THROUGH csr2csc -- no row.index_select(0, csr2csc) /
        // value.index_select(0, csr2csc) temporaries as in the reference (spmm.cpp:104-106)
        OptTensor w = has_value ? OptTensor(value.detach()) : std::nullopt;
        grad_mat = std::get<0>(spmm_fw(colptr, row, w, grad_out, "sum", csr2csc));
      } else {
        Tensor row_t = row.index_select(0, csr2csc);
        Tensor cnt = rowcount.index_select(0, row_t).to(mat.scalar_type()).clamp_min_(1);
        Tensor w = has_value ? value.detach().index_select(0, csr2csc).div_(cnt) : cnt.reciprocal_();
        grad_mat = std::get<0>(spmm_fw(colptr, row_t, w, grad_out, "sum"));
      }
    }
    return {Tensor(), Tensor(), Tensor(), grad_value, Tensor(), Tensor(), Tensor(), grad_mat,
            Tensor(), Tensor()};
  }
};

// ---- min / max ----------------------------------------------------------------------------
// With the CSC arrays of the matrix (colptr, csr2csc, row -- SparseTensor.matmul hands them over when a
// gradient w.r.t. `mat` is wanted, exactly as it does for sum) the backward is the atomic-free pull of
// tsamd_spmm_minmax_bw_csc; the bare reference op (rowptr, col, value, mat) keeps the scatter kernel.
class SpmmMinMaxFunction : public torch::autograd::Function<SpmmMinMaxFunction> {
 public:
  static variable_list forward(AutogradContext *ctx, Tensor rowptr, Tensor col, Tensor value,
                               Tensor mat, bool has_value, bool is_max, OptTensor opt_colptr,
                               OptTensor opt_csr2csc, OptTensor opt_row) {
    OptTensor v = has_value ? OptTensor(value) : std::nullopt;
    auto res = spmm_fw_cached(rowptr, col, v, mat, is_max ? "max" : "min");
    Tensor out = std::get<0>(res), arg_out = std::get<1>(res).value();
    const bool has_csc = opt_colptr.has_value() && opt_csr2csc.has_value() && opt_row.has_value();
    if (has_csc) {
      check_index(opt_colptr.value(), "colptr");
      check_index(opt_csr2csc.value(), "csr2csc");
      check_index(opt_row.value(), "row");
      TORCH_CHECK(opt_csr2csc.value().numel() == col.numel() && opt_row.value().numel() == col.numel() &&
                      opt_colptr.value().numel() == mat.size(-2) + 1,
                  "Input mismatch");
    }
    ctx->saved_data["has_value"] = has_value;
    ctx->saved_data["has_csc"] = has_csc;
    // the reference saves {col, value, mat, arg_out} (spmm.cpp:199); rowptr is kept as well so that
    // grad_value can be accumulated row by row (tsamd.h)
    ctx->save_for_backward({col, value, mat, arg_out, rowptr, opt_colptr.value_or(col),
                            opt_csr2csc.value_or(col), opt_row.value_or(col)});
    ctx->mark_non_differentiable({arg_out});
    return {out, arg_out};
  }

  static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
    const bool has_value = ctx->saved_data["has_value"].toBool();
    const bool has_csc = ctx->saved_data["has_csc"].toBool();
    Tensor grad_out = grad_outs[0].contiguous();
    auto s = ctx->get_saved_variables();
    Tensor col = s[0].contiguous(), value = s[1].contiguous(), mat = s[2].contiguous(),
           arg_out = s[3].contiguous(), rowptr = s[4].contiguous();
    const bool want_value = has_value && needs_grad(value);
    const bool want_mat = needs_grad(mat);
    Tensor grad_value, grad_mat;
    if (want_value || want_mat) {
      c10::hip::HIPGuard guard(mat.get_device());
      const int64_t N = mat.size(-2), K = mat.size(-1), M = grad_out.size(-2), E = col.numel();
      const int64_t B = (N * K) > 0 ? mat.numel() / (N * K) : 1;
      if (want_value) grad_value = torch::empty({E}, mat.options().requires_grad(false));
      if (want_mat) grad_mat = torch::empty_like(mat, mat.options().requires_grad(false));
      const int dt = dtype_code(mat);
      int st = TSAMD_ERR_UNSUPPORTED;
      // The pull is deterministic and 27 % faster for grad_mat alone (config 3: 1.8 vs 2.5 ms).  When
      // grad_value is wanted as well the scatter kernel gets it nearly for free (fused, +0.1-0.3 ms) while the
      // pull pays a masked SDDMM (+1.0 ms): 2.55 vs 2.85 ms -- so that case only takes the pull when
      // torch.use_deterministic_algorithms(True) asks for reproducible gradients.
      const bool pull = has_csc && want_mat && (!want_value || at::globalContext().deterministicAlgorithms());
      if (pull) {
        Tensor colptr = s[5].contiguous(), csr2csc = s[6].contiguous(), row = s[7].contiguous();
        Tensor ws = workspace(tsamd_spmm_minmax_bw_csc_workspace_bytes(dt, B, M, N, K, E), mat);
        st = tsamd_spmm_minmax_bw_csc(dt, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                      has_value ? value.data_ptr() : nullptr, mat.data_ptr(),
                                      grad_out.data_ptr(), arg_out.data_ptr<int64_t>(),
                                      colptr.data_ptr<int64_t>(), csr2csc.data_ptr<int64_t>(),
                                      row.data_ptr<int64_t>(), want_value ? grad_value.data_ptr() : nullptr,
                                      grad_mat.data_ptr(), B, M, N, K, E, ws.data_ptr(), (size_t)ws.numel(),
                                      current_stream(mat));
        if (st != TSAMD_ERR_UNSUPPORTED) check_status(st, "tsamd_spmm_minmax_bw_csc");
      }
      if (st == TSAMD_ERR_UNSUPPORTED) {  // no CSC arrays (bare op), or sizes beyond the pull kernel's 32-bit ids
        Tensor ws = workspace(tsamd_spmm_minmax_bw_workspace_bytes(dt, B, N, K, E), mat);
        check_status(
            tsamd_spmm_minmax_bw(dt, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                 has_value ? value.data_ptr() : nullptr,
                                 mat.data_ptr(), grad_out.data_ptr(), arg_out.data_ptr<int64_t>(),
                                 want_value ? grad_value.data_ptr() : nullptr,
                                 want_mat ? grad_mat.data_ptr() : nullptr, B, M, N, K, E,
                                 ws.data_ptr(), (size_t)ws.numel(), current_stream(mat)),
            "tsamd_spmm_minmax_bw");
      }
    }
    return {Tensor(), Tensor(), grad_value, grad_mat, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ---- relabelled ("channel-camping free") layout, end to end (include/tsamd.h) -----------------
// position of every id of `ids` in a relabelled [n, *] matrix; ids == None: the whole map [n]
Tensor relabel_ids(OptTensor ids, int64_t n, Tensor like) {
  check_gpu(like, "like");
  c10::hip::HIPGuard guard(like.get_device());
  Tensor src;
  int64_t count = n;
  if (ids.has_value()) {
    check_index(ids.value(), "ids");
    src = ids.value().contiguous();
    count = src.numel();
  }
  Tensor out = torch::empty({count}, like.options().dtype(at::kLong).requires_grad(false));
  check_status(tsamd_relabel_ids(ids.has_value() ? src.data_ptr<int64_t>() : nullptr, count, n,
                                 out.data_ptr<int64_t>(), current_stream(like)),
               "tsamd_relabel_ids");
  return out;
}

std::tuple<Tensor, OptTensor> spmm_relabelled_fw(const Tensor &rowptr, const Tensor &col_h,
                                                 const OptTensor &opt_value, Tensor mat_h,
                                                 const std::string &reduce) {
  check_index(rowptr, "rowptr");
  check_index(col_h, "col_h");
  check_gpu(mat_h, "mat");
  TORCH_CHECK(mat_h.dim() >= 2, "Input mismatch");
  if (opt_value.has_value()) {
    check_gpu(opt_value.value(), "value");
    TORCH_CHECK(opt_value.value().dim() == 1 && opt_value.value().size(0) == col_h.size(0), "Input mismatch");
    TORCH_CHECK(opt_value.value().scalar_type() == mat_h.scalar_type(), "expected scalar type ",
                mat_h.scalar_type(), " but found ", opt_value.value().scalar_type());
  }
  c10::hip::HIPGuard guard(rowptr.get_device());
  mat_h = mat_h.contiguous();
  Tensor rp = rowptr.contiguous(), c = col_h.contiguous();
  OptTensor value = opt_value.has_value() ? OptTensor(opt_value.value().contiguous()) : std::nullopt;
  auto sizes = mat_h.sizes().vec();
  const int64_t M = rp.numel() - 1, E = c.numel();
  const int64_t N = mat_h.size(-2), K = mat_h.size(-1);
  const int64_t B = (N * K) > 0 ? mat_h.numel() / (N * K) : 1;
  sizes[mat_h.dim() - 2] = M;
  Tensor out = torch::empty(sizes, mat_h.options().requires_grad(false));
  const int red = reduce_code(reduce);
  OptTensor arg_out = std::nullopt;
  int64_t *arg_ptr = nullptr;
  if (red == TSAMD_MIN || red == TSAMD_MAX) {
    arg_out = torch::empty(sizes, rp.options());
    arg_ptr = arg_out.value().data_ptr<int64_t>();
  }
  const int dt = dtype_code(mat_h);
  Tensor ws = workspace(tsamd_spmm_relabelled_workspace_bytes(dt, red, B, M, N, K, E), mat_h);
  check_status(tsamd_spmm_relabelled(dt, red, rp.data_ptr<int64_t>(), c.data_ptr<int64_t>(),
                                     ptr_or_null(value), mat_h.data_ptr(), out.data_ptr(), arg_ptr, B, M,
                                     N, K, E, ws.data_ptr(), (size_t)ws.numel(), current_stream(mat_h)),
               "tsamd_spmm_relabelled");
  return std::make_tuple(out, arg_out);
}

// sum / mean in the relabelled layout with both gradients; the backward works in the same layout
// (grad_out arrives relabelled over M, grad_mat leaves relabelled over N).
// Saved: row, rowptr, col_h, value, rowcount, colptr, csr2csc, mat_h.
class SpmmRelabelledFunction : public torch::autograd::Function<SpmmRelabelledFunction> {
 public:
  static variable_list forward(AutogradContext *ctx, OptTensor opt_row, Tensor rowptr, Tensor col_h,
                               Tensor value, OptTensor opt_rowcount, OptTensor opt_colptr,
                               OptTensor opt_csr2csc, Tensor mat_h, bool has_value, bool mean) {
    if ((has_value && needs_grad(value)) || needs_grad(mat_h)) {
      TORCH_CHECK(opt_row.has_value(), "Argument `row` is missing");
      if (mean) TORCH_CHECK(opt_rowcount.has_value(), "Argument `rowcount` is missing");
    }
    if (needs_grad(mat_h)) {
      TORCH_CHECK(opt_colptr.has_value(), "Argument `colptr` is missing");
      TORCH_CHECK(opt_csr2csc.has_value(), "Argument `csr2csc` is missing");
    }
    OptTensor v = has_value ? OptTensor(value) : std::nullopt;
    Tensor out = std::get<0>(spmm_relabelled_fw(rowptr, col_h, v, mat_h, mean ? "mean" : "sum"));
    ctx->saved_data["has_value"] = has_value;
    ctx->saved_data["mean"] = mean;
    ctx->save_for_backward({opt_row.value_or(col_h), rowptr, col_h, value, opt_rowcount.value_or(col_h),
                            opt_colptr.value_or(col_h), opt_csr2csc.value_or(col_h), mat_h});
    return {out};
  }

  static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
    const bool has_value = ctx->saved_data["has_value"].toBool();
    const bool mean = ctx->saved_data["mean"].toBool();
    Tensor grad_h = grad_outs[0].contiguous();
    auto s = ctx->get_saved_variables();
    Tensor row = s[0], rowptr = s[1], col_h = s[2], value = s[3], rowcount = s[4], colptr = s[5],
           csr2csc = s[6], mat_h = s[7];
    const int64_t M = rowptr.numel() - 1;
    Tensor grad_value, grad_mat;
    if (has_value && needs_grad(value)) {
      // SDDMM over the pattern with both operands gathered at relabelled positions
      Tensor row_h = relabel_ids(row, M, row);
      grad_value = spmm_value_bw(row_h, rowptr, col_h, mat_h, grad_h, "sum");
      if (mean) grad_value = grad_value / rowcount.index_select(0, row).to(grad_value.scalar_type()).clamp_min_(1);
    }
    if (needs_grad(mat_h)) {
      Tensor row_t = row.index_select(0, csr2csc);
      OptTensor w = std::nullopt;
      if (mean) {
        Tensor cnt = rowcount.index_select(0, row_t).to(mat_h.scalar_type()).clamp_min_(1);
        w = has_value ? value.detach().index_select(0, csr2csc).div_(cnt) : cnt.reciprocal_();
      } else if (has_value) {
        w = value.detach().index_select(0, csr2csc);
      }
      grad_mat = std::get<0>(spmm_relabelled_fw(colptr, relabel_ids(row_t, M, row_t), w, grad_h, "sum"));
    }
    return {Tensor(), Tensor(), Tensor(), grad_value, Tensor(), Tensor(), Tensor(), grad_mat,
            Tensor(), Tensor()};
  }
};

// (out_h, arg_out_h or an empty tensor).  min / max are forward only in this layout.
std::tuple<Tensor, Tensor> spmm_relabelled(OptTensor opt_row, Tensor rowptr, Tensor col_h,
                                           OptTensor opt_value, OptTensor opt_rowcount,
                                           OptTensor opt_colptr, OptTensor opt_csr2csc, Tensor mat_h,
                                           std::string reduce) {
  const int red = reduce_code(reduce);
  if (red == TSAMD_MIN || red == TSAMD_MAX) {
    TORCH_CHECK(!needs_grad(mat_h) && !(opt_value.has_value() && needs_grad(opt_value.value())),
                "spmm_relabelled: min / max have no backward in the relabelled layout; use "
                "torch.ops.torch_sparse.spmm_", reduce, " on the plain layout for training");
    auto r = spmm_relabelled_fw(rowptr, col_h, opt_value, mat_h, reduce);
    return std::make_tuple(std::get<0>(r), std::get<1>(r).value());
  }
  Tensor value = opt_value.value_or(col_h);
  Tensor out = SpmmRelabelledFunction::apply(opt_row, rowptr, col_h, value, opt_rowcount, opt_colptr,
                                             opt_csr2csc, mat_h, opt_value.has_value(),
                                             red == TSAMD_MEAN)[0];
  return std::make_tuple(out, torch::empty({0}, rowptr.options()));
}

// dst = src[idx] for a 2-D row-major matrix (send buffer of the sharded SpMM's row exchange)
Tensor gather_rows(Tensor src, Tensor idx) {
  check_gpu(src, "src");
  check_index(idx, "idx");
  TORCH_CHECK(src.dim() == 2, "gather_rows: src must be 2-D");
  c10::hip::HIPGuard guard(src.get_device());
  src = src.contiguous();
  idx = idx.contiguous();
  Tensor out = torch::empty({idx.numel(), src.size(1)}, src.options().requires_grad(false));
  check_status(tsamd_gather_rows(src.data_ptr(), idx.data_ptr<int64_t>(), out.data_ptr(), idx.numel(),
                                 src.size(0), src.size(1) * (int64_t)src.element_size(),
                                 current_stream(src)),
               "tsamd_gather_rows");
  return out;
}

// ---- registered entry points (reference signatures) -----------------------------------------
Tensor spmm_sum(OptTensor opt_row, Tensor rowptr, Tensor col, OptTensor opt_value,
                OptTensor opt_colptr, OptTensor opt_csr2csc, Tensor mat) {
  Tensor value = opt_value.value_or(col);
  return SpmmAddFunction::apply(opt_row, rowptr, col, value, std::nullopt, opt_colptr, opt_csr2csc,
                                mat, opt_value.has_value(), false)[0];
}

Tensor spmm_mean(OptTensor opt_row, Tensor rowptr, Tensor col, OptTensor opt_value,
                 OptTensor opt_rowcount, OptTensor opt_colptr, OptTensor opt_csr2csc, Tensor mat) {
  Tensor value = opt_value.value_or(col);
  return SpmmAddFunction::apply(opt_row, rowptr, col, value, opt_rowcount, opt_colptr, opt_csr2csc,
                                mat, opt_value.has_value(), true)[0];
}

std::tuple<Tensor, Tensor> spmm_min(Tensor rowptr, Tensor col, OptTensor opt_value, Tensor mat) {
  auto r = SpmmMinMaxFunction::apply(rowptr, col, opt_value.value_or(col), mat,
                                     opt_value.has_value(), false, std::nullopt, std::nullopt, std::nullopt);
  return std::make_tuple(r[0], r[1]);
}

std::tuple<Tensor, Tensor> spmm_max(Tensor rowptr, Tensor col, OptTensor opt_value, Tensor mat) {
  auto r = SpmmMinMaxFunction::apply(rowptr, col, opt_value.value_or(col), mat,
                                     opt_value.has_value(), true, std::nullopt, std::nullopt, std::nullopt);
  return std::make_tuple(r[0], r[1]);
}

// tsamd::spmm_minmax(Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr, Tensor? csr2csc, Tensor? row,
//                    Tensor mat, bool is_max) -> (Tensor, Tensor)
// spmm_min / spmm_max with the CSC arrays of the matrix: same forward, atomic-free deterministic backward.
std::tuple<Tensor, Tensor> spmm_minmax(Tensor rowptr, Tensor col, OptTensor opt_value, OptTensor opt_colptr,
                                       OptTensor opt_csr2csc, OptTensor opt_row, Tensor mat, bool is_max) {
  auto r = SpmmMinMaxFunction::apply(rowptr, col, opt_value.value_or(col), mat, opt_value.has_value(), is_max,
                                     opt_colptr, opt_csr2csc, opt_row);
  return std::make_tuple(r[0], r[1]);
}

Tensor ind2ptr(Tensor ind, int64_t M) {
  check_gpu(ind, "ind");
  TORCH_CHECK(ind.scalar_type() == at::kLong, "ind must be int64");
  c10::hip::HIPGuard guard(ind.get_device());
  ind = ind.contiguous();
  Tensor out = torch::empty({M + 1}, ind.options());
  check_status(tsamd_ind2ptr(ind.data_ptr<int64_t>(), M, ind.numel(), out.data_ptr<int64_t>(),
                             current_stream(ind)),
               "tsamd_ind2ptr");
  return out;
}

Tensor ptr2ind(Tensor ptr, int64_t E) {
  check_gpu(ptr, "ptr");
  TORCH_CHECK(ptr.scalar_type() == at::kLong, "ptr must be int64");
  c10::hip::HIPGuard guard(ptr.get_device());
  ptr = ptr.contiguous();
  Tensor out = torch::empty({E}, ptr.options());
  check_status(tsamd_ptr2ind(ptr.data_ptr<int64_t>(), ptr.numel() - 1, E, out.data_ptr<int64_t>(),
                             current_stream(ptr)),
               "tsamd_ptr2ind");
  return out;
}

int64_t cuda_version() { return tsamd_hip_version(); }

// ---- fused storage ops (no reference op of the same name: they replace Python/ATen
//      compositions of torch_sparse/storage.py, see include/tsamd.h) --------------------------
void check_index(const Tensor &t, const char *name) {
  check_gpu(t, name);
  TORCH_CHECK(t.scalar_type() == at::kLong && t.dim() == 1, name, " must be a 1-D int64 tensor");
}

// -> int64[2] on the device: {#descents, #adjacent duplicates} of key = row * N + col
Tensor coo_order(Tensor row, Tensor col, int64_t N) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  Tensor counts = torch::empty({2}, row.options());
  check_status(tsamd_coo_order(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), row.numel(), N,
                               counts.data_ptr<int64_t>(), current_stream(row)),
               "tsamd_coo_order");
  return counts;
}

// -> int64[4] on the device: {#descents, #adjacent duplicates, max row id, max col id}
Tensor coo_check(Tensor row, Tensor col) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  Tensor counts = torch::empty({4}, row.options());
  check_status(tsamd_coo_check(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), row.numel(),
                               counts.data_ptr<int64_t>(), current_stream(row)),
               "tsamd_coo_check");
  return counts;
}

// sort_coo decided on the device, no host sync: -> (row_sorted, col_sorted, perm, counts[2] on the device)
std::tuple<Tensor, Tensor, Tensor, Tensor> sort_coo_auto(Tensor row, Tensor col, int64_t M, int64_t N) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  Tensor perm = torch::empty({E}, row.options()), row_s = torch::empty({E}, row.options()),
         col_s = torch::empty({E}, row.options()), counts = torch::empty({2}, row.options());
  Tensor ws = workspace(tsamd_sort_coo_workspace_bytes(E), row);
  check_status(tsamd_sort_coo_auto(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N,
                                   row_s.data_ptr<int64_t>(), col_s.data_ptr<int64_t>(), perm.data_ptr<int64_t>(),
                                   counts.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), current_stream(row)),
               "tsamd_sort_coo_auto");
  return std::make_tuple(row_s, col_s, perm, counts);
}

// stable sort by row * N + col -> (row_sorted, col_sorted, perm); with index=false only perm
std::tuple<Tensor, Tensor, Tensor> sort_coo(Tensor row, Tensor col, int64_t M, int64_t N,
                                            bool index) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  Tensor perm = torch::empty({E}, row.options());
  Tensor row_s = index ? torch::empty({E}, row.options()) : torch::empty({0}, row.options());
  Tensor col_s = index ? torch::empty({E}, row.options()) : torch::empty({0}, row.options());
  Tensor ws = workspace(tsamd_sort_coo_workspace_bytes(E), row);
  check_status(tsamd_sort_coo(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N,
                              index ? row_s.data_ptr<int64_t>() : nullptr,
                              index ? col_s.data_ptr<int64_t>() : nullptr, perm.data_ptr<int64_t>(),
                              ws.data_ptr(), (size_t)ws.numel(), current_stream(row)),
               "tsamd_sort_coo");
  return std::make_tuple(row_s, col_s, perm);
}

// sorted (row, col) -> (row_u[E], col_u[E], seg_ptr[E+1], nnz[1]); only the first nnz (+1)
// entries are meaningful, nnz lives on the device.
std::tuple<Tensor, Tensor, Tensor, Tensor> coalesce_index(Tensor row, Tensor col) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  Tensor row_u = torch::empty({E}, row.options()), col_u = torch::empty({E}, row.options());
  Tensor seg = torch::empty({E + 1}, row.options()), nnz = torch::empty({1}, row.options());
  Tensor ws = workspace(tsamd_coalesce_workspace_bytes(E), row);
  check_status(tsamd_coalesce_index(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E,
                                    row_u.data_ptr<int64_t>(), col_u.data_ptr<int64_t>(),
                                    seg.data_ptr<int64_t>(), nnz.data_ptr<int64_t>(), ws.data_ptr(),
                                    (size_t)ws.numel(), current_stream(row)),
               "tsamd_coalesce_index");
  return std::make_tuple(row_u, col_u, seg, nnz);
}

// out[j] = REDUCE_{i in [seg_ptr[j], seg_ptr[j+1])} value[perm ? perm[i] : i]   (dim 0)
// balanced = false: one thread per (segment, feature) -- right for the short runs of duplicates
//   that coalesce reduces.
// balanced = true: the segments are the rows / columns of a matrix (hubs with 1e5+ entries on
//   power-law graphs, where one thread per segment took 13-37 ms for 40 M entries): the
//   entry-balanced kernel of csrc/segreduce.hip.
Tensor segment_reduce(Tensor value, OptTensor perm, Tensor seg_ptr, int64_t nseg, std::string reduce,
                      bool balanced) {
  check_gpu(value, "value");
  check_index(seg_ptr, "seg_ptr");
  if (perm.has_value()) check_index(perm.value(), "perm");
  TORCH_CHECK(value.dim() >= 1, "value must have at least one dimension");
  TORCH_CHECK(seg_ptr.numel() >= nseg + 1, "seg_ptr shorter than nseg + 1");
  c10::hip::HIPGuard guard(value.get_device());
  value = value.contiguous();
  seg_ptr = seg_ptr.contiguous();
  auto sizes = value.sizes().vec();
  const int64_t E = perm.has_value() ? perm.value().numel() : value.size(0);
  const int64_t D = value.size(0) > 0 ? value.numel() / value.size(0) : 1;
  sizes[0] = nseg;
  const int red = reduce_code(reduce);
  if (E == 0) return torch::zeros(sizes, value.options().requires_grad(false));  // every segment is empty
  Tensor out = torch::empty(sizes, value.options().requires_grad(false));
  Tensor p = perm.has_value() ? perm.value().contiguous() : Tensor();
  const int64_t *pp = perm.has_value() ? p.data_ptr<int64_t>() : nullptr;
  if (balanced && E >= 32768 && D <= 65535) {
    const int dt = dtype_code(value);
    Tensor ws = workspace(tsamd_segment_reduce_balanced_workspace_bytes(dt, E, D), value);
    check_status(tsamd_segment_reduce_balanced(dt, red, value.data_ptr(), pp, seg_ptr.data_ptr<int64_t>(),
                                               nseg, E, D, out.data_ptr(), ws.data_ptr(),
                                               (size_t)ws.numel(), current_stream(value)),
                 "tsamd_segment_reduce_balanced");
    return out;
  }
  check_status(tsamd_segment_reduce(dtype_code(value), red, value.data_ptr(), pp,
                                    seg_ptr.data_ptr<int64_t>(), nseg, D, out.data_ptr(),
                                    current_stream(value)),
               "tsamd_segment_reduce");
  return out;
}

// C = A * B on CSR operands -> (rowptrC, colC, valueC); valueC is empty unless with_value.
// Count first, write once (csrc/spspmm.hip); two host syncs (size classes, nnz(C)) because the
// scratch of oversized rows and the output size are data dependent.
std::tuple<Tensor, Tensor, Tensor> spspmm(Tensor rowptrA, Tensor colA, OptTensor valA,
                                          Tensor rowptrB, Tensor colB, OptTensor valB, int64_t N,
                                          bool with_value) {
  check_index(rowptrA, "rowptrA");
  check_index(colA, "colA");
  check_index(rowptrB, "rowptrB");
  check_index(colB, "colB");
  c10::hip::HIPGuard guard(rowptrA.get_device());
  rowptrA = rowptrA.contiguous();
  colA = colA.contiguous();
  rowptrB = rowptrB.contiguous();
  colB = colB.contiguous();
  auto vdtype = valA.has_value() ? valA.value().scalar_type()
                                 : (valB.has_value() ? valB.value().scalar_type() : at::kFloat);
  TORCH_CHECK(vdtype == at::kFloat || vdtype == at::kDouble,
              "spspmm: only float32 and float64 values are supported (got ", vdtype, ")");
  if (valA.has_value() && valB.has_value())
    TORCH_CHECK(valA.value().scalar_type() == valB.value().scalar_type(), "spspmm: dtype mismatch");
  Tensor va = valA.has_value() ? valA.value().contiguous() : Tensor();
  Tensor vb = valB.has_value() ? valB.value().contiguous() : Tensor();
  const int dt = vdtype == at::kFloat ? TSAMD_F32 : TSAMD_F64;
  const int64_t M = rowptrA.numel() - 1;
  TORCH_CHECK(rowptrB.numel() - 1 >= 0 && M >= 0, "spspmm: bad rowptr");
  auto iopt = rowptrA.options();
  auto vopt = iopt.dtype(vdtype);
  void *stream = current_stream(rowptrA);

  Tensor prod = torch::empty({M + 1}, iopt), bins = torch::empty({2 * M + 1}, iopt);
  Tensor stats = torch::empty({8}, iopt);
  Tensor colB32 = torch::empty({colB.numel()}, iopt.dtype(at::kInt));  // 32-bit copy for the gathers
  uint32_t *cb32 = reinterpret_cast<uint32_t *>(colB32.data_ptr<int32_t>());
  check_status(tsamd_spspmm_plan(rowptrA.data_ptr<int64_t>(), colA.data_ptr<int64_t>(),
                                 rowptrB.data_ptr<int64_t>(), colB.data_ptr<int64_t>(), colB.numel(), M,
                                 prod.data_ptr<int64_t>(), bins.data_ptr<int64_t>(), cb32,
                                 stats.data_ptr<int64_t>(), stream),
               "tsamd_spspmm_plan");
  Tensor h = stats.cpu();  // sync 1: grid sizes, workspace of the rows beyond the LDS capacity
  const int64_t *hs = h.data_ptr<int64_t>();
  const int64_t n_medium = hs[2], n_large = hs[3], P_large = hs[4];
  // the large-row path counts the products of a (row, column range) bin in 32-bit LDS words
  TORCH_CHECK(hs[5] < ((int64_t)1 << 31), "spspmm: a row of the product has ", hs[5],
              " intermediate products; rows of 2^31 or more are not supported");

  const size_t ws_bytes = tsamd_spspmm_workspace_bytes(dt, n_large, P_large, N);
  if (n_large > 0) {  // data dependent scratch: refuse politely instead of an allocator OOM
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      TORCH_CHECK((double)ws_bytes < 0.9 * (double)free_b, "spspmm: ", P_large, " intermediate products in ",
                  n_large, " rows beyond the LDS capacity need ~", (int64_t)((double)ws_bytes / 1e9),
                  " GB of scratch, more than the free device memory");
  }
  Tensor ws1 = workspace(ws_bytes, rowptrA);
  Tensor rowptrC = torch::zeros({M + 1}, iopt);  // nnzC in [0, M), scanned in place below
  // with values: the large rows are binned once, values included (no third expansion in the numeric stage)
  const int bin_values = with_value ? 1 : 0;
  check_status(tsamd_spspmm_symbolic(dt, rowptrA.data_ptr<int64_t>(), colA.data_ptr<int64_t>(),
                                     valA.has_value() ? va.data_ptr() : nullptr,
                                     rowptrB.data_ptr<int64_t>(), cb32,
                                     valB.has_value() ? vb.data_ptr() : nullptr, bin_values, M, N,
                                     prod.data_ptr<int64_t>(), bins.data_ptr<int64_t>(), n_medium,
                                     n_large, P_large, rowptrC.data_ptr<int64_t>(),
                                     ws1.data_ptr(), (size_t)ws1.numel(), stream),
               "tsamd_spspmm_symbolic");
  Tensor total = torch::empty({1}, iopt);
  Tensor ws2 = workspace(tsamd_exclusive_scan_workspace_bytes(M + 1), rowptrA);
  check_status(tsamd_exclusive_scan_i64(rowptrC.data_ptr<int64_t>(), rowptrC.data_ptr<int64_t>(),
                                        M + 1, total.data_ptr<int64_t>(), ws2.data_ptr(),
                                        (size_t)ws2.numel(), stream),
               "tsamd_exclusive_scan_i64");
  const int64_t nnz = total.item<int64_t>();  // sync 2: the output size
  Tensor colC = torch::empty({nnz}, iopt);
  Tensor valC = with_value ? torch::empty({nnz}, vopt) : torch::empty({0}, vopt);
  check_status(
      tsamd_spspmm_numeric(dt, rowptrA.data_ptr<int64_t>(), colA.data_ptr<int64_t>(),
                           valA.has_value() ? va.data_ptr() : nullptr, rowptrB.data_ptr<int64_t>(),
                           cb32, valB.has_value() ? vb.data_ptr() : nullptr, M, N,
                           prod.data_ptr<int64_t>(), bins.data_ptr<int64_t>(), n_medium, n_large,
                           P_large, rowptrC.data_ptr<int64_t>(), colC.data_ptr<int64_t>(),
                           with_value ? valC.data_ptr() : nullptr, bin_values, ws1.data_ptr(),
                           (size_t)ws1.numel(), stream),
      "tsamd_spspmm_numeric");
  return std::make_tuple(rowptrC, colC, valC);
}

// ---- sub-matrix extraction (include/tsamd.h "select" / "filter") ---------
// Pick K segments of a (ptr, ind) pattern: -> (out_ptr[K+1], seg[T], ind_out[T], pos[T]) where
// seg/ind_out are empty unless asked for.  One host sync (T is data dependent); ids outside
// [-S, S) raise IndexError like torch indexing does.
std::tuple<Tensor, Tensor, Tensor, Tensor> select_segments(Tensor ptr, Tensor ind, Tensor idx,
                                                           bool want_seg, bool want_ind) {
  check_index(ptr, "ptr");
  check_index(ind, "ind");
  check_index(idx, "idx");
  TORCH_CHECK(ptr.numel() >= 1, "select_segments: empty ptr");
  c10::hip::HIPGuard guard(ptr.get_device());
  ptr = ptr.contiguous();
  ind = ind.contiguous();
  idx = idx.contiguous();
  const int64_t S = ptr.numel() - 1, K = idx.numel();
  auto iopt = ptr.options().requires_grad(false);
  void *stream = current_stream(ptr);
  Tensor out_ptr = torch::empty({K + 1}, iopt), info = torch::empty({2}, iopt);
  Tensor ws = workspace(tsamd_select_workspace_bytes(K), ptr);
  check_status(tsamd_select_plan(ptr.data_ptr<int64_t>(), S, idx.data_ptr<int64_t>(), K,
                                 out_ptr.data_ptr<int64_t>(), info.data_ptr<int64_t>(),
                                 ws.data_ptr(), (size_t)ws.numel(), stream),
               "tsamd_select_plan");
  Tensor h = info.cpu();  // the one sync
  const int64_t total = h.data_ptr<int64_t>()[0], bad = h.data_ptr<int64_t>()[1];
  TORCH_CHECK_INDEX(bad == 0, "index out of range: ", bad, " of ", K,
                    " selected ids are outside [-", S, ", ", S, ")");
  TORCH_CHECK(total <= ind.numel() * (K > 0 ? K : 1), "select_segments: inconsistent ptr");
  Tensor seg = torch::empty({want_seg ? total : 0}, iopt);
  Tensor ind_out = torch::empty({want_ind ? total : 0}, iopt);
  Tensor pos = torch::empty({total}, iopt);
  check_status(tsamd_select_fill(ptr.data_ptr<int64_t>(), S, ind.data_ptr<int64_t>(),
                                 idx.data_ptr<int64_t>(), K, out_ptr.data_ptr<int64_t>(), total,
                                 want_seg ? seg.data_ptr<int64_t>() : nullptr,
                                 want_ind ? ind_out.data_ptr<int64_t>() : nullptr,
                                 pos.data_ptr<int64_t>(), stream),
               "tsamd_select_fill");
  return std::make_tuple(out_ptr, seg, ind_out, pos);
}

int keep_code(const std::string &p) {
  if (p == "col_range") return TSAMD_KEEP_COL_RANGE;
  if (p == "off_diag") return TSAMD_KEEP_OFF_DIAG;
  if (p == "mask") return TSAMD_KEEP_MASK;
  if (p == "mask_row") return TSAMD_KEEP_MASK_ROW;
  if (p == "mask_col") return TSAMD_KEEP_MASK_COL;
  TORCH_CHECK(false, "unknown predicate '", p, "'");
}

// Keep the entries of (row, col) that satisfy `pred` -> (row_out, col_out, src, n_mask).
// remap (mask_row / mask_col only): kept rows / columns are renumbered by their rank among the set
// mask bytes and n_mask is the number of set bytes (the new sparse size); otherwise n_mask = -1.
// One host sync.  The caller guarantees len(mask) covers every row / col id it is indexed with.
std::tuple<Tensor, Tensor, Tensor, int64_t> filter_coo(std::string pred, OptTensor row_,
                                                       OptTensor col_, OptTensor mask_, int64_t a,
                                                       int64_t b, bool remap, int64_t row_shift,
                                                       int64_t col_shift, bool want_row,
                                                       bool want_col) {
  const int code = keep_code(pred);
  TORCH_CHECK(row_.has_value() || col_.has_value() || mask_.has_value(), "filter_coo: no input");
  const Tensor &like = row_.has_value() ? row_.value() : (col_.has_value() ? col_.value() : mask_.value());
  c10::hip::HIPGuard guard(like.get_device());
  Tensor row, col, mask;
  int64_t n = -1;
  if (row_.has_value()) {
    check_index(row_.value(), "row");
    row = row_.value().contiguous();
    n = row.numel();
  }
  if (col_.has_value()) {
    check_index(col_.value(), "col");
    col = col_.value().contiguous();
    TORCH_CHECK(n < 0 || n == col.numel(), "row and col differ in length");
    n = col.numel();
  }
  if (mask_.has_value()) {
    check_gpu(mask_.value(), "mask");
    TORCH_CHECK(mask_.value().dim() == 1 && (mask_.value().scalar_type() == at::kBool ||
                                             mask_.value().scalar_type() == at::kByte),
                "mask must be a 1-D bool / uint8 tensor");
    mask = mask_.value().contiguous();
    if (code == TSAMD_KEEP_MASK) {
      TORCH_CHECK(n < 0 || n == mask.numel(), "mask and index differ in length");
      n = mask.numel();
    }
  }
  TORCH_CHECK(n >= 0, "filter_coo: nothing to filter");
  TORCH_CHECK(code < TSAMD_KEEP_MASK || mask.defined(), "predicate '", pred, "' needs a mask");
  TORCH_CHECK(!remap || code == TSAMD_KEEP_MASK_ROW || code == TSAMD_KEEP_MASK_COL,
              "remap needs a mask_row / mask_col predicate");
  TORCH_CHECK(!want_row || row.defined(), "want_row without row");
  TORCH_CHECK(!want_col || col.defined(), "want_col without col");
  auto iopt = like.options().dtype(torch::kLong).requires_grad(false);
  void *stream = current_stream(like);
  const uint8_t *mp = mask.defined() ? reinterpret_cast<const uint8_t *>(mask.data_ptr()) : nullptr;
  const int64_t *rp = row.defined() ? row.data_ptr<int64_t>() : nullptr;
  const int64_t *cp = col.defined() ? col.data_ptr<int64_t>() : nullptr;

  Tensor cnt = torch::zeros({2}, iopt), rank;
  if (remap) {
    const int64_t L = mask.numel();
    rank = torch::empty({L + 1}, iopt);
    Tensor ws0 = workspace(tsamd_filter_workspace_bytes(L), like);
    check_status(tsamd_filter_plan(TSAMD_KEEP_MASK, nullptr, nullptr, mp, nullptr, L, 0, 0,
                                   rank.data_ptr<int64_t>(), cnt.data_ptr<int64_t>() + 1,
                                   ws0.data_ptr(), (size_t)ws0.numel(), stream),
                 "tsamd_filter_plan");
  }
  Tensor ws = workspace(tsamd_filter_tiles_workspace_bytes(n), like);
  check_status(tsamd_filter_count(code, rp, cp, mp, nullptr, n, a, b, cnt.data_ptr<int64_t>(), ws.data_ptr(),
                                  (size_t)ws.numel(), stream),
               "tsamd_filter_count");
  Tensor h = cnt.cpu();  // the one sync
  const int64_t kept = h.data_ptr<int64_t>()[0];
  const int64_t n_mask = remap ? h.data_ptr<int64_t>()[1] : -1;
  Tensor row_out = torch::empty({want_row ? kept : 0}, iopt);
  Tensor col_out = torch::empty({want_col ? kept : 0}, iopt);
  Tensor src = torch::empty({kept}, iopt);
  const int64_t *map = remap ? rank.data_ptr<int64_t>() : nullptr;
  check_status(
      tsamd_filter_write(code, rp, cp, mp, nullptr, n, a, b, ws.data_ptr(),
                         code == TSAMD_KEEP_MASK_ROW ? map : nullptr,
                         code == TSAMD_KEEP_MASK_COL ? map : nullptr, row_shift, col_shift,
                         want_row ? row_out.data_ptr<int64_t>() : nullptr,
                         want_col ? col_out.data_ptr<int64_t>() : nullptr, src.data_ptr<int64_t>(),
                         stream),
      "tsamd_filter_write");
  return std::make_tuple(row_out, col_out, src, n_mask);
}

// One operand of a column-wise concatenation: writes its entries into the preallocated,
// row-interleaved (row_out, col_out, src_out); see tsamd_scatter_rows.  No sync.
void scatter_rows(Tensor row, Tensor col, Tensor delta, int64_t col_shift, int64_t src_offset,
                  Tensor row_out, Tensor col_out, Tensor src_out) {
  check_index(row, "row");
  check_index(col, "col");
  check_index(delta, "delta");
  check_index(row_out, "row_out");
  check_index(col_out, "col_out");
  check_index(src_out, "src_out");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  TORCH_CHECK(row_out.is_contiguous() && col_out.is_contiguous() && src_out.is_contiguous(),
              "outputs must be contiguous");
  TORCH_CHECK(row_out.numel() == col_out.numel() && row_out.numel() == src_out.numel() &&
                  row_out.numel() >= row.numel(),
              "outputs too small");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  delta = delta.contiguous();
  check_status(tsamd_scatter_rows(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), row.numel(),
                                  delta.data_ptr<int64_t>(), col_shift, src_offset,
                                  row_out.data_ptr<int64_t>(), col_out.data_ptr<int64_t>(),
                                  src_out.data_ptr<int64_t>(), current_stream(row)),
               "tsamd_scatter_rows");
}

// torch_sparse::non_diag_mask(Tensor row, Tensor col, int M, int N, int k) -> Tensor  (reference
// schema, csrc/diag.cpp:22-36)
Tensor non_diag_mask(Tensor row, Tensor col, int64_t M, int64_t N, int64_t k) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  Tensor mask = torch::empty({E + tsamd_num_diag(M, N, k)}, row.options().dtype(torch::kBool));
  check_status(tsamd_non_diag_mask(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N, k,
                                   reinterpret_cast<uint8_t *>(mask.data_ptr()), current_stream(row)),
               "tsamd_non_diag_mask");
  return mask;
}

// merged (row, col, src) of a sorted off-diagonal pattern and the full k-th diagonal; no sync
std::tuple<Tensor, Tensor, Tensor> insert_diag(Tensor row, Tensor col, int64_t M, int64_t N,
                                               int64_t k) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel(), T = E + tsamd_num_diag(M, N, k);
  Tensor row_out = torch::empty({T}, row.options()), col_out = torch::empty({T}, row.options());
  Tensor src = torch::empty({T}, row.options());
  check_status(tsamd_insert_diag(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N, k,
                                 row_out.data_ptr<int64_t>(), col_out.data_ptr<int64_t>(),
                                 src.data_ptr<int64_t>(), current_stream(row)),
               "tsamd_insert_diag");
  return std::make_tuple(row_out, col_out, src);
}

// set_diag in one pass: sorted (row, col), possibly with entries on the k-th diagonal ->
// (row, col, src) of the pattern with the full diagonal, old diagonal entries dropped;
// src[p] = input position, or E + j for the j-th diagonal entry.  One host sync.
std::tuple<Tensor, Tensor, Tensor> set_diag_pattern(Tensor row, Tensor col, int64_t M, int64_t N,
                                                    int64_t k) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  auto iopt = row.options().requires_grad(false);
  void *stream = current_stream(row);
  Tensor pos = torch::empty({E + 1}, iopt), cnt = torch::empty({1}, iopt);
  Tensor ws = workspace(tsamd_filter_workspace_bytes(E), row);
  check_status(tsamd_filter_plan(TSAMD_KEEP_OFF_DIAG, row.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                 nullptr, nullptr, E, k, 0, pos.data_ptr<int64_t>(),
                                 cnt.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), stream),
               "tsamd_filter_plan");
  const int64_t T = cnt.item<int64_t>() + tsamd_num_diag(M, N, k);  // the one sync
  Tensor row_out = torch::empty({T}, iopt), col_out = torch::empty({T}, iopt), src = torch::empty({T}, iopt);
  check_status(tsamd_set_diag_apply(pos.data_ptr<int64_t>(), row.data_ptr<int64_t>(),
                                    col.data_ptr<int64_t>(), E, M, N, k, row_out.data_ptr<int64_t>(),
                                    col_out.data_ptr<int64_t>(), src.data_ptr<int64_t>(), stream),
               "tsamd_set_diag_apply");
  return std::make_tuple(row_out, col_out, src);
}

// ---- mini-batch producers (include/tsamd.h) -----------------------------
// walk with the uniform floats handed in: out[n, L+1] is a pure function of the inputs
Tensor random_walk_with_rand(Tensor rowptr, Tensor col, Tensor start, Tensor rand) {
  check_index(rowptr, "rowptr");
  check_index(col, "col");
  check_index(start, "start");
  check_gpu(rand, "rand");
  TORCH_CHECK(rand.dim() == 2 && rand.size(0) == start.numel() && rand.scalar_type() == at::kFloat,
              "rand must be float32 [start.numel(), walk_length]");
  c10::hip::HIPGuard guard(rowptr.get_device());
  rowptr = rowptr.contiguous();
  col = col.contiguous();
  start = start.contiguous();
  rand = rand.contiguous();
  const int64_t n = start.numel(), L = rand.size(1);
  Tensor out = torch::empty({n, L + 1}, start.options());
  check_status(tsamd_random_walk(rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                 start.data_ptr<int64_t>(), rand.data_ptr<float>(), n, L,
                                 out.data_ptr<int64_t>(), current_stream(rowptr)),
               "tsamd_random_walk");
  return out;
}

// torch_sparse::random_walk(Tensor rowptr, Tensor col, Tensor start, int walk_length) -> Tensor
// (reference schema, csrc/rw.cpp:21-36); the floats come from torch's generator of the device.
Tensor random_walk(Tensor rowptr, Tensor col, Tensor start, int64_t walk_length) {
  check_index(start, "start");
  TORCH_CHECK(walk_length >= 0, "walk_length must be non-negative");
  Tensor rand = torch::rand({start.numel(), walk_length}, start.options().dtype(torch::kFloat));
  return random_walk_with_rand(rowptr, col, start, rand);
}

struct Relabelled {
  Tensor local, n_id;
  int64_t n_new;
};

// first-occurrence relabel of nbr against the seeds idx over M node ids (one host sync)
Relabelled relabel_impl(const Tensor &idx, const Tensor &nbr, int64_t M, bool want_local) {
  const int64_t n = idx.numel(), T = nbr.numel();
  auto iopt = idx.options().requires_grad(false);
  void *stream = current_stream(idx);
  Tensor slot = torch::empty({M}, iopt), rank = torch::empty({T + 1}, iopt);
  Tensor info = torch::empty({2}, iopt);
  Tensor ws = workspace(tsamd_relabel_workspace_bytes(T), idx);
  check_status(tsamd_relabel_plan(idx.data_ptr<int64_t>(), n, nbr.data_ptr<int64_t>(), T, M,
                                  slot.data_ptr<int64_t>(), rank.data_ptr<int64_t>(),
                                  info.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), stream),
               "tsamd_relabel_plan");
  Tensor h = info.cpu();
  const int64_t n_new = h.data_ptr<int64_t>()[0], bad = h.data_ptr<int64_t>()[1];
  TORCH_CHECK_INDEX(bad == 0, "node id out of range: ", bad, " ids are outside [0, ", M, ")");
  Relabelled r;
  r.n_new = n_new;
  r.local = torch::empty({want_local ? T : 0}, iopt);
  r.n_id = torch::empty({n + n_new}, iopt);
  check_status(tsamd_relabel_apply(idx.data_ptr<int64_t>(), n, nbr.data_ptr<int64_t>(), T, M,
                                   slot.data_ptr<int64_t>(), rank.data_ptr<int64_t>(),
                                   want_local ? r.local.data_ptr<int64_t>() : nullptr,
                                   r.n_id.data_ptr<int64_t>(), stream),
               "tsamd_relabel_apply");
  return r;
}

// torch_sparse::sample_adj(Tensor rowptr, Tensor col, Tensor idx, int num_neighbors, bool replace)
//   -> (Tensor rowptr, Tensor col, Tensor n_id, Tensor e_id)      (reference schema, csrc/sample.cpp)
// Adjacency of the seeds idx restricted to sampled neighbours, columns renumbered: seeds first, new
// nodes in first-occurrence order; every row sorted by the new column id.  Two host syncs.
std::tuple<Tensor, Tensor, Tensor, Tensor> sample_adj(Tensor rowptr, Tensor col, Tensor idx,
                                                      int64_t num_neighbors, bool replace) {
  check_index(rowptr, "rowptr");
  check_index(col, "col");
  check_index(idx, "idx");
  TORCH_CHECK(rowptr.numel() >= 1, "sample_adj: empty rowptr");
  c10::hip::HIPGuard guard(rowptr.get_device());
  rowptr = rowptr.contiguous();
  col = col.contiguous();
  idx = idx.contiguous();
  const int64_t M = rowptr.numel() - 1, n = idx.numel();
  auto iopt = rowptr.options().requires_grad(false);
  void *stream = current_stream(rowptr);

  Tensor out_ptr = torch::empty({n + 1}, iopt), info = torch::empty({2}, iopt);
  Tensor ws = workspace(tsamd_sample_workspace_bytes(n), rowptr);
  check_status(tsamd_sample_plan(rowptr.data_ptr<int64_t>(), M, idx.data_ptr<int64_t>(), n,
                                 num_neighbors, replace ? 1 : 0, out_ptr.data_ptr<int64_t>(),
                                 info.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), stream),
               "tsamd_sample_plan");
  Tensor h = info.cpu();  // sync 1
  const int64_t T = h.data_ptr<int64_t>()[0], bad = h.data_ptr<int64_t>()[1];
  TORCH_CHECK_INDEX(bad == 0, "index out of range: ", bad, " seed ids are outside [0, ", M, ")");

  Tensor e_id = torch::empty({T}, iopt), nbr = torch::empty({T}, iopt);
  if (num_neighbors < 0) {
    check_status(tsamd_select_fill(rowptr.data_ptr<int64_t>(), M, col.data_ptr<int64_t>(),
                                   idx.data_ptr<int64_t>(), n, out_ptr.data_ptr<int64_t>(), T,
                                   nullptr, nbr.data_ptr<int64_t>(), e_id.data_ptr<int64_t>(), stream),
                 "tsamd_select_fill");
  } else {
    // the seed of the draw comes from torch's CPU generator: torch.manual_seed() makes it reproducible
    const uint64_t seed = (uint64_t)torch::randint(0, std::numeric_limits<int64_t>::max(), {1},
                                                   torch::TensorOptions().dtype(torch::kLong))
                              .item<int64_t>();
    check_status(tsamd_sample_draw(rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                   idx.data_ptr<int64_t>(), n, num_neighbors, replace ? 1 : 0, seed,
                                   out_ptr.data_ptr<int64_t>(), e_id.data_ptr<int64_t>(),
                                   nbr.data_ptr<int64_t>(), stream),
                 "tsamd_sample_draw");
  }
  Relabelled r = relabel_impl(idx, nbr, M, true);  // sync 2
  // rows sorted by the new column id (sample_cpu.cpp:124-129)
  Tensor row = torch::empty({T}, iopt);
  check_status(tsamd_ptr2ind(out_ptr.data_ptr<int64_t>(), n, T, row.data_ptr<int64_t>(), stream),
               "tsamd_ptr2ind");
  const int64_t Nloc = n + r.n_new;
  Tensor col_s = torch::empty({T}, iopt), perm = torch::empty({T}, iopt);
  Tensor ws2 = workspace(tsamd_sort_coo_workspace_bytes(T), rowptr);
  check_status(tsamd_sort_coo(row.data_ptr<int64_t>(), r.local.data_ptr<int64_t>(), T, n > 0 ? n : 1,
                              Nloc > 0 ? Nloc : 1, nullptr, col_s.data_ptr<int64_t>(),
                              perm.data_ptr<int64_t>(), ws2.data_ptr(), (size_t)ws2.numel(), stream),
               "tsamd_sort_coo");
  return std::make_tuple(out_ptr, col_s, r.n_id, e_id.index_select(0, perm));
}

// torch_sparse::relabel(Tensor col, Tensor idx) -> (Tensor col, Tensor idx)  (csrc/relabel.cpp:18-29)
std::tuple<Tensor, Tensor> relabel(Tensor col, Tensor idx) {
  check_index(col, "col");
  check_index(idx, "idx");
  c10::hip::HIPGuard guard(col.get_device());
  col = col.contiguous();
  idx = idx.contiguous();
  // the reference keys a hash map, here the id space must be known: one extra sync for the max id
  int64_t M = 0;
  if (col.numel() > 0) M = std::max(M, col.max().item<int64_t>() + 1);
  if (idx.numel() > 0) M = std::max(M, idx.max().item<int64_t>() + 1);
  Relabelled r = relabel_impl(idx, col, M, true);
  return std::make_tuple(r.local, r.n_id);
}

// torch_sparse::relabel_one_hop(Tensor rowptr, Tensor col, Tensor? value, Tensor idx, bool bipartite)
//   -> (Tensor rowptr, Tensor col, Tensor? value, Tensor idx)           (csrc/relabel.cpp:32-45)
std::tuple<Tensor, Tensor, OptTensor, Tensor> relabel_one_hop(Tensor rowptr, Tensor col,
                                                              OptTensor value, Tensor idx,
                                                              bool bipartite) {
  check_index(rowptr, "rowptr");
  check_index(col, "col");
  check_index(idx, "idx");
  TORCH_CHECK(rowptr.numel() >= 1, "relabel_one_hop: empty rowptr");
  c10::hip::HIPGuard guard(rowptr.get_device());
  auto sel = select_segments(rowptr, col, idx, false, true);  // (out_ptr, -, nbr, pos), sync 1
  Tensor out_ptr = std::get<0>(sel), nbr = std::get<2>(sel), pos = std::get<3>(sel);
  const int64_t M = rowptr.numel() - 1;
  Relabelled r = relabel_impl(idx.contiguous(), nbr, M, true);  // sync 2
  OptTensor out_value = std::nullopt;
  if (value.has_value()) out_value = value.value().index_select(0, pos);
  if (!bipartite)
    out_ptr = torch::cat({out_ptr, torch::full({r.n_new}, nbr.numel(), out_ptr.options())});
  return std::make_tuple(out_ptr, r.local, out_value, r.n_id);
}

// Entries of the segments idx of (ptr, ind) whose index is itself in idx:
// -> (position of the segment in idx, position of the index in idx, position of the entry), in
// idx order and stored order inside a segment.  Two host syncs.
std::tuple<Tensor, Tensor, Tensor> induced_entries(Tensor idx, Tensor ptr, Tensor ind) {
  idx = idx.contiguous();
  const int64_t M = ptr.numel() - 1, n = idx.numel();
  auto iopt = ptr.options().requires_grad(false);
  void *stream = current_stream(ptr);
  Tensor assoc = torch::empty({M}, iopt), err = torch::empty({1}, iopt);
  check_status(tsamd_subset_assoc(idx.data_ptr<int64_t>(), n, M, assoc.data_ptr<int64_t>(),
                                  err.data_ptr<int64_t>(), stream),
               "tsamd_subset_assoc");
  auto sel = select_segments(ptr, ind, idx, true, true);  // sync 1 (raises on bad ids)
  Tensor seg = std::get<1>(sel), nbr = std::get<2>(sel), pos = std::get<3>(sel);
  const int64_t T = nbr.numel();
  Tensor cnt = torch::empty({1}, iopt);
  Tensor ws = workspace(tsamd_filter_tiles_workspace_bytes(T), ptr);
  check_status(tsamd_filter_count(TSAMD_KEEP_COL_MAPPED, nullptr, nbr.data_ptr<int64_t>(), nullptr,
                                  assoc.data_ptr<int64_t>(), T, 0, 0, cnt.data_ptr<int64_t>(), ws.data_ptr(),
                                  (size_t)ws.numel(), stream),
               "tsamd_filter_count");
  const int64_t kept = cnt.item<int64_t>();  // sync 2
  Tensor seg_out = torch::empty({kept}, iopt), map_out = torch::empty({kept}, iopt);
  Tensor src = torch::empty({kept}, iopt);
  check_status(tsamd_filter_write(TSAMD_KEEP_COL_MAPPED, seg.data_ptr<int64_t>(), nbr.data_ptr<int64_t>(),
                                  nullptr, assoc.data_ptr<int64_t>(), T, 0, 0, ws.data_ptr(), nullptr,
                                  assoc.data_ptr<int64_t>(), 0, 0, seg_out.data_ptr<int64_t>(),
                                  map_out.data_ptr<int64_t>(), src.data_ptr<int64_t>(), stream),
               "tsamd_filter_write");
  return std::make_tuple(seg_out, map_out, pos.index_select(0, src));
}

// torch_sparse::saint_subgraph(Tensor idx, Tensor rowptr, Tensor row, Tensor col)
//   -> (Tensor row, Tensor col, Tensor edge_index)                       (csrc/saint.cpp:20-33)
// Sub-graph induced by the node subset idx, nodes renumbered by their position in idx; rows in
// idx order, every row keeps its stored column order (as subgraph_cpu does).
std::tuple<Tensor, Tensor, Tensor> saint_subgraph(Tensor idx, Tensor rowptr, Tensor row, Tensor col) {
  check_index(idx, "idx");
  check_index(rowptr, "rowptr");
  check_index(col, "col");
  TORCH_CHECK(rowptr.numel() >= 1, "saint_subgraph: empty rowptr");
  c10::hip::HIPGuard guard(rowptr.get_device());
  return induced_entries(idx, rowptr, col);
}

// torch_sparse::neighbor_sample(Tensor colptr, Tensor row, Tensor input_node, int[] num_neighbors,
//                               bool replace, bool directed) -> (Tensor node, Tensor row, Tensor col, Tensor edge)
// (reference schema, csrc/neighbor_sample.cpp:18-27; CPU-only there).  Multi-hop sampling on the
// CSC view: hop l draws num_neighbors[l] in-neighbours of every node discovered in hop l-1; nodes
// are numbered in first-occurrence order across hops; an edge is (local id of the drawn source,
// local id of the frontier node, position in `row`).  directed=false returns instead every stored
// edge between the sampled nodes.  Two host syncs per hop.
std::tuple<Tensor, Tensor, Tensor, Tensor> neighbor_sample(const Tensor &colptr_, const Tensor &row_,
                                                           const Tensor &input_node,
                                                           std::vector<int64_t> num_neighbors,
                                                           bool replace, bool directed) {
  check_index(colptr_, "colptr");
  check_index(row_, "row");
  check_index(input_node, "input_node");
  TORCH_CHECK(colptr_.numel() >= 1, "neighbor_sample: empty colptr");
  c10::hip::HIPGuard guard(colptr_.get_device());
  Tensor colptr = colptr_.contiguous(), row = row_.contiguous();
  const int64_t M = colptr.numel() - 1;
  auto iopt = colptr.options().requires_grad(false);
  void *stream = current_stream(colptr);
  Tensor samples = input_node.contiguous();
  int64_t begin = 0, end = samples.numel();
  std::vector<Tensor> rows, cols, edges;
  const uint64_t seed0 = (uint64_t)torch::randint(0, std::numeric_limits<int64_t>::max(), {1},
                                                  torch::TensorOptions().dtype(torch::kLong))
                             .item<int64_t>();
  for (size_t ell = 0; ell < num_neighbors.size(); ++ell) {
    const int64_t k = num_neighbors[ell], F = end - begin;
    Tensor frontier = samples.narrow(0, begin, F);
    Tensor out_ptr = torch::empty({F + 1}, iopt), info = torch::empty({2}, iopt);
    Tensor ws = workspace(tsamd_sample_workspace_bytes(F), colptr);
    check_status(tsamd_sample_plan(colptr.data_ptr<int64_t>(), M, frontier.data_ptr<int64_t>(), F, k,
                                   replace ? 1 : 0, out_ptr.data_ptr<int64_t>(),
                                   info.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), stream),
                 "tsamd_sample_plan");
    Tensor h = info.cpu();  // sync 1
    const int64_t T = h.data_ptr<int64_t>()[0], bad = h.data_ptr<int64_t>()[1];
    TORCH_CHECK_INDEX(bad == 0, "index out of range: ", bad, " node ids are outside [0, ", M, ")");
    Tensor e = torch::empty({T}, iopt), nbr = torch::empty({T}, iopt);
    if (k < 0) {
      check_status(tsamd_select_fill(colptr.data_ptr<int64_t>(), M, row.data_ptr<int64_t>(),
                                     frontier.data_ptr<int64_t>(), F, out_ptr.data_ptr<int64_t>(), T,
                                     nullptr, nbr.data_ptr<int64_t>(), e.data_ptr<int64_t>(), stream),
                   "tsamd_select_fill");
    } else {
      check_status(tsamd_sample_draw(colptr.data_ptr<int64_t>(), row.data_ptr<int64_t>(),
                                     frontier.data_ptr<int64_t>(), F, k, replace ? 1 : 0,
                                     seed0 + 0x9E3779B97F4A7C15ull * (uint64_t)(ell + 1),
                                     out_ptr.data_ptr<int64_t>(), e.data_ptr<int64_t>(),
                                     nbr.data_ptr<int64_t>(), stream),
                   "tsamd_sample_draw");
    }
    Relabelled r = relabel_impl(samples, nbr, M, directed);  // sync 2
    if (directed) {
      Tensor seg = torch::empty({T}, iopt);
      check_status(tsamd_ptr2ind(out_ptr.data_ptr<int64_t>(), F, T, seg.data_ptr<int64_t>(), stream),
                   "tsamd_ptr2ind");
      rows.push_back(r.local);
      cols.push_back(begin > 0 ? seg + begin : seg);
      edges.push_back(e);
    }
    samples = r.n_id;
    begin = end;
    end = samples.numel();
  }
  if (!directed) {
    auto sub = induced_entries(samples, colptr, row);
    return std::make_tuple(samples, std::get<1>(sub), std::get<0>(sub), std::get<2>(sub));
  }
  Tensor none = torch::empty({0}, iopt);
  return std::make_tuple(samples, rows.empty() ? none : torch::cat(rows), cols.empty() ? none : torch::cat(cols),
                         edges.empty() ? none : torch::cat(edges));
}

}  // namespace

static auto registry = torch::RegisterOperators()
                           .op("torch_sparse::spmm_sum", &spmm_sum)
                           .op("torch_sparse::spmm_mean", &spmm_mean)
                           .op("torch_sparse::spmm_min", &spmm_min)
                           .op("torch_sparse::spmm_max", &spmm_max)
                           .op("torch_sparse::ind2ptr", &ind2ptr)
                           .op("torch_sparse::ptr2ind", &ptr2ind)
                           .op("torch_sparse::cuda_version", &cuda_version)
                           .op("tsamd::spmm_minmax", &spmm_minmax)
                           .op("tsamd::operand_cache", &operand_cache_ctl)
                           .op("tsamd::coo_order", &coo_order)
                           .op("tsamd::sort_coo", &sort_coo)
                           .op("tsamd::coo_check", &coo_check)
                           .op("tsamd::sort_coo_auto", &sort_coo_auto)
                           .op("tsamd::coalesce_index", &coalesce_index)
                           .op("tsamd::segment_reduce", &segment_reduce)
                           .op("tsamd::spspmm", &spspmm)
                           .op("tsamd::relabel_ids", &relabel_ids)
                           .op("tsamd::gather_rows", &gather_rows)
                           .op("tsamd::spmm_relabelled", &spmm_relabelled)
                           .op("tsamd::select_segments", &select_segments)
                           .op("tsamd::filter_coo", &filter_coo)
                           .op("tsamd::scatter_rows", &scatter_rows)
                           .op("torch_sparse::non_diag_mask", &non_diag_mask)
                           .op("tsamd::insert_diag", &insert_diag)
                           .op("tsamd::set_diag_pattern", &set_diag_pattern)
                           .op("torch_sparse::random_walk", &random_walk)
                           .op("tsamd::random_walk_with_rand", &random_walk_with_rand)
                           .op("torch_sparse::sample_adj", &sample_adj)
                           .op("torch_sparse::relabel", &relabel)
                           .op("torch_sparse::relabel_one_hop", &relabel_one_hop)
                           .op("torch_sparse::saint_subgraph", &saint_subgraph)
                           .op("torch_sparse::neighbor_sample", &neighbor_sample);
