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
#include "ops_common.h"

namespace tsamd_ops {
namespace {

// Neither cache may hold on to memory that was allocated while a HIP graph is being captured (it would come
// from the graph's private pool): a capturing stream bypasses them.
bool stream_is_capturing(void *stream) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &cs) != hipSuccess ||
         cs != hipStreamCaptureStatusNone;
}

// ---- operand cache (include/tsamd.h: tsamd_spmm_cached) ------------------------------------------------
// One entry: the relabelled copy of the last dense operand that needed one.  A call may reuse it when the
// operand is provably the same tensor contents as far as torch can tell -- same storage object (held weakly:
// while it is alive its address cannot be handed to another tensor), same data pointer, same version
// counter, same shape / dtype / device / stream, same sparse pattern (col pointer and length) and reduction
// class -- and the kernel side re-checks a sampled fingerprint on the device.
//
// OPT-IN (off by default): the reference boundary keeps no state between calls (csrc/cuda/spmm_cuda.cu:102,134
// allocate fresh outputs, nothing else survives), and neither the version counter nor a SAMPLED fingerprint can
// see a sparse write that bypasses autograd's bookkeeping (`x.data[i] = ...`, a DLPack / raw-pointer writer, a
// collective landing in a persistent buffer): with the cache on such a write returns a stale product.  A caller
// that knows its operand is only ever updated through torch (or densely) turns it on with
// torch.ops.tsamd.operand_cache(True) or TSAMD_OPERAND_CACHE=1; inference tensors (no version counter) never use it.
struct OperandCache {
  std::mutex mu;
  bool enabled = false;
  c10::weak_intrusive_ptr<c10::StorageImpl> storage{c10::weak_intrusive_ptr<c10::StorageImpl>(
      c10::make_intrusive<c10::StorageImpl>(c10::StorageImpl::use_byte_size_t(), 0, c10::DataPtr(), nullptr, false))};
  const void *ptr = nullptr, *col_ptr = nullptr;
  uint32_t version = 0, col_version = 0;
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
    if (env != nullptr) c.enabled = env[0] == '1';
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

// ---- transposed-pattern cache ------------------------------------------------------------------------------
// grad_mat = A^T * grad_out multiplies by the CSC view (colptr, row[csr2csc], value[csr2csc]).  The reference
// gathers row[csr2csc] and value[csr2csc] anew in every backward (spmm.cpp:104-106).  The row ids of the CSC
// order only depend on the PATTERN, which a training loop does not change: they are gathered once and kept
// (one entry; key = storage objects, data pointers and version counters of `row` and `csr2csc`), so that every
// later backward reads them sequentially; the stream is part of the key (a backward on another stream starts a
// new entry instead of reading arrays whose producer it is not ordered after).  TSAMD_PATTERN_CACHE=0 turns it off (then, and for a pattern seen
// for the first time in a no-reuse setting, the kernel reads (row, value) through csr2csc -- tsamd_spmm_permuted).
//
// Only the `tsamd::spmm_sum_owned` / `tsamd::spmm_mean_owned` ops consult it -- the ones SparseTensor.matmul
// calls with arrays that its SparseStorage owns (immutable by the storage's contract, exactly like the
// reference's own storage-level caches rowcount / colptr / csr2csc).  The bare reference ops
// `torch_sparse::spmm_sum / spmm_mean` keep NO state between calls.
struct PatternCache {
  std::mutex mu;
  bool enabled = true;
  c10::weak_intrusive_ptr<c10::StorageImpl> row_storage{c10::weak_intrusive_ptr<c10::StorageImpl>(
      c10::make_intrusive<c10::StorageImpl>(c10::StorageImpl::use_byte_size_t(), 0, c10::DataPtr(), nullptr, false))};
  c10::weak_intrusive_ptr<c10::StorageImpl> perm_storage{c10::weak_intrusive_ptr<c10::StorageImpl>(
      c10::make_intrusive<c10::StorageImpl>(c10::StorageImpl::use_byte_size_t(), 0, c10::DataPtr(), nullptr, false))};
  const void *row_ptr = nullptr, *perm_ptr = nullptr;
  uint32_t row_version = 0, perm_version = 0;
  int64_t E = -1;
  void *stream = nullptr;  // the gathered arrays are produced and consumed on one stream only (no cross-stream events)
  int seen = 0;  // calls with this key so far
  Tensor row_t;
  // value[csr2csc] of FIXED edge weights (a value tensor that does not require grad, e.g. GCN's normalised
  // adjacency): valid while the pattern entry is and the value tensor keeps its storage / pointer / version
  c10::weak_intrusive_ptr<c10::StorageImpl> val_storage{c10::weak_intrusive_ptr<c10::StorageImpl>(
      c10::make_intrusive<c10::StorageImpl>(c10::StorageImpl::use_byte_size_t(), 0, c10::DataPtr(), nullptr, false))};
  const void *val_ptr = nullptr;
  uint32_t val_version = 0;
  Tensor value_t;
};

PatternCache &pattern_cache_state() {
  static PatternCache c;
  static bool init = [] {
    const char *env = getenv("TSAMD_PATTERN_CACHE");
    if (env != nullptr && env[0] == '0') c.enabled = false;
    return true;
  }();
  (void)init;
  return c;
}

// torch.ops.tsamd.pattern_cache(enable) -> was enabled; drops the cached row ids
bool pattern_cache_ctl(bool enable) {
  PatternCache &c = pattern_cache_state();
  std::lock_guard<std::mutex> lock(c.mu);
  const bool was = c.enabled;
  c.enabled = enable;
  c.row_t = Tensor();
  c.value_t = Tensor();
  c.row_ptr = nullptr;
  c.val_ptr = nullptr;
  return was;
}

// row[csr2csc] from the cache, or an undefined tensor when the caller should read through csr2csc instead
// (cache off, inference tensors, or the FIRST backward with this pattern: the gather only pays off when the
// pattern comes back, so it is made on the second sighting)
Tensor cached_csc_rows(const Tensor &row, const Tensor &csr2csc) {
  PatternCache &pc = pattern_cache_state();
  if (!pc.enabled || row.is_inference() || csr2csc.is_inference() || !row.is_contiguous() || !csr2csc.is_contiguous() ||
      !row.device().is_cuda() || stream_is_capturing(current_stream(row)))
    return Tensor();
  c10::StorageImpl *rs = row.storage().unsafeGetStorageImpl(), *ps = csr2csc.storage().unsafeGetStorageImpl();
  const uint32_t rv = row.unsafeGetTensorImpl()->version_counter().current_version();
  const uint32_t pv = csr2csc.unsafeGetTensorImpl()->version_counter().current_version();
  void *stream = current_stream(row);
  std::lock_guard<std::mutex> lock(pc.mu);
  bool same = pc.row_ptr == row.data_ptr() && pc.perm_ptr == csr2csc.data_ptr() && pc.E == row.numel() &&
              pc.row_version == rv && pc.perm_version == pv && pc.stream == stream;
  if (same) {
    auto a = pc.row_storage.lock(), b = pc.perm_storage.lock();
    same = a && b && a.get() == rs && b.get() == ps;
  }
  if (!same) {
    pc.row_storage = c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(rs));
    pc.perm_storage = c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(ps));
    pc.row_ptr = row.data_ptr();
    pc.perm_ptr = csr2csc.data_ptr();
    pc.row_version = rv;
    pc.perm_version = pv;
    pc.E = row.numel();
    pc.stream = stream;
    pc.seen = 1;
    pc.row_t = Tensor();
    pc.value_t = Tensor();
    pc.val_ptr = nullptr;
    return Tensor();
  }
  ++pc.seen;
  if (!pc.row_t.defined()) pc.row_t = row.index_select(0, csr2csc);
  return pc.row_t;
}

// value[csr2csc] for the pattern that cached_csc_rows() just confirmed (call right after it returned a defined
// tensor): kept across backwards when `value` is not trainable, gathered anew otherwise
Tensor csc_values(const Tensor &value, const Tensor &csr2csc) {
  Tensor v = value.detach();
  PatternCache &pc = pattern_cache_state();
  if (needs_grad(value) || v.is_inference() || !v.is_contiguous() || stream_is_capturing(current_stream(v)))
    return v.index_select(0, csr2csc);
  c10::StorageImpl *vs = v.storage().unsafeGetStorageImpl();
  const uint32_t vv = v.unsafeGetTensorImpl()->version_counter().current_version();
  std::lock_guard<std::mutex> lock(pc.mu);
  if (!pc.enabled || pc.perm_ptr != csr2csc.data_ptr() || !pc.row_t.defined() || pc.stream != current_stream(v))
    return v.index_select(0, csr2csc);
  bool same = pc.value_t.defined() && pc.val_ptr == v.data_ptr() && pc.val_version == vv &&
              pc.value_t.scalar_type() == v.scalar_type() && pc.value_t.numel() == v.numel();
  if (same) {
    auto a = pc.val_storage.lock();
    same = a && a.get() == vs;
  }
  if (!same) {
    pc.value_t = v.index_select(0, csr2csc);
    pc.val_storage = c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(vs));
    pc.val_ptr = v.data_ptr();
    pc.val_version = vv;
  }
  return pc.value_t;
}

// Forward launch: mirrors the argument checks of spmm_cpu.cpp:12-24 / spmm_cuda.cu:96-109.
// arg32: min / max winners as int32 ids (callers that keep them for their own backward only, tsamd.h)
std::tuple<Tensor, OptTensor> spmm_fw(const Tensor &rowptr, const Tensor &col,
                                      const OptTensor &opt_value, Tensor mat,
                                      const std::string &reduce, const OptTensor &opt_perm = std::nullopt,
                                      bool arg32 = false) {
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
  arg32 = arg32 && (red == TSAMD_MIN || red == TSAMD_MAX) && E < ((int64_t)1 << 31) && !opt_perm.has_value();
  if (arg32) {
    arg_out = torch::empty(sizes, rp.options().dtype(at::kInt));
  } else if (red == TSAMD_MIN || red == TSAMD_MAX) {
    arg_out = torch::empty(sizes, rp.options());
    arg_ptr = arg_out.value().data_ptr<int64_t>();
  }
  const int dt = dtype_code(mat);
  const size_t need = tsamd_spmm_workspace_bytes(dt, red, B, M, N, K, E);
  Tensor ws = workspace(need, mat);
  if (arg32) {
    check_status(tsamd_spmm_minmax_arg32(dt, red, rp.data_ptr<int64_t>(), c.data_ptr<int64_t>(), ptr_or_null(value),
                                         mat.data_ptr(), out.data_ptr(), arg_out.value().data_ptr<int32_t>(), B, M, N,
                                         K, E, ws.data_ptr(), (size_t)ws.numel(), nullptr, 0, 0, current_stream(mat)),
                 "tsamd_spmm_minmax_arg32");
    return std::make_tuple(out, arg_out);
  }
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
                                             const Tensor &mat_in, const std::string &reduce, bool arg32 = false) {
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
  if (!eligible || cache_bytes == 0 || stream_is_capturing(current_stream(mat_in)))
    return spmm_fw(rowptr, col, opt_value, mat_in, reduce, std::nullopt, arg32);

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
  arg32 = arg32 && (red == TSAMD_MIN || red == TSAMD_MAX) && E < ((int64_t)1 << 31);
  if (arg32) {
    arg_out = torch::empty(sizes, rowptr.options().dtype(at::kInt));
  } else if (red == TSAMD_MIN || red == TSAMD_MAX) {
    arg_out = torch::empty(sizes, rowptr.options());
    arg_ptr = arg_out.value().data_ptr<int64_t>();
  }
  void *stream = current_stream(mat_in);
  c10::StorageImpl *simpl = mat_in.storage().unsafeGetStorageImpl();
  const uint32_t version = mat_in.unsafeGetTensorImpl()->version_counter().current_version();
  const uint32_t col_version = col.is_inference() ? 0u : col.unsafeGetTensorImpl()->version_counter().current_version();
  const int red_class = (red == TSAMD_MIN || red == TSAMD_MAX) ? 1 : 0;

  std::lock_guard<std::mutex> lock(oc.mu);
  bool valid = false;
  if (oc.buf.defined() && oc.ptr == mat_in.data_ptr() && oc.version == version && oc.dtype == dt &&
      oc.red_class == red_class && oc.device == mat_in.get_device() && oc.stream == stream &&
      oc.col_ptr == col.data_ptr() && oc.col_version == col_version && oc.E == E && oc.sizes == mat_in.sizes().vec() &&
      (size_t)oc.buf.numel() >= cache_bytes) {
    auto locked = oc.storage.lock();  // the storage the copy was made from is still alive and is this one
    valid = locked && locked.get() == simpl;
  }
  if (!valid) {
    // a buffer is only ever touched on the stream it was allocated under (the caching allocator's reuse is
    // stream-ordered on that stream): another stream gets a fresh one instead of racing with pending work
    if (!oc.buf.defined() || (size_t)oc.buf.numel() < cache_bytes || oc.buf.get_device() != mat_in.get_device() ||
        oc.stream != stream) {
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
    oc.col_version = col_version;
    oc.E = E;
    oc.sizes = mat_in.sizes().vec();
    ++oc.fills;
  } else {
    ++oc.hits;
  }
  Tensor ws = workspace(tsamd_spmm_cached_workspace_bytes(dt, red, B, M, N, K, E), mat_in);
  if (arg32) {
    check_status(tsamd_spmm_minmax_arg32(dt, red, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                         ptr_or_null(value), mat_in.data_ptr(), out.data_ptr(),
                                         arg_out.value().data_ptr<int32_t>(), B, M, N, K, E, ws.data_ptr(),
                                         (size_t)ws.numel(), oc.buf.data_ptr(), (size_t)oc.buf.numel(), valid ? 1 : 0,
                                         stream),
                 "tsamd_spmm_minmax_arg32");
    return std::make_tuple(out, arg_out);
  }
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


// ---- sum / mean ---------------------------------------------------------------------------
// One Function serves both: `mean` selects the divisor handling in both directions.
// Saved tensors: row, rowptr, col, value, rowcount, colptr, csr2csc, mat.
class SpmmAddFunction : public torch::autograd::Function<SpmmAddFunction> {
 public:
  static variable_list forward(AutogradContext *ctx, OptTensor opt_row, Tensor rowptr, Tensor col,
                               Tensor value, OptTensor opt_rowcount, OptTensor opt_colptr,
                               OptTensor opt_csr2csc, Tensor mat, bool has_value, bool mean, bool owned) {
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
    ctx->saved_data["owned"] = owned;
    // absent optionals are parked as `col` (any tensor will do; they are never read then)
    ctx->save_for_backward({opt_row.value_or(col), rowptr, col, value, opt_rowcount.value_or(col),
                            opt_colptr.value_or(col), opt_csr2csc.value_or(col), mat});
    return {out};
  }

  static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
    const bool has_value = ctx->saved_data["has_value"].toBool();
    const bool mean = ctx->saved_data["mean"].toBool();
    const bool owned = ctx->saved_data["owned"].toBool();
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
        Tensor row_t = owned ? cached_csc_rows(row, csr2csc) : Tensor();
        if (row_t.defined()) {
          // the pattern came back: its CSC row ids are at hand, only the values are gathered (one ATen gather)
          OptTensor w = has_value ? OptTensor(csc_values(value, csr2csc)) : std::nullopt;
          grad_mat = std::get<0>(spmm_fw(colptr, row_t, w, grad_out, "sum"));
        } else {
          OptTensor w = has_value ? OptTensor(value.detach()) : std::nullopt;
          grad_mat = std::get<0>(spmm_fw(colptr, row, w, grad_out, "sum", csr2csc));
        }
      } else {
        Tensor row_t = owned ? cached_csc_rows(row, csr2csc) : Tensor();
        if (!row_t.defined()) row_t = row.index_select(0, csr2csc);
        Tensor cnt = rowcount.index_select(0, row_t).to(mat.scalar_type()).clamp_min_(1);
        Tensor w = has_value ? value.detach().index_select(0, csr2csc).div_(cnt) : cnt.reciprocal_();
        grad_mat = std::get<0>(spmm_fw(colptr, row_t, w, grad_out, "sum"));
      }
    }
    return {Tensor(), Tensor(), Tensor(), grad_value, Tensor(), Tensor(), Tensor(), grad_mat,
            Tensor(), Tensor(), Tensor()};
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
                               OptTensor opt_csr2csc, OptTensor opt_row, bool arg32) {
    OptTensor v = has_value ? OptTensor(value) : std::nullopt;
    const bool has_csc = opt_colptr.has_value() && opt_csr2csc.has_value() && opt_row.has_value();
    // Round 6: when the winners stay inside this node (arg32) and the pull backward is going to run (CSC arrays handed
    // over, a gradient w.r.t. mat recorded), the forward leaves the pull's winner RECORDS instead of ids
    // (tsamd_spmm_minmax_records: the merge kernel writes them where the winners sit in registers) and the backward
    // starts at the masked sum: configs[2] forward + backward 2.26 -> 2.15 ms, gradients bit-identical
    // (profiles/r06_ab_fwd_winrec.md).  Only when the records (32 bytes per entry up to 128 features, 48 up to 256) are at
    // most three times the ids they replace (4 bytes per output element) -- they are what the node holds until the backward.
    Tensor records;
    bool use_records = false;
    // (grad mode is off inside a Function's forward; the front-end hands the CSC arrays over only when it was on)
    if (arg32 && has_csc && needs_grad(mat) && !operand_cache_state().enabled &&
        mat.device().is_cuda() && mat.dim() >= 2 && rowptr.dim() == 1 && col.dim() == 1 &&
        (mat.scalar_type() == at::kFloat || mat.scalar_type() == at::kHalf || mat.scalar_type() == at::kBFloat16) &&
        (!has_value || value.scalar_type() == mat.scalar_type()) && opt_row.value().numel() == col.numel() &&
        !stream_is_capturing(current_stream(mat))) {
      const int64_t M = rowptr.numel() - 1, E = col.numel(), N = mat.size(-2), K = mat.size(-1);
      const int64_t B = (N * K) > 0 ? mat.numel() / (N * K) : 1;
      const int dt = dtype_code(mat);
      const bool value_grad_ok = !(has_value && needs_grad(value)) || (K * (int64_t)mat.element_size()) % 16 == 0;
      use_records = M > 0 && E > 0 && value_grad_ok && tsamd_spmm_minmax_records_in_forward(dt, B, M, K, E) == 1 &&
                    tsamd_spmm_minmax_records_bytes(B, K, E) <= (size_t)(3 * 4 * B * M * K);
    }
    Tensor out, arg_out;
    if (use_records) {
      check_index(rowptr, "rowptr");
      check_index(col, "col");
      check_index(opt_row.value(), "row");
      c10::hip::HIPGuard guard(mat.get_device());
      Tensor m = mat.contiguous(), rp = rowptr.contiguous(), c = col.contiguous(), r = opt_row.value().contiguous();
      OptTensor vc = has_value ? OptTensor(value.contiguous()) : std::nullopt;
      const int64_t M = rp.numel() - 1, E = c.numel(), N = m.size(-2), K = m.size(-1);
      const int64_t B = (N * K) > 0 ? m.numel() / (N * K) : 1;
      const int dt = dtype_code(m), red = is_max ? TSAMD_MAX : TSAMD_MIN;
      if (has_value) TORCH_CHECK(value.dim() == 1 && value.size(0) == E, "Input mismatch");
      auto sizes = m.sizes().vec();
      sizes[m.dim() - 2] = M;
      out = torch::empty(sizes, m.options().requires_grad(false));
      records = torch::empty({(int64_t)(tsamd_spmm_minmax_records_bytes(B, K, E) / 4)}, rp.options().dtype(at::kInt));
      Tensor ws = workspace(tsamd_spmm_minmax_records_workspace_bytes(dt, red, B, M, N, K, E), m);
      check_status(tsamd_spmm_minmax_records(dt, red, rp.data_ptr<int64_t>(), c.data_ptr<int64_t>(), ptr_or_null(vc),
                                             m.data_ptr(), out.data_ptr(), r.data_ptr<int64_t>(),
                                             reinterpret_cast<uint32_t *>(records.data_ptr<int32_t>()), B, M, N, K, E,
                                             ws.data_ptr(), (size_t)ws.numel(), current_stream(m)),
                   "tsamd_spmm_minmax_records");
      arg_out = records;  // (what this mode returns in the ids' place: the caller asked not to see them)
    } else {
      auto res = spmm_fw_cached(rowptr, col, v, mat, is_max ? "max" : "min", arg32);
      out = std::get<0>(res);
      arg_out = std::get<1>(res).value();
    }
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
    ctx->saved_data["records"] = use_records;
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
      // The pull is deterministic and faster for grad_mat alone (configs[2]: 1.36 vs 2.47 ms).  With grad_value as
      // well it pays a masked SDDMM where the scatter kernel gets the value gradient fused: WHICH of the two wins
      // depends on the row size, and the front-end decides that BEFORE the forward -- it hands the CSC arrays over
      // exactly when it wants the pull (pytorch_sparse_amd/tensor.py: storage_spmm; until round 5 this line asked
      // for deterministic algorithms again and sent the with-values case down the scatter route although the
      // front-end had built the CSC arrays for it).
      const bool pull = has_csc && want_mat;
      if (ctx->saved_data["records"].toBool()) {  // the forward left the pull's winner records (see forward)
        Tensor colptr = s[5].contiguous(), csr2csc = s[6].contiguous(), row = s[7].contiguous();
        if (!want_mat) grad_mat = torch::empty_like(mat, mat.options().requires_grad(false));  // (the entry needs it)
        Tensor ws = workspace(tsamd_spmm_minmax_bw_csc_records_workspace_bytes(dt, B, M, N, K, E), mat);
        check_status(tsamd_spmm_minmax_bw_csc_records(
                         dt, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(), has_value ? 1 : 0, mat.data_ptr(),
                         grad_out.data_ptr(), reinterpret_cast<const uint32_t *>(arg_out.data_ptr<int32_t>()),
                         colptr.data_ptr<int64_t>(), csr2csc.data_ptr<int64_t>(), row.data_ptr<int64_t>(),
                         want_value ? grad_value.data_ptr() : nullptr, grad_mat.data_ptr(), B, M, N, K, E, ws.data_ptr(),
                         (size_t)ws.numel(), current_stream(mat)),
                     "tsamd_spmm_minmax_bw_csc_records");
        if (!want_mat) grad_mat = Tensor();
        return {Tensor(), Tensor(), grad_value, grad_mat, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
      }
      if (pull && arg_out.scalar_type() == at::kInt) {  // the winners were kept as 32-bit ids (forward, arg32)
        Tensor colptr = s[5].contiguous(), csr2csc = s[6].contiguous(), row = s[7].contiguous();
        Tensor ws = workspace(tsamd_spmm_minmax_bw_csc_workspace_bytes(dt, B, M, N, K, E), mat);
        st = tsamd_spmm_minmax_bw_csc_arg32(dt, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                            has_value ? value.data_ptr() : nullptr, mat.data_ptr(),
                                            grad_out.data_ptr(), arg_out.data_ptr<int32_t>(),
                                            colptr.data_ptr<int64_t>(), csr2csc.data_ptr<int64_t>(),
                                            row.data_ptr<int64_t>(), want_value ? grad_value.data_ptr() : nullptr,
                                            grad_mat.data_ptr(), B, M, N, K, E, ws.data_ptr(), (size_t)ws.numel(),
                                            current_stream(mat));
        if (st != TSAMD_ERR_UNSUPPORTED) check_status(st, "tsamd_spmm_minmax_bw_csc_arg32");
      }
      if (arg_out.scalar_type() == at::kInt && st == TSAMD_ERR_UNSUPPORTED) arg_out = arg_out.to(at::kLong);
      if (pull && st == TSAMD_ERR_UNSUPPORTED) {
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
    return {Tensor(), Tensor(), grad_value, grad_mat, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
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
                                mat, opt_value.has_value(), false, false)[0];
}

Tensor spmm_mean(OptTensor opt_row, Tensor rowptr, Tensor col, OptTensor opt_value,
                 OptTensor opt_rowcount, OptTensor opt_colptr, OptTensor opt_csr2csc, Tensor mat) {
  Tensor value = opt_value.value_or(col);
  return SpmmAddFunction::apply(opt_row, rowptr, col, value, opt_rowcount, opt_colptr, opt_csr2csc,
                                mat, opt_value.has_value(), true, false)[0];
}

// The same two ops for callers whose index arrays belong to a SparseStorage (SparseTensor.matmul): the backward
// may keep row[csr2csc] (and value[csr2csc] of non-trainable weights) across calls -- see PatternCache.
Tensor spmm_sum_owned(OptTensor opt_row, Tensor rowptr, Tensor col, OptTensor opt_value,
                      OptTensor opt_colptr, OptTensor opt_csr2csc, Tensor mat) {
  Tensor value = opt_value.value_or(col);
  return SpmmAddFunction::apply(opt_row, rowptr, col, value, std::nullopt, opt_colptr, opt_csr2csc,
                                mat, opt_value.has_value(), false, true)[0];
}

Tensor spmm_mean_owned(OptTensor opt_row, Tensor rowptr, Tensor col, OptTensor opt_value,
                       OptTensor opt_rowcount, OptTensor opt_colptr, OptTensor opt_csr2csc, Tensor mat) {
  Tensor value = opt_value.value_or(col);
  return SpmmAddFunction::apply(opt_row, rowptr, col, value, opt_rowcount, opt_colptr, opt_csr2csc,
                                mat, opt_value.has_value(), true, true)[0];
}

std::tuple<Tensor, Tensor> spmm_min(Tensor rowptr, Tensor col, OptTensor opt_value, Tensor mat) {
  auto r = SpmmMinMaxFunction::apply(rowptr, col, opt_value.value_or(col), mat,
                                     opt_value.has_value(), false, std::nullopt, std::nullopt, std::nullopt, false);
  return std::make_tuple(r[0], r[1]);
}

std::tuple<Tensor, Tensor> spmm_max(Tensor rowptr, Tensor col, OptTensor opt_value, Tensor mat) {
  auto r = SpmmMinMaxFunction::apply(rowptr, col, opt_value.value_or(col), mat,
                                     opt_value.has_value(), true, std::nullopt, std::nullopt, std::nullopt, false);
  return std::make_tuple(r[0], r[1]);
}

// tsamd::spmm_minmax(Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr, Tensor? csr2csc, Tensor? row,
//                    Tensor mat, bool is_max, bool arg32) -> (Tensor, Tensor)
// spmm_min / spmm_max with the CSC arrays of the matrix: same forward, atomic-free deterministic backward.
// arg32: the second result holds the winners as int32 ids (for callers that only keep it for this backward:
// SparseTensor.matmul returns `out` alone) -- half the bytes written by the forward and read by the backward.
std::tuple<Tensor, Tensor> spmm_minmax(Tensor rowptr, Tensor col, OptTensor opt_value, OptTensor opt_colptr,
                                       OptTensor opt_csr2csc, OptTensor opt_row, Tensor mat, bool is_max, bool arg32) {
  auto r = SpmmMinMaxFunction::apply(rowptr, col, opt_value.value_or(col), mat, opt_value.has_value(), is_max,
                                     opt_colptr, opt_csr2csc, opt_row, arg32);
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

// torch_sparse.spmm(index, value, m, n, matrix) on a small unsorted COO: ONE launch (tsamd_spmm_coo_small).  No
// autograd: the Python front-end takes this route only when nothing asks for a gradient.
bool spmm_coo_small_supported(Tensor value, int64_t E, int64_t M, int64_t K) {
  switch (value.scalar_type()) {
    case at::kFloat: case at::kDouble: case at::kHalf: case at::kBFloat16: case at::kInt: case at::kLong:
    case at::kByte: case at::kChar: case at::kShort: break;
    default: return false;
  }
  return tsamd_spmm_coo_small_supported(dtype_code(value), E, M, K) != 0;
}

// -> the product [M, K], or an EMPTY 1-D tensor when the direct route does not take this problem (too big, a batch
// of matrices, a dtype it has no kernel for): one operator call decides and computes, the caller falls through to the
// sorted route on the sentinel.
Tensor spmm_coo_small(Tensor index, Tensor value, int64_t M, int64_t N, Tensor mat) {
  if (!(index.device().is_cuda() && value.device().is_cuda() && mat.device().is_cuda()) || mat.dim() != 2 ||
      index.dim() != 2 || index.size(0) != 2 || index.scalar_type() != at::kLong || value.dim() != 1 ||
      value.size(0) != index.size(1) || value.scalar_type() != mat.scalar_type() || mat.size(0) != N ||
      !spmm_coo_small_supported(value, index.size(1), M, mat.size(1)))
    return torch::empty({0}, mat.options().requires_grad(false));
  c10::hip::HIPGuard guard(mat.get_device());
  index = index.contiguous();
  value = value.contiguous();
  mat = mat.contiguous();
  const int64_t E = index.size(1), K = mat.size(1);
  Tensor out = torch::empty({M, K}, mat.options().requires_grad(false));
  check_status(tsamd_spmm_coo_small(dtype_code(mat), index.data_ptr<int64_t>(), index.data_ptr<int64_t>() + E,
                                    value.data_ptr(), mat.data_ptr(), out.data_ptr(), E, M, N, K, current_stream(mat)),
               "tsamd_spmm_coo_small");
  return out;
}

int64_t cuda_version() { return tsamd_hip_version(); }

// tsamd_spmm_reference_order (include/tsamd.h): 1 = every SpMM forward in the reference CPU kernel's order of
// operations (bit-identical results, slow), 0 = the product kernels, anything else = query; returns the mode in force
int64_t reference_order(int64_t set) { return (int64_t)tsamd_spmm_reference_order((int)set); }

// torch.are_deterministic_algorithms_enabled() for TorchScript callers (storage_spmm)
bool deterministic_mode() { return at::globalContext().deterministicAlgorithms(); }


}  // namespace
}  // namespace tsamd_ops

using namespace tsamd_ops;

static auto registry_spmm = torch::RegisterOperators()
                           .op("torch_sparse::spmm_sum", &spmm_sum)
                           .op("torch_sparse::spmm_mean", &spmm_mean)
                           .op("torch_sparse::spmm_min", &spmm_min)
                           .op("torch_sparse::spmm_max", &spmm_max)
                           .op("torch_sparse::ind2ptr", &ind2ptr)
                           .op("torch_sparse::ptr2ind", &ptr2ind)
                           .op("torch_sparse::cuda_version", &cuda_version)
                           .op("tsamd::spmm_coo_small", &spmm_coo_small)
                           .op("tsamd::spmm_coo_small_supported", &spmm_coo_small_supported)
                           .op("tsamd::spmm_minmax", &spmm_minmax)
                           .op("tsamd::deterministic", &deterministic_mode)
                           .op("tsamd::reference_order", &reference_order)
                           .op("tsamd::spmm_sum_owned", &spmm_sum_owned)
                           .op("tsamd::spmm_mean_owned", &spmm_mean_owned)
                           .op("tsamd::operand_cache", &operand_cache_ctl)
                           .op("tsamd::pattern_cache", &pattern_cache_ctl)
                           .op("tsamd::relabel_ids", &relabel_ids)
                           .op("tsamd::gather_rows", &gather_rows)
                           .op("tsamd::spmm_relabelled", &spmm_relabelled);
