// torch operator glue, part 2: the fused storage ops (sort / coalesce / segment reductions / SpSpMM) and the
// sub-matrix extraction ops.  See ops_spmm.cpp for the conventions.
#include "ops_common.h"

namespace tsamd_ops {

// (external linkage: the samplers in ops_sample.cpp call it)
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

namespace {

// ---- fused storage ops (no reference op of the same name: they replace Python/ATen
//      compositions of torch_sparse/storage.py, see include/tsamd.h) --------------------------

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

// sort_coo decided on the device from an EXISTING probe: counts[0] (device) = #descents (tsamd::coo_check)
std::tuple<Tensor, Tensor, Tensor> sort_coo_probed(Tensor row, Tensor col, int64_t M, int64_t N, Tensor counts) {
  check_index(row, "row");
  check_index(col, "col");
  check_index(counts, "counts");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  TORCH_CHECK(counts.numel() >= 1 && counts.is_contiguous(), "counts must hold the number of descents");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  Tensor perm = torch::empty({E}, row.options()), row_s = torch::empty({E}, row.options()),
         col_s = torch::empty({E}, row.options());
  Tensor ws = workspace(tsamd_sort_coo_workspace_bytes(E), row);
  check_status(tsamd_sort_coo_probed(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N,
                                     row_s.data_ptr<int64_t>(), col_s.data_ptr<int64_t>(), perm.data_ptr<int64_t>(),
                                     counts.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), current_stream(row)),
               "tsamd_sort_coo_probed");
  return std::make_tuple(row_s, col_s, perm);
}

// The sorts with the entries' values riding along: mode 0 plain | 1 device-decided (probe) | 2 device-decided from
// counts[0] (tsamd::coo_check) | 3 = check + device-decided sort in one go (counts[4]) -> (row_sorted, col_sorted, perm, counts, value[perm] or an empty tensor).
// 1-D values of 4- or 8-byte elements that need no gradient are written by the sort's last pass
// (tsamd_sort_coo_values); anything else is gathered through the permutation afterwards (differentiable).
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> sort_coo_values(Tensor row, Tensor col, int64_t M, int64_t N,
                                                                 int64_t mode, OptTensor opt_counts, OptTensor opt_value) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  TORCH_CHECK(mode >= 0 && mode <= 3, "mode must be 0 .. 3");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  Tensor perm = torch::empty({E}, row.options()), row_s = torch::empty({E}, row.options()),
         col_s = torch::empty({E}, row.options());
  Tensor counts;
  if (mode == 2) {
    TORCH_CHECK(opt_counts.has_value(), "mode 2 needs the probe's counts");
    counts = opt_counts.value();
    check_index(counts, "counts");
    TORCH_CHECK(counts.numel() >= 1 && counts.is_contiguous(), "counts must hold the number of descents");
  } else {
    counts = torch::empty({mode == 3 ? 4 : 2}, row.options());
  }
  Tensor value, value_s = torch::empty({0}, row.options());
  bool fused = false;
  if (opt_value.has_value()) {
    value = opt_value.value();
    check_gpu(value, "value");
    TORCH_CHECK(value.dim() >= 1 && value.size(0) == E, "value must have one entry per (row, col) pair");
    fused = value.dim() == 1 && (value.element_size() == 4 || value.element_size() == 8) && !needs_grad(value) && E > 0;
    if (fused) {
      value = value.contiguous();
      value_s = torch::empty_like(value);
    }
  }
  Tensor ws = workspace(tsamd_sort_coo_workspace_bytes(E), row);
  check_status(tsamd_sort_coo_values((int)mode, row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N,
                                     row_s.data_ptr<int64_t>(), col_s.data_ptr<int64_t>(), perm.data_ptr<int64_t>(),
                                     mode == 0 ? nullptr : counts.data_ptr<int64_t>(), fused ? value.data_ptr() : nullptr,
                                     fused ? value_s.data_ptr() : nullptr, fused ? (int64_t)value.element_size() : 0,
                                     ws.data_ptr(), (size_t)ws.numel(), current_stream(row)),
               "tsamd_sort_coo_values");
  if (opt_value.has_value() && !fused) value_s = value.index_select(0, perm);
  return std::make_tuple(row_s, col_s, perm, counts, value_s);
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

// tsamd_sort_coalesce: unsorted (row, col) [+ a 1-D value of 4- / 8-byte elements without a gradient] ->
// (index_u[2, E], seg_ptr[E+1], counts[3] = (#descents, #adjacent duplicates, #distinct pairs) on the device,
//  value in sorted order or an empty tensor); only the first counts[2] (+1) columns of index_u / entries of seg_ptr count.
std::tuple<Tensor, Tensor, Tensor, Tensor> sort_coalesce(Tensor row, Tensor col, int64_t M, int64_t N,
                                                       OptTensor opt_value) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  // (the two rows of ONE [2, E] tensor: without duplicates it IS the result, nothing is stacked afterwards)
  Tensor index_u = torch::empty({2, E}, row.options());
  Tensor row_u = index_u.select(0, 0), col_u = index_u.select(0, 1);
  Tensor row_t = torch::empty({E}, row.options()), col_t = torch::empty({E}, row.options());
  Tensor seg = torch::empty({E + 1}, row.options()), counts = torch::empty({3}, row.options());
  Tensor value, value_s = torch::empty({0}, row.options());
  if (opt_value.has_value()) {
    value = opt_value.value();
    check_gpu(value, "value");
    TORCH_CHECK(value.dim() == 1 && value.size(0) == E && (value.element_size() == 4 || value.element_size() == 8) &&
                    !needs_grad(value),
                "sort_coalesce: the value must be 1-D, of 4- / 8-byte elements and need no gradient");
    value = value.contiguous();
    value_s = torch::empty_like(value);
  }
  const bool with_value = opt_value.has_value() && E > 0;
  Tensor ws = workspace(tsamd_sort_coalesce_workspace_bytes(E), row);
  check_status(tsamd_sort_coalesce(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N, row_t.data_ptr<int64_t>(),
                                   col_t.data_ptr<int64_t>(), row_u.data_ptr<int64_t>(), col_u.data_ptr<int64_t>(),
                                   seg.data_ptr<int64_t>(), counts.data_ptr<int64_t>(),
                                   with_value ? value.data_ptr() : nullptr, with_value ? value_s.data_ptr() : nullptr,
                                   with_value ? (int64_t)value.element_size() : 0, ws.data_ptr(), (size_t)ws.numel(),
                                   current_stream(row)),
               "tsamd_sort_coalesce");
  return std::make_tuple(index_u, seg, counts, value_s);
}

// tsamd_sort_coalesce_reduce: the same with the reduction of the duplicates' values (float32 / int32, 1-D, no
// gradient; reduce = 0 sum, 1 mean, 2 min, 3 max; or no value: index only) fused into the bucket sort -> (index_u[2, E], seg_ptr[E+1],
// counts[4] = (#descents, #adjacent duplicates, #distinct pairs, 1 = value_u holds the reduced values) on the device,
// value in sorted order (valid when counts[3] == 0), value_u[E] (valid when counts[3] == 1)).
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> sort_coalesce_reduce(Tensor row, Tensor col, int64_t M, int64_t N,
                                                                      OptTensor opt_value, int64_t reduce) {
  check_index(row, "row");
  check_index(col, "col");
  TORCH_CHECK(row.numel() == col.numel(), "row and col differ in length");
  c10::hip::HIPGuard guard(row.get_device());
  row = row.contiguous();
  col = col.contiguous();
  const int64_t E = row.numel();
  TORCH_CHECK(reduce >= 0 && reduce <= 3, "sort_coalesce_reduce: reduce must be 0 (sum), 1 (mean), 2 (min) or 3 (max)");
  // without a value: index only (seg_ptr is then meaningful on the one-sweep route only -- nobody needs it)
  const bool with_value = opt_value.has_value();
  Tensor value, value_s = torch::empty({0}, row.options()), value_u = torch::empty({0}, row.options());
  bool is_float = true;
  if (with_value) {
    value = opt_value.value();
    check_gpu(value, "value");
    TORCH_CHECK(value.dim() == 1 && value.size(0) == E &&
                    (value.scalar_type() == at::kFloat || value.scalar_type() == at::kInt) && !needs_grad(value),
                "sort_coalesce_reduce: the value must be a 1-D float32 / int32 tensor that needs no gradient");
    is_float = value.scalar_type() == at::kFloat;
    value = value.contiguous();
    value_s = torch::empty_like(value);
    value_u = torch::empty_like(value);
  }
  Tensor index_u = torch::empty({2, E}, row.options());
  Tensor row_u = index_u.select(0, 0), col_u = index_u.select(0, 1);
  Tensor row_t = torch::empty({E}, row.options()), col_t = torch::empty({E}, row.options());
  Tensor seg = torch::empty({E + 1}, row.options()), counts = torch::empty({4}, row.options());
  if (E == 0) {
    counts.zero_();
    seg.zero_();
    return std::make_tuple(index_u, seg, counts, value_s, value_u);
  }
  Tensor ws = workspace(tsamd_sort_coalesce_workspace_bytes(E), row);
  check_status(tsamd_sort_coalesce_reduce(row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), E, M, N,
                                          row_t.data_ptr<int64_t>(), col_t.data_ptr<int64_t>(), row_u.data_ptr<int64_t>(),
                                          col_u.data_ptr<int64_t>(), seg.data_ptr<int64_t>(), counts.data_ptr<int64_t>(),
                                          is_float ? TSAMD_F32 : TSAMD_I32, (int)reduce,
                                          with_value ? value.data_ptr() : nullptr, with_value ? value_s.data_ptr() : nullptr,
                                          with_value ? value_u.data_ptr() : nullptr, ws.data_ptr(), (size_t)ws.numel(),
                                          current_stream(row)),
               "tsamd_sort_coalesce_reduce");
  return std::make_tuple(index_u, seg, counts, value_s, value_u);
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


}  // namespace

// tsamd_sort_rank_mode (include/tsamd.h): -1 query / 0, 1 force / 2 re-run the device self-test; returns the mode
int64_t sort_rank_mode(int64_t set) { return (int64_t)tsamd_sort_rank_mode((int)set); }
}  // namespace tsamd_ops

using namespace tsamd_ops;

static auto registry_storage = torch::RegisterOperators()
                           .op("tsamd::coo_order", &coo_order)
                           .op("tsamd::sort_coo", &sort_coo)
                           .op("tsamd::coo_check", &coo_check)
                           .op("tsamd::sort_coo_values", &sort_coo_values)
                           .op("tsamd::sort_coo_auto", &sort_coo_auto)
                           .op("tsamd::sort_coo_probed", &sort_coo_probed)
                           .op("tsamd::sort_rank_mode", &sort_rank_mode)
                           .op("tsamd::coalesce_index", &coalesce_index)
                           .op("tsamd::sort_coalesce", &sort_coalesce)
                           .op("tsamd::sort_coalesce_reduce", &sort_coalesce_reduce)
                           .op("tsamd::segment_reduce", &segment_reduce)
                           .op("tsamd::spspmm", &spspmm)
                           .op("tsamd::select_segments", &select_segments)
                           .op("tsamd::filter_coo", &filter_coo)
                           .op("tsamd::scatter_rows", &scatter_rows)
                           .op("torch_sparse::non_diag_mask", &non_diag_mask)
                           .op("tsamd::insert_diag", &insert_diag)
                           .op("tsamd::set_diag_pattern", &set_diag_pattern);
