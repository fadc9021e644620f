// torch operator glue, part 3: the mini-batch producers: random walks, neighbour
// sampling, relabelling, induced sub-graphs.  See ops_spmm.cpp for the conventions.
#include "ops_common.h"

namespace tsamd_ops {
namespace {

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
}  // namespace tsamd_ops

using namespace tsamd_ops;

static auto registry_sample = torch::RegisterOperators()
                           .op("torch_sparse::random_walk", &random_walk)
                           .op("tsamd::random_walk_with_rand", &random_walk_with_rand)
                           .op("torch_sparse::sample_adj", &sample_adj)
                           .op("torch_sparse::relabel", &relabel)
                           .op("torch_sparse::relabel_one_hop", &relabel_one_hop)
                           .op("torch_sparse::saint_subgraph", &saint_subgraph)
                           .op("torch_sparse::neighbor_sample", &neighbor_sample);
