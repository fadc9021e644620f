// torch operator glue, part 3: the mini-batch producers: random walks, neighbour
// sampling, relabelling, induced sub-graphs.  See ops_spmm.cpp for the conventions.
#include "ops_common.h"

#include <algorithm>
#include <map>
#include <unordered_map>

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

// Entries of the segments `seg_idx` of (ptr, ind) whose index is in `map_idx` (ids of a space of M_map nodes):
// -> (position of the segment in seg_idx, position of the index in map_idx, position of the entry), in
// seg_idx order and stored order inside a segment.  Two host syncs.  (Homogeneous graphs: seg_idx == map_idx;
// a relation of a heterogeneous graph: destination nodes select the segments, source nodes the entries.)
// first_wins: a node listed twice in map_idx keeps its FIRST position (the map insert of the neighbour samplers,
// neighbor_sample_cpu.cpp:31, 195) instead of its last (the assignment of subgraph_cpu, saint_cpu.cpp:17-18)
std::tuple<Tensor, Tensor, Tensor> induced_entries_bipartite(Tensor seg_idx, Tensor map_idx, int64_t M_map, Tensor ptr,
                                                             Tensor ind, bool first_wins = false) {
  seg_idx = seg_idx.contiguous();
  map_idx = map_idx.contiguous();
  const int64_t n = map_idx.numel();
  auto iopt = ptr.options().requires_grad(false);
  void *stream = current_stream(ptr);
  Tensor assoc = torch::empty({M_map}, iopt), err = torch::empty({1}, iopt);
  if (first_wins && n > 1) {  // last position in the reversed list = first position in the list
    Tensor rev = map_idx.flip(0).contiguous();
    check_status(tsamd_subset_assoc(rev.data_ptr<int64_t>(), n, M_map, assoc.data_ptr<int64_t>(),
                                    err.data_ptr<int64_t>(), stream),
                 "tsamd_subset_assoc");
    assoc = torch::where(assoc >= 0, (n - 1) - assoc, assoc);
  } else
  check_status(tsamd_subset_assoc(map_idx.data_ptr<int64_t>(), n, M_map, assoc.data_ptr<int64_t>(),
                                  err.data_ptr<int64_t>(), stream),
               "tsamd_subset_assoc");
  auto sel = select_segments(ptr, ind, seg_idx, true, true);  // sync 1 (raises on bad ids)
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

std::tuple<Tensor, Tensor, Tensor> induced_entries(Tensor idx, Tensor ptr, Tensor ind, bool first_wins = false) {
  return induced_entries_bipartite(idx, idx, ptr.numel() - 1, ptr, ind, first_wins);
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

// ---- heterogeneous multi-hop sampling (csrc/cpu/neighbor_sample_cpu.cpp:135-507) --------------------------------
using node_t = std::string;
using rel_t = std::string;
using edge_t = std::tuple<std::string, std::string, std::string>;
using TensorDict = c10::Dict<std::string, Tensor>;

// one hop over one CSC in two steps, so that a caller can plan SEVERAL draws, read their sizes back in ONE transfer and
// only then allocate and draw (the frontier slices of a hop do not depend on one another)
struct Drawn {
  Tensor frontier, colptr, row;
  Tensor out_ptr, info;  // per frontier node: segment pointer; info = (total, #bad ids), on the device
  Tensor nbr, e;         // per draw: the neighbour id and its position in `row`
  int64_t k = 0, F = 0, T = 0;
  bool replace = false;
  uint64_t seed = 0;
};

Drawn plan_neighbors(const Tensor &colptr, const Tensor &row, const Tensor &frontier, int64_t k, bool replace, uint64_t seed) {
  const int64_t M = colptr.numel() - 1, F = frontier.numel();
  auto iopt = colptr.options().requires_grad(false);
  Drawn d;
  d.frontier = frontier;
  d.colptr = colptr;
  d.row = row;
  d.k = k;
  d.F = F;
  d.replace = replace;
  d.seed = seed;
  d.out_ptr = torch::empty({F + 1}, iopt);
  d.info = torch::empty({2}, iopt);
  Tensor ws = workspace(tsamd_sample_workspace_bytes(F), colptr);
  check_status(tsamd_sample_plan(colptr.data_ptr<int64_t>(), M, frontier.data_ptr<int64_t>(), F, k, replace ? 1 : 0,
                                 d.out_ptr.data_ptr<int64_t>(), d.info.data_ptr<int64_t>(), ws.data_ptr(),
                                 (size_t)ws.numel(), current_stream(colptr)),
               "tsamd_sample_plan");
  return d;
}

// ONE host read-back for the sizes of all planned draws
void read_plans(std::vector<Drawn> &plans) {
  if (plans.empty()) return;
  std::vector<Tensor> infos;
  for (auto &d : plans) infos.push_back(d.info);
  const Tensor host = (infos.size() == 1 ? infos[0] : torch::cat(infos)).cpu();  // host sync
  const int64_t *h = host.data_ptr<int64_t>();
  for (size_t i = 0; i < plans.size(); ++i) {
    plans[i].T = h[2 * i];
    const int64_t bad = h[2 * i + 1];
    TORCH_CHECK_INDEX(bad == 0, "index out of range: ", bad, " node ids are outside [0, ", plans[i].colptr.numel() - 1, ")");
  }
}

void draw_planned(Drawn &d) {
  const int64_t M = d.colptr.numel() - 1;
  auto iopt = d.colptr.options().requires_grad(false);
  void *stream = current_stream(d.colptr);
  d.e = torch::empty({d.T}, iopt);
  d.nbr = torch::empty({d.T}, iopt);
  if (d.T == 0) return;  // nothing to draw (e.g. a relation without entries: its `row` has no storage to pass on)
  if (d.k < 0)
    check_status(tsamd_select_fill(d.colptr.data_ptr<int64_t>(), M, d.row.data_ptr<int64_t>(), d.frontier.data_ptr<int64_t>(),
                                   d.F, d.out_ptr.data_ptr<int64_t>(), d.T, nullptr, d.nbr.data_ptr<int64_t>(),
                                   d.e.data_ptr<int64_t>(), stream),
                 "tsamd_select_fill");
  else
    check_status(tsamd_sample_draw(d.colptr.data_ptr<int64_t>(), d.row.data_ptr<int64_t>(), d.frontier.data_ptr<int64_t>(),
                                   d.F, d.k, d.replace ? 1 : 0, d.seed, d.out_ptr.data_ptr<int64_t>(),
                                   d.e.data_ptr<int64_t>(), d.nbr.data_ptr<int64_t>(), stream),
                 "tsamd_sample_draw");
}

Tensor segment_ids(const Tensor &out_ptr, int64_t F, int64_t T) {
  Tensor seg = torch::empty({T}, out_ptr.options());
  check_status(tsamd_ptr2ind(out_ptr.data_ptr<int64_t>(), F, T, seg.data_ptr<int64_t>(), current_stream(out_ptr)),
               "tsamd_ptr2ind");
  return seg;
}

// The node list of one node type while a multi-hop sampler runs: ids in a capacity buffer, the length in a DEVICE
// counter, and the dense slot[] array of the relabel alive for the whole call (tsamd_relabel_seed / _extend): a hop
// appends without any host read-back; the host learns the lengths once per hop (read_counts), for all types together.
struct NodeList {
  Tensor buf, slot, state;  // state (device) = (length of the list, #bad ids + appends beyond the capacity)
  int64_t n = 0, M = 0;     // n: the length as of the last read-back
  bool seeded = false;

  void init(const Tensor &seeds, int64_t num_nodes) {
    buf = seeds.contiguous();
    n = buf.numel();
    M = num_nodes;
  }
  // the first use as a source type: slot[] is filled and the seeds are entered (legal while count == n)
  void seed() {
    if (seeded) return;
    auto iopt = buf.options().requires_grad(false);
    slot = torch::empty({M}, iopt);
    state = torch::empty({2}, iopt);
    check_status(tsamd_relabel_seed(buf.data_ptr<int64_t>(), n, M, slot.data_ptr<int64_t>(), state.data_ptr<int64_t>(),
                                    state.data_ptr<int64_t>() + 1, /*first_wins=*/1, current_stream(buf)),
                 "tsamd_relabel_seed");
    seeded = true;
  }
  // room for `extra` more ids (called between hops, while count == n)
  void reserve(int64_t extra) {
    if (buf.numel() >= n + extra) return;
    Tensor grown = torch::empty({n + extra}, buf.options());
    if (n > 0) grown.narrow(0, 0, n).copy_(buf.narrow(0, 0, n));
    buf = grown;
  }
  // numbers the ids of nbr (first-occurrence order behind everything listed so far) -> their local ids (or nothing)
  Tensor extend(const Tensor &nbr, bool want_local) {
    const int64_t T = nbr.numel();
    auto iopt = buf.options().requires_grad(false);
    Tensor local = torch::empty({want_local ? T : 0}, iopt);
    if (T == 0) return local;
    Tensor rank = torch::empty({T + 1}, iopt);
    Tensor ws = workspace(tsamd_relabel_workspace_bytes(T), buf);
    check_status(tsamd_relabel_extend(nbr.data_ptr<int64_t>(), T, M, slot.data_ptr<int64_t>(), rank.data_ptr<int64_t>(),
                                      state.data_ptr<int64_t>(), want_local ? local.data_ptr<int64_t>() : nullptr,
                                      buf.data_ptr<int64_t>(), buf.numel(), state.data_ptr<int64_t>() + 1, ws.data_ptr(),
                                      (size_t)ws.numel(), current_stream(buf)),
                 "tsamd_relabel_extend");
    return local;
  }
  // (a view while the buffer is at most twice the list: the copy would cost more than the slack)
  Tensor nodes() const { return buf.numel() == n ? buf : (buf.numel() <= 2 * n ? buf.narrow(0, 0, n) : buf.narrow(0, 0, n).clone()); }
};

// ONE host read-back for the lengths (and error counts) of all lists that were extended
void read_counts(std::vector<NodeList *> lists) {
  std::vector<Tensor> words;
  std::vector<NodeList *> live;
  for (NodeList *l : lists)
    if (l->seeded) {
      words.push_back(l->state);
      live.push_back(l);
    }
  if (live.empty()) return;
  const Tensor host = (words.size() == 1 ? words[0] : torch::cat(words)).cpu();  // host sync
  const int64_t *h = host.data_ptr<int64_t>();
  for (size_t i = 0; i < live.size(); ++i) {
    TORCH_CHECK_INDEX(h[2 * i + 1] == 0, "node id out of range: ", h[2 * i + 1], " ids are outside [0, ", live[i]->M, ")");
    live[i]->n = h[2 * i];
  }
}

uint64_t host_seed() {  // from torch's CPU generator: torch.manual_seed() makes the draws reproducible
  return (uint64_t)torch::randint(0, std::numeric_limits<int64_t>::max(), {1}, torch::TensorOptions().dtype(torch::kLong))
      .item<int64_t>();
}

// torch_sparse::neighbor_sample(Tensor colptr, Tensor row, Tensor input_node, int[] num_neighbors,
//                               bool replace, bool directed) -> (Tensor node, Tensor row, Tensor col, Tensor edge)
// (reference schema, csrc/neighbor_sample.cpp:18-27; CPU-only there).  Multi-hop sampling on the
// CSC view: hop l draws num_neighbors[l] in-neighbours of every node discovered in hop l-1; nodes
// are numbered in first-occurrence order across hops; an edge is (local id of the drawn source,
// local id of the frontier node, position in `row`).  directed=false returns instead every stored
// edge between the sampled nodes.  Two host read-backs per hop (size of the draw; length of the node list); the
// relabel's slot array is set up once per call (round 6; before: an M x 8-byte fill and a re-seeding of every node
// sampled so far in every hop).
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
  NodeList list;
  list.init(input_node, M);
  int64_t begin = 0, end = list.n;
  std::vector<Tensor> rows, cols, edges;
  const uint64_t seed0 = host_seed();
  for (size_t ell = 0; ell < num_neighbors.size(); ++ell) {
    const int64_t k = num_neighbors[ell], F = end - begin;
    std::vector<Drawn> plan;
    plan.push_back(plan_neighbors(colptr, row, list.buf.narrow(0, begin, F), k, replace,
                                  seed0 + 0x9E3779B97F4A7C15ull * (uint64_t)(ell + 1)));
    read_plans(plan);  // sync 1
    Drawn &d = plan[0];
    list.seed();
    list.reserve(d.T);
    draw_planned(d);
    Tensor local = list.extend(d.nbr, directed);
    if (directed) {
      Tensor seg = segment_ids(d.out_ptr, F, d.T);
      rows.push_back(local);
      cols.push_back(begin > 0 ? seg + begin : seg);
      edges.push_back(d.e);
    }
    read_counts({&list});  // sync 2
    begin = end;
    end = list.n;
  }
  Tensor samples = list.nodes();
  if (!directed) {
    auto sub = induced_entries(samples, colptr, row, /*first_wins=*/true);
    return std::make_tuple(samples, std::get<1>(sub), std::get<0>(sub), std::get<2>(sub));
  }
  Tensor none = torch::empty({0}, iopt);
  return std::make_tuple(samples, rows.empty() ? none : torch::cat(rows), cols.empty() ? none : torch::cat(cols),
                         edges.empty() ? none : torch::cat(edges));
}

struct HeteroSetup {
  std::map<rel_t, edge_t> to_edge_type;
  std::map<node_t, int64_t> num_nodes;  // id space of every node type (the dense relabel needs it)
  std::vector<rel_t> rels_sorted;       // the keys of num_neighbors_dict in ascending order (the reference's hop order)
  Tensor any;                           // some device tensor (options / device of the outputs)
};

// largest id of an index array, remembered per (data pointer, length, version counter): see hetero_setup
struct MaxKey {
  const void *data;
  int64_t numel;
  uint64_t version;
  bool operator==(const MaxKey &o) const { return data == o.data && numel == o.numel && version == o.version; }
};
struct MaxKeyHash {
  size_t operator()(const MaxKey &k) const {
    return std::hash<const void *>()(k.data) ^ (std::hash<int64_t>()(k.numel) * 1000003u) ^ (std::hash<uint64_t>()(k.version) * 7919u);
  }
};
// (an entry also holds the tensor's storage WEAKLY: while that storage is alive its address cannot be handed to another
// tensor, so pointer + length + version identify the contents; once it has expired the entry is dead)
struct MaxEntry {
  int64_t value;
  c10::weak_intrusive_ptr<c10::StorageImpl> storage;
};
std::mutex g_max_cache_mutex;
std::unordered_map<MaxKey, MaxEntry, MaxKeyHash> g_max_cache;

bool max_cache_lookup(const MaxKey &k, int64_t *out) {
  std::lock_guard<std::mutex> lock(g_max_cache_mutex);
  auto it = g_max_cache.find(k);
  if (it == g_max_cache.end()) return false;
  if (it->second.storage.expired()) {
    g_max_cache.erase(it);
    return false;
  }
  *out = it->second.value;
  return true;
}
void max_cache_store(const MaxKey &k, int64_t v, const Tensor &t) {
  std::lock_guard<std::mutex> lock(g_max_cache_mutex);
  if (g_max_cache.size() > 1024) g_max_cache.clear();
  g_max_cache.erase(k);
  g_max_cache.emplace(k, MaxEntry{v, c10::weak_intrusive_ptr<c10::StorageImpl>(t.storage().getWeakStorageImpl())});
}

HeteroSetup hetero_setup(const std::vector<node_t> &node_types, const std::vector<edge_t> &edge_types,
                         const TensorDict &colptr_dict, const TensorDict &row_dict, const TensorDict &input_node_dict,
                         const c10::Dict<rel_t, std::vector<int64_t>> &num_neighbors_dict) {
  HeteroSetup hs;
  for (const auto &k : edge_types) hs.to_edge_type[std::get<0>(k) + "__" + std::get<1>(k) + "__" + std::get<2>(k)] = k;
  for (const auto &t : node_types) hs.num_nodes[t] = 0;
  bool have = false;
  for (const auto &kv : colptr_dict) {
    const Tensor &cp = kv.value();
    check_index(cp, "colptr");
    TORCH_CHECK(cp.numel() >= 1, "hetero_neighbor_sample: empty colptr");
    TORCH_CHECK(hs.to_edge_type.count(kv.key()), "unknown relation ", kv.key());
    if (!have) {
      hs.any = cp;
      have = true;
    }
    auto &n = hs.num_nodes[std::get<2>(hs.to_edge_type.at(kv.key()))];  // destination type: one column per node
    n = std::max<int64_t>(n, cp.numel() - 1);
  }
  TORCH_CHECK(have, "hetero_neighbor_sample: no relations");
  // source types: the largest id any relation or input refers to.  The relations' row arrays are the graph -- the same
  // tensors call after call: their maxima are remembered per (data pointer, length, version counter); what is left
  // (first call: every array; later: the input nodes of the mini-batch) is reduced on the device and read back in ONE
  // transfer (round 5 paid one reduction over the relation's whole edge list + one host sync per dict entry, per call).
  std::vector<Tensor> pending;              // 0-dim device maxima still to be read
  std::vector<int64_t *> pending_dst;       // where each goes (num_nodes entry), as max(n, value + 1)
  std::vector<MaxKey> pending_key;          // cache slot to fill (data == nullptr: not cached)
  std::vector<Tensor> pending_src;
  for (const auto &kv : row_dict) {
    const Tensor &r = kv.value();
    check_index(r, "row");
    auto &n = hs.num_nodes[std::get<0>(hs.to_edge_type.at(kv.key()))];
    if (r.numel() == 0) continue;
    const MaxKey key{r.data_ptr(), r.numel(), (uint64_t)r._version()};
    int64_t cached = 0;
    if (max_cache_lookup(key, &cached)) {
      n = std::max<int64_t>(n, cached + 1);
    } else {
      pending.push_back(r.max());
      pending_dst.push_back(&n);
      pending_key.push_back(key);
      pending_src.push_back(r);
    }
  }
  for (const auto &kv : input_node_dict) {
    const Tensor &x = kv.value();
    check_index(x, "input_node");
    TORCH_CHECK(hs.num_nodes.count(kv.key()), "unknown node type ", kv.key());
    if (x.numel() == 0) continue;
    pending.push_back(x.max());
    pending_dst.push_back(&hs.num_nodes[kv.key()]);
    pending_key.push_back(MaxKey{nullptr, 0, 0});
    pending_src.push_back(x);
  }
  if (!pending.empty()) {
    const Tensor host = torch::stack(pending).cpu();  // the one read-back of the set-up
    const int64_t *h = host.data_ptr<int64_t>();
    for (size_t i = 0; i < pending.size(); ++i) {
      *pending_dst[i] = std::max<int64_t>(*pending_dst[i], h[i] + 1);
      if (pending_key[i].data != nullptr) max_cache_store(pending_key[i], h[i], pending_src[i]);
    }
  }
  for (const auto &kv : num_neighbors_dict) hs.rels_sorted.push_back(kv.key());
  std::sort(hs.rels_sorted.begin(), hs.rels_sorted.end());
  return hs;
}

std::tuple<TensorDict, TensorDict, TensorDict, TensorDict> pack_hetero(
    const std::vector<node_t> &node_types, const TensorDict &colptr_dict, std::map<node_t, Tensor> &samples,
    std::map<rel_t, std::vector<Tensor>> &rows, std::map<rel_t, std::vector<Tensor>> &cols,
    std::map<rel_t, std::vector<Tensor>> &edges, const Tensor &any) {
  auto iopt = any.options().requires_grad(false);
  Tensor none = torch::empty({0}, iopt);
  auto join = [&](std::vector<Tensor> &v) { return v.empty() ? none : (v.size() == 1 ? v[0] : torch::cat(v)); };
  TensorDict out_node, out_row, out_col, out_edge;
  for (const auto &t : node_types) out_node.insert(t, samples.count(t) ? samples[t] : none);
  for (const auto &kv : colptr_dict) {
    out_row.insert(kv.key(), join(rows[kv.key()]));
    out_col.insert(kv.key(), join(cols[kv.key()]));
    out_edge.insert(kv.key(), join(edges[kv.key()]));
  }
  return std::make_tuple(out_node, out_row, out_col, out_edge);
}

// torch_sparse::hetero_neighbor_sample(str[] node_types, (str, str, str)[] edge_types, Dict(str, Tensor) colptr_dict,
//     Dict(str, Tensor) row_dict, Dict(str, Tensor) input_node_dict, Dict(str, int[]) num_neighbors_dict, int num_hops,
//     bool replace, bool directed) -> (Dict(str, Tensor) node, Dict(str, Tensor) row, Dict(str, Tensor) col, Dict(str, Tensor) edge)
// (reference schema, csrc/neighbor_sample.cpp:29-45; CPU-only there).  Per hop the relations are visited in ascending
// key order; relation src__rel__dst draws in-neighbours (of type src) of the dst nodes discovered in the PREVIOUS hop
// (the slices move at the end of a hop, neighbor_sample_cpu.cpp:216-219, 352-361) and appends the new ones to src's
// node list in first-occurrence order.  Everything that is not random is identical to the reference.
std::tuple<TensorDict, TensorDict, TensorDict, TensorDict> hetero_neighbor_sample(
    const std::vector<node_t> &node_types, const std::vector<edge_t> &edge_types, const TensorDict &colptr_dict,
    const TensorDict &row_dict, const TensorDict &input_node_dict,
    const c10::Dict<rel_t, std::vector<int64_t>> &num_neighbors_dict, int64_t num_hops, bool replace, bool directed) {
  HeteroSetup hs = hetero_setup(node_types, edge_types, colptr_dict, row_dict, input_node_dict, num_neighbors_dict);
  c10::hip::HIPGuard guard(hs.any.get_device());
  auto iopt = hs.any.options().requires_grad(false);
  std::map<node_t, NodeList> lists;
  std::map<node_t, std::pair<int64_t, int64_t>> slice;
  std::vector<NodeList *> all_lists;
  for (const auto &t : node_types) {
    lists[t].init(input_node_dict.contains(t) ? input_node_dict.at(t) : torch::empty({0}, iopt), hs.num_nodes.at(t));
    slice[t] = {0, lists[t].n};
  }
  for (auto &kv : lists) all_lists.push_back(&kv.second);
  std::map<rel_t, std::vector<Tensor>> rows, cols, edges;
  const uint64_t seed0 = host_seed();
  uint64_t draw_no = 0;
  // Per hop: every relation's draw is PLANNED first (the frontier slices are fixed for the hop), ONE transfer reads all
  // their sizes; the draws and relabels of the hop then run back to back on the device (NodeList); a second transfer at
  // the end of the hop reads the new length of every node list.  Two host read-backs per hop whatever the number of
  // relations (round 5: two per relation and hop).
  for (int64_t ell = 0; ell < num_hops; ++ell) {
    std::vector<Drawn> plans;
    std::vector<const rel_t *> plan_rel;
    for (const auto &rel : hs.rels_sorted) {
      const edge_t &et = hs.to_edge_type.at(rel);
      const node_t &dst_t = std::get<2>(et);
      const auto &fan = num_neighbors_dict.at(rel);
      TORCH_CHECK((int64_t)fan.size() > ell, "num_neighbors_dict[", rel, "] has fewer than num_hops entries");
      const int64_t begin = slice.at(dst_t).first, F = slice.at(dst_t).second - begin;
      ++draw_no;
      if (F == 0) continue;
      plans.push_back(plan_neighbors(colptr_dict.at(rel).contiguous(), row_dict.at(rel).contiguous(),
                                     lists.at(dst_t).buf.narrow(0, begin, F), fan[ell], replace,
                                     seed0 + 0x9E3779B97F4A7C15ull * draw_no));
      plan_rel.push_back(&rel);
    }
    read_plans(plans);  // host sync 1 of the hop
    std::map<node_t, int64_t> extra;
    for (size_t i = 0; i < plans.size(); ++i) extra[std::get<0>(hs.to_edge_type.at(*plan_rel[i]))] += plans[i].T;
    for (const auto &kv : extra) {
      lists.at(kv.first).seed();
      lists.at(kv.first).reserve(kv.second);
    }
    for (size_t i = 0; i < plans.size(); ++i) {
      Drawn &d = plans[i];
      const rel_t &rel = *plan_rel[i];
      const edge_t &et = hs.to_edge_type.at(rel);
      const int64_t begin = slice.at(std::get<2>(et)).first;
      draw_planned(d);
      Tensor local = lists.at(std::get<0>(et)).extend(d.nbr, directed);
      if (directed) {
        Tensor seg = segment_ids(d.out_ptr, d.F, d.T);
        rows[rel].push_back(local);
        cols[rel].push_back(begin > 0 ? seg + begin : seg);
        edges[rel].push_back(d.e);
      }
    }
    if (!plans.empty()) read_counts(all_lists);  // host sync 2 of the hop
    for (const auto &t : node_types) slice[t] = {slice[t].second, lists[t].n};
  }
  std::map<node_t, Tensor> samples;
  for (const auto &t : node_types) samples[t] = lists[t].nodes();
  if (!directed) {  // every stored edge between the sampled nodes, relation by relation (neighbor_sample_cpu.cpp:366-397)
    for (const auto &kv : colptr_dict) {
      const edge_t &et = hs.to_edge_type.at(kv.key());
      const node_t &src_t = std::get<0>(et), &dst_t = std::get<2>(et);
      if (samples.at(dst_t).numel() == 0 || samples.at(src_t).numel() == 0) continue;
      auto sub = induced_entries_bipartite(samples.at(dst_t), samples.at(src_t), hs.num_nodes.at(src_t),
                                           kv.value().contiguous(), row_dict.at(kv.key()).contiguous(), /*first_wins=*/true);
      rows[kv.key()].push_back(std::get<1>(sub));
      cols[kv.key()].push_back(std::get<0>(sub));
      edges[kv.key()].push_back(std::get<2>(sub));
    }
  }
  return pack_hetero(node_types, colptr_dict, samples, rows, cols, edges, hs.any);
}

// torch_sparse::hetero_temporal_neighbor_sample(..., Dict(str, Tensor) node_time_dict, int num_hops, bool replace,
//     bool directed)                               (reference schema, csrc/neighbor_sample.cpp:47-63; directed only)
// The hetero sampler with a time constraint: a neighbour v of type src may be drawn for a node of root time t only if
// node_time[src][v] <= t (types without a time tensor are unconstrained), every root keeps its own computation tree --
// nodes are (node, root) pairs -- and a drawn node inherits the root time of the node it was drawn for.
//   take-all / without replacement: the draw of the untimed sampler, then the violating draws are dropped (as in
//       neighbor_sample_cpu.cpp:240-262, 305-330: fewer than num_neighbors may remain);
//   with replacement: num_neighbors uniform draws among the neighbours that satisfy the constraint (the reference
//       redraws until one does, :268-300 -- and never returns when none does; here such a node draws nothing).
// Device-driven inside a (relation, hop) (csrc/sample.hip, tsamd_temporal_*): the constraint is a FLAG per draw, the
// (node, root) pairs are numbered by one stable sort of old pairs + candidates, and only then ONE transfer reads
// (#draws kept, #new pairs) and one kernel writes the compacted outputs.  Host read-backs: one per hop (the sizes of all
// the hop's draws) + one per relation and hop.  (Round 5: an ATen composition with four to five per relation and hop.)
std::tuple<TensorDict, TensorDict, TensorDict, TensorDict> hetero_temporal_neighbor_sample(
    const std::vector<node_t> &node_types, const std::vector<edge_t> &edge_types, const TensorDict &colptr_dict,
    const TensorDict &row_dict, const TensorDict &input_node_dict,
    const c10::Dict<rel_t, std::vector<int64_t>> &num_neighbors_dict, const TensorDict &node_time_dict, int64_t num_hops,
    bool replace, bool directed) {
  TORCH_CHECK(directed, "Temporal sampling requires 'directed' sampling");
  HeteroSetup hs = hetero_setup(node_types, edge_types, colptr_dict, row_dict, input_node_dict, num_neighbors_dict);
  c10::hip::HIPGuard guard(hs.any.get_device());
  auto iopt = hs.any.options().requires_grad(false);
  std::map<node_t, Tensor> tnode, troot, ttime;
  std::map<node_t, std::pair<int64_t, int64_t>> slice;
  int64_t R = 1;  // number of roots
  for (const auto &kv : input_node_dict) R = std::max<int64_t>(R, kv.value().numel());
  for (const auto &t : node_types) {
    if (input_node_dict.contains(t)) {
      TORCH_CHECK(node_time_dict.contains(t), "hetero_temporal_neighbor_sample: no node_time for input type ", t);
      Tensor x = input_node_dict.at(t).contiguous();
      check_index(node_time_dict.at(t), "node_time");
      tnode[t] = x;
      troot[t] = torch::arange(x.numel(), iopt);
      ttime[t] = node_time_dict.at(t).index_select(0, x);
    } else {
      tnode[t] = torch::empty({0}, iopt);
      troot[t] = torch::empty({0}, iopt);
      ttime[t] = torch::empty({0}, iopt);
    }
    slice[t] = {0, tnode[t].numel()};
  }
  std::map<rel_t, std::vector<Tensor>> rows, cols, edges;
  const uint64_t seed0 = host_seed();
  uint64_t draw_no = 0;
  for (int64_t ell = 0; ell < num_hops; ++ell) {
    std::vector<Drawn> plans;
    std::vector<const rel_t *> plan_rel;
    for (const auto &rel : hs.rels_sorted) {
      const edge_t &et = hs.to_edge_type.at(rel);
      const node_t &dst_t = std::get<2>(et);
      const auto &fan = num_neighbors_dict.at(rel);
      TORCH_CHECK((int64_t)fan.size() > ell, "num_neighbors_dict[", rel, "] has fewer than num_hops entries");
      const int64_t k = fan[ell];
      const int64_t begin = slice.at(dst_t).first, F = slice.at(dst_t).second - begin;
      ++draw_no;
      if (F == 0) continue;
      const bool redraw = replace && k >= 0;  // draws with replacement are made among the VALID neighbours: list them all
      plans.push_back(plan_neighbors(colptr_dict.at(rel).contiguous(), row_dict.at(rel).contiguous(),
                                     tnode.at(dst_t).narrow(0, begin, F).contiguous(), redraw ? -1 : k, false,
                                     seed0 + 0x9E3779B97F4A7C15ull * draw_no));
      plan_rel.push_back(&rel);
    }
    read_plans(plans);  // the hop's one read-back of draw sizes
    for (size_t i = 0; i < plans.size(); ++i) {
      Drawn &d = plans[i];
      const rel_t &rel = *plan_rel[i];
      const edge_t &et = hs.to_edge_type.at(rel);
      const node_t &src_t = std::get<0>(et), &dst_t = std::get<2>(et);
      const int64_t k = num_neighbors_dict.at(rel)[ell];
      const int64_t begin = slice.at(dst_t).first, F = d.F;
      const bool redraw = replace && k >= 0;
      if (d.T == 0) continue;
      draw_planned(d);
      void *stream = current_stream(d.colptr);
      Tensor f_root = troot.at(dst_t).narrow(0, begin, F).contiguous(), f_time = ttime.at(dst_t).narrow(0, begin, F).contiguous();
      Tensor src_time;
      if (node_time_dict.contains(src_t)) {
        src_time = node_time_dict.at(src_t).contiguous();
        check_index(src_time, "node_time");
      }
      int64_t T = d.T;
      Tensor seg = segment_ids(d.out_ptr, F, T), nbr = d.nbr, e = d.e;
      Tensor keep = torch::empty({T}, iopt);
      check_status(tsamd_temporal_mark(nbr.data_ptr<int64_t>(), seg.data_ptr<int64_t>(), T,
                                       src_time.defined() ? src_time.data_ptr<int64_t>() : nullptr,
                                       f_time.data_ptr<int64_t>(), keep.data_ptr<int64_t>(), stream),
                   "tsamd_temporal_mark");
      if (redraw) {
        if (k == 0) continue;
        const int64_t T2 = F * k;
        Tensor nbr2 = torch::empty({T2}, iopt), e2 = torch::empty({T2}, iopt), seg2 = torch::empty({T2}, iopt),
               keep2 = torch::empty({T2}, iopt);
        Tensor ws = workspace(tsamd_temporal_redraw_workspace_bytes(T), nbr);
        check_status(tsamd_temporal_redraw(d.out_ptr.data_ptr<int64_t>(), F, T, k, d.seed ^ 0xA5A5A5A55A5A5A5Aull,
                                           nbr.data_ptr<int64_t>(), e.data_ptr<int64_t>(), keep.data_ptr<int64_t>(),
                                           nbr2.data_ptr<int64_t>(), e2.data_ptr<int64_t>(), seg2.data_ptr<int64_t>(),
                                           keep2.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), stream),
                     "tsamd_temporal_redraw");
        nbr = nbr2;
        e = e2;
        seg = seg2;
        keep = keep2;
        T = T2;
      }
      const int64_t n_old = tnode.at(src_t).numel();
      Tensor local = torch::empty({T}, iopt), keep_rank = torch::empty({T + 1}, iopt), open_rank = torch::empty({T + 1}, iopt);
      Tensor info = torch::empty({2}, iopt);
      Tensor ws = workspace(tsamd_temporal_relabel_workspace_bytes(n_old, T), nbr);
      check_status(tsamd_temporal_relabel(tnode.at(src_t).data_ptr<int64_t>(), troot.at(src_t).data_ptr<int64_t>(), n_old,
                                          nbr.data_ptr<int64_t>(), seg.data_ptr<int64_t>(), f_root.data_ptr<int64_t>(),
                                          keep.data_ptr<int64_t>(), T, hs.num_nodes.at(src_t), R,
                                          local.data_ptr<int64_t>(), keep_rank.data_ptr<int64_t>(),
                                          open_rank.data_ptr<int64_t>(), info.data_ptr<int64_t>(), ws.data_ptr(),
                                          (size_t)ws.numel(), stream),
                   "tsamd_temporal_relabel");
      const Tensor host = info.cpu();  // the relation's one read-back
      const int64_t n_keep = host.data_ptr<int64_t>()[0], n_open = host.data_ptr<int64_t>()[1];
      if (n_keep == 0) continue;
      Tensor r_out = torch::empty({n_keep}, iopt), c_out = torch::empty({n_keep}, iopt), e_out = torch::empty({n_keep}, iopt);
      Tensor node_new = tnode.at(src_t), root_new = troot.at(src_t), time_new = ttime.at(src_t);
      if (n_open > 0) {
        node_new = torch::empty({n_old + n_open}, iopt);
        root_new = torch::empty({n_old + n_open}, iopt);
        time_new = torch::empty({n_old + n_open}, iopt);
        if (n_old > 0) {
          node_new.narrow(0, 0, n_old).copy_(tnode.at(src_t));
          root_new.narrow(0, 0, n_old).copy_(troot.at(src_t));
          time_new.narrow(0, 0, n_old).copy_(ttime.at(src_t));
        }
      }
      check_status(tsamd_temporal_emit(nbr.data_ptr<int64_t>(), e.data_ptr<int64_t>(), seg.data_ptr<int64_t>(),
                                       f_root.data_ptr<int64_t>(), f_time.data_ptr<int64_t>(), keep_rank.data_ptr<int64_t>(),
                                       open_rank.data_ptr<int64_t>(), local.data_ptr<int64_t>(), T, begin,
                                       r_out.data_ptr<int64_t>(), c_out.data_ptr<int64_t>(), e_out.data_ptr<int64_t>(),
                                       node_new.data_ptr<int64_t>() + n_old, root_new.data_ptr<int64_t>() + n_old,
                                       time_new.data_ptr<int64_t>() + n_old, stream),
                   "tsamd_temporal_emit");
      rows[rel].push_back(r_out);
      cols[rel].push_back(c_out);
      edges[rel].push_back(e_out);
      tnode[src_t] = node_new;
      troot[src_t] = root_new;
      ttime[src_t] = time_new;
    }
    for (const auto &t : node_types) slice[t] = {slice[t].second, tnode[t].numel()};
  }
  return pack_hetero(node_types, colptr_dict, tnode, rows, cols, edges, hs.any);
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
                           .op("torch_sparse::neighbor_sample", &neighbor_sample)
                           .op("torch_sparse::hetero_neighbor_sample", &hetero_neighbor_sample)
                           .op("torch_sparse::hetero_temporal_neighbor_sample", &hetero_temporal_neighbor_sample);
