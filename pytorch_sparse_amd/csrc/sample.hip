// Mini-batch producers on gfx950: uniform random walks, one-hop neighbour sampling and the
// first-occurrence relabelling every sampler of the reference ends with.
// The reference has these on the CPU only (a CUDA kernel exists just for the random walk):
//   * random_walk        csrc/cpu/rw_cpu.cpp:5-45, csrc/cuda/rw_cuda.cu:10-53
//   * sample_adj         csrc/cpu/sample_cpu.cpp:10-140  (std::unordered_map relabel, Floyd sampling,
//                        one torch::randint call per draw)
//   * relabel(_one_hop)  csrc/cpu/relabel_cpu.cpp:5-155
// Here:
//   * walks: one lane per walk, `rand` handed in (so the result is a pure function of its inputs);
//   * draws: one lane per (row, j).  Without replacement the j-th draw of a row with more than 64
//     neighbours is pi_row(j) for a keyed pseudo-random BIJECTION pi_row of [0, deg) (6-round
//     Feistel network with Philox4x32-10 round keys of (seed, row) + cycle walking): distinct by
//     construction, O(1) state, no per-row set, hubs cost the same as anything else.  Rows with at
//     most 64 neighbours take an exactly uniform k-subset by Floyd's algorithm with the set in a
//     64-bit mask (one lane per row).  With replacement: a Philox draw per (row, j);
//   * relabel: a dense slot[] array over the node ids (8 B per node of the graph): seeds hold
//     -(i+1), every other drawn node the minimum draw position (atomicMin) = its FIRST OCCURRENCE in
//     row-major order; flags + device scan rank the first occurrences -> new ids n, n+1, ... in the
//     order the reference's sequential std::unordered_map walk assigns them;
//   * multi-hop samplers (neighbor_sample, hetero_neighbor_sample): the slot[] array of a node type
//     LIVES for the whole call (tsamd_relabel_seed once, tsamd_relabel_extend per hop / relation):
//     a node that has been numbered holds -(id + 1) from then on, the node list grows inside a
//     capacity buffer and its length is a DEVICE counter -- a relation's relabel costs neither the
//     M x 8-byte fill nor the re-seeding of everything sampled so far, and no host read-back.
#include "common.h"
#include "scan.h"

namespace tsamd {
namespace {

// ---- random walk ------------------------------------------------------------------------------
__global__ void random_walk_kernel(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
                                   const int64_t *__restrict__ start, const float *__restrict__ rand,
                                   int64_t n, int64_t L, int64_t *__restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  int64_t cur = start[w];
  out[w * (L + 1)] = cur;
  for (int64_t l = 0; l < L; ++l) {
    const int64_t s = rowptr[cur], e = rowptr[cur + 1];
    // reference: col[s + int64(rand * deg)]; a node without neighbours keeps the walk in place
    // (the reference reads the next row's first entry there)
    if (e > s) {
      int64_t p = (int64_t)(rand[w * L + l] * (float)(e - s));
      if (p >= e - s) p = e - s - 1;  // float rounding of a rand just below 1 on a huge row
      cur = col[s + p];
    }
    out[w * (L + 1) + l + 1] = cur;
  }
}

// ---- Philox4x32-10 (Salmon et al., SC'11) --------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

__device__ inline U4 philox(uint64_t seed, uint64_t c_lo, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  U4 c = {(uint32_t)c_lo, (uint32_t)(c_lo >> 32), c2, c3};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

__device__ inline uint64_t u64(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

__device__ inline uint32_t fmix32(uint32_t h) {  // MurmurHash3 finaliser
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// j -> pi(j): keyed pseudo-random bijection of [0, deg), used for deg > 64: a balanced 6-round
// Feistel network on 2 * ceil(b / 2) bits (b = bits of deg - 1) whose round function is a strong
// 32-bit mixer of the right half and a Philox-derived round key, cycle-walked back into range
// (the domain is < 4 deg: < 4 rounds expected).  (Mitchell et al., "Bandwidth-optimal random
// shuffling for GPUs", use the same construction.)
__device__ inline uint64_t permute_index(uint64_t j, uint64_t deg, uint64_t seed, uint64_t row) {
  const int b = 64 - __clzll((long long)(deg - 1));
  const int h = (b + 1) >> 1;  // half width, 4..32 here
  const uint64_t hmask = (1ull << h) - 1;
  const U4 a = philox(seed, row, 0u, 0x5A17u), c = philox(seed, row, 1u, 0x5A17u);
  const uint32_t key[6] = {a.x, a.y, a.z, a.w, c.x, c.y};
  uint64_t x = j;
  do {
    uint64_t L = x >> h, R = x & hmask;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const uint32_t f = fmix32((uint32_t)R * 0x9E3779B1u + key[r]);  // R has at most 32 bits
      const uint64_t nR = (L ^ (uint64_t)(f >> (32 - h))) & hmask;
      L = R;
      R = nR;
    }
    x = (L << h) | R;
  } while (x >= deg);
  return x;
}

__device__ inline int64_t wrap_id(int64_t j, int64_t S) { return j < 0 ? j + S : j; }

// cnt[i] = number of neighbours row idx[i] contributes
__global__ void sample_count_kernel(const int64_t *__restrict__ rowptr, int64_t M,
                                    const int64_t *__restrict__ idx, int64_t n, int64_t k,
                                    int replace, int64_t *__restrict__ cnt,
                                    unsigned long long *err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  if (v < 0 || v >= M) {
    atomicAdd(err, 1ull);
    cnt[i] = 0;
    return;
  }
  const int64_t deg = rowptr[v + 1] - rowptr[v];
  cnt[i] = k < 0 ? deg : (replace ? (deg > 0 ? k : 0) : (deg < k ? deg : k));
}

// one lane per (row i, draw j), j < k
__global__ void sample_draw_kernel(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
                                   const int64_t *__restrict__ idx, int64_t n, int64_t k, int replace,
                                   uint64_t seed, const int64_t *__restrict__ out_ptr,
                                   int64_t *__restrict__ e_id, int64_t *__restrict__ nbr) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * k) return;
  const int64_t i = t / k, j = t - i * k;
  const int64_t o = out_ptr[i];
  if (j >= out_ptr[i + 1] - o) return;
  const int64_t v = idx[i];
  const int64_t s = rowptr[v], deg = rowptr[v + 1] - s;
  int64_t p;
  if (replace) {
    const U4 r = philox(seed, (uint64_t)i, (uint32_t)j, 0xD4A3u ^ (uint32_t)((uint64_t)j >> 32));
    p = (int64_t)__umul64hi(u64(r.x, r.y), (uint64_t)deg);
  } else if (deg <= k) {
    p = j;  // the whole row, in stored order
  } else if (deg <= 64) {
    // small rows: an exactly uniform k-subset by Floyd's algorithm, the set kept in a 64-bit mask;
    // the lane of draw 0 does the whole row (k <= 64 steps)
    if (j != 0) return;
    uint64_t used = 0;
    int64_t cnt = 0;
    for (int64_t t = deg - k; t < deg; ++t) {
      const U4 r = philox(seed, (uint64_t)i, (uint32_t)t, 0xF10Du);
      const uint64_t pick0 = __umul64hi(u64(r.x, r.y), (uint64_t)(t + 1));  // uniform in [0, t]
      const uint64_t pick = ((used >> pick0) & 1ull) ? (uint64_t)t : pick0;
      used |= 1ull << pick;
      e_id[o + cnt] = s + (int64_t)pick;
      nbr[o + cnt] = col[s + (int64_t)pick];
      ++cnt;
    }
    return;
  } else {
    p = (int64_t)permute_index((uint64_t)j, (uint64_t)deg, seed, (uint64_t)i);
  }
  e_id[o + j] = s + p;
  nbr[o + j] = col[s + p];
}

// ---- first-occurrence relabel -------------------------------------------------------------------
__global__ void relabel_seed_kernel(const int64_t *__restrict__ idx, int64_t n, int64_t M,
                                    int64_t *__restrict__ slot, unsigned long long *err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  if (v < 0 || v >= M) {
    atomicAdd(err, 1ull);
    return;
  }
  // a node listed twice keeps its LAST position, as the reference's sequential map insert does
  atomicMin(reinterpret_cast<long long *>(&slot[v]), (long long)(-(i + 1)));
}

// seeds listed twice, multi-hop samplers: neighbor_sample_cpu.cpp:31, 195 INSERT the seeds into the map, so the FIRST
// position keeps the node (relabel_cpu / sample_cpu assign, so the last one does: relabel_seed_kernel).  Runs behind
// relabel_seed_kernel: every seeded slot is negative by then, the largest -(i + 1) is the first position
__global__ void relabel_seed_first_kernel(const int64_t *__restrict__ idx, int64_t n, int64_t M,
                                          int64_t *__restrict__ slot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  if (v < 0 || v >= M) return;
  atomicMax(reinterpret_cast<long long *>(&slot[v]), (long long)(-(i + 1)));
}

__global__ void relabel_first_kernel(const int64_t *__restrict__ nbr, int64_t T, int64_t M,
                                     int64_t *__restrict__ slot, unsigned long long *err) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t c = nbr[t];
  if (c < 0 || c >= M) {
    atomicAdd(err, 1ull);
    return;
  }
  if (slot[c] >= 0) atomicMin(reinterpret_cast<long long *>(&slot[c]), (long long)t);
}

__global__ void relabel_flag_kernel(const int64_t *__restrict__ nbr, int64_t T, int64_t M,
                                    const int64_t *__restrict__ slot, int64_t *__restrict__ rank) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t c = nbr[t];
  rank[t] = (c >= 0 && c < M && slot[c] == t) ? 1 : 0;
}

__global__ void relabel_apply_kernel(const int64_t *__restrict__ idx, int64_t n,
                                     const int64_t *__restrict__ nbr, int64_t T, int64_t M,
                                     const int64_t *__restrict__ slot, const int64_t *__restrict__ rank,
                                     int64_t *__restrict__ local, int64_t *__restrict__ n_id) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && n_id) n_id[t] = idx[t];
  if (t >= T) return;
  const int64_t c = nbr[t];
  if (c < 0 || c >= M) return;
  const int64_t s = slot[c];
  if (s < 0) {
    if (local) local[t] = -(s + 1);
  } else {
    const int64_t id = n + rank[s];
    if (local) local[t] = id;
    if (s == t && n_id) n_id[id] = c;
  }
}

// the apply step of tsamd_relabel_extend: the list length n is read from the device counter; a first occurrence
// appends its node at n + rank and turns its slot into -(id + 1) -- a lane that reads the slot after that store
// computes the same id from the negative form, so the race between the two reads is benign (aligned 8-byte store)
__global__ void relabel_extend_kernel(const int64_t *__restrict__ nbr, int64_t T, int64_t M,
                                      int64_t *__restrict__ slot, const int64_t *__restrict__ rank,
                                      const int64_t *__restrict__ count, int64_t *__restrict__ local,
                                      int64_t *__restrict__ n_id, int64_t capacity, unsigned long long *err) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t c = nbr[t];
  if (c < 0 || c >= M) {
    if (local) local[t] = -1;
    return;
  }
  const int64_t n = *count;
  const int64_t s = *reinterpret_cast<volatile int64_t *>(&slot[c]);
  if (s < 0) {
    if (local) local[t] = -(s + 1);
    return;
  }
  const int64_t id = n + rank[s];
  if (local) local[t] = id;
  if (s == t) {
    if (id < capacity) n_id[id] = c;
    else atomicAdd(err, 1ull);
    *reinterpret_cast<volatile int64_t *>(&slot[c]) = -(id + 1);
  }
}

__global__ void relabel_count_add_kernel(int64_t *count, const int64_t *add) { *count += *add; }
__global__ void relabel_count_set_kernel(int64_t *count, int64_t n, int64_t *err) {
  *count = n;
  *err = 0;
}

// ---- temporal sampling: computation trees per root, nodes are (node, root) PAIRS -----------------------------
// (neighbor_sample_cpu.cpp:222-340, hetero_temporal_neighbor_sample).  The pair space is far too large for a dense slot
// array: the pairs already numbered and the hop's candidates are sorted together ONCE (stable: tsamd_sort_coo with
// pair = (row, col)); the first position of a run of equal pairs is the pair's first occurrence.  The time constraint is a
// flag per draw until the very end: nothing is compacted before the relabel, so a relation costs ONE size read-back.

// keep[t] = 1 when draw t satisfies node_time[src][v] <= root time of the node it was drawn for (or src has no time)
__global__ void temporal_mark_kernel(const int64_t *__restrict__ nbr, const int64_t *__restrict__ seg, int64_t T,
                                     const int64_t *__restrict__ src_time, const int64_t *__restrict__ f_time,
                                     int64_t *__restrict__ keep) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  keep[t] = src_time ? (src_time[nbr[t]] <= f_time[seg[t]] ? 1 : 0) : 1;
}

// order[r] = the draw that is the r-th kept one (rank = exclusive scan of the flags, T + 1 entries)
__global__ void temporal_order_kernel(const int64_t *__restrict__ rank, int64_t T, int64_t *__restrict__ order) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  if (rank[t + 1] != rank[t]) order[rank[t]] = t;
}

// with replacement: k uniform picks among the VALID neighbours of every frontier node (one lane per pick)
__global__ void temporal_redraw_kernel(const int64_t *__restrict__ out_ptr, int64_t F, int64_t k, uint64_t seed,
                                       const int64_t *__restrict__ rank, const int64_t *__restrict__ order,
                                       const int64_t *__restrict__ nbr, const int64_t *__restrict__ e,
                                       int64_t *__restrict__ nbr2, int64_t *__restrict__ e2,
                                       int64_t *__restrict__ seg2, int64_t *__restrict__ keep2) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= F * k) return;
  const int64_t i = x / k, j = x - i * k;
  const int64_t lo = rank[out_ptr[i]], cnt = rank[out_ptr[i + 1]] - lo;
  seg2[x] = i;
  if (cnt <= 0) {
    keep2[x] = 0;
    nbr2[x] = 0;
    e2[x] = 0;
    return;
  }
  const U4 r = philox(seed, (uint64_t)i, (uint32_t)j, 0x7E4Du ^ (uint32_t)((uint64_t)j >> 32));
  const int64_t t = order[lo + (int64_t)__umul64hi(u64(r.x, r.y), (uint64_t)cnt)];
  keep2[x] = 1;
  nbr2[x] = nbr[t];
  e2[x] = e[t];
}

// keys of the pair sort: the pairs numbered so far, then the candidates (those that do not count share (num_nodes, 0))
__global__ void temporal_keys_kernel(const int64_t *__restrict__ old_node, const int64_t *__restrict__ old_root,
                                     int64_t n, const int64_t *__restrict__ nbr, const int64_t *__restrict__ seg,
                                     const int64_t *__restrict__ f_root, const int64_t *__restrict__ keep, int64_t T,
                                     int64_t num_nodes, int64_t *__restrict__ node, int64_t *__restrict__ root) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n + T) return;
  if (p < n) {
    node[p] = old_node[p];
    root[p] = old_root[p];
  } else {
    const int64_t t = p - n;
    const bool kept = keep[t] != 0;
    node[p] = kept ? nbr[t] : num_nodes;
    root[p] = kept ? f_root[seg[t]] : 0;
  }
}

__global__ void temporal_head_kernel(const int64_t *__restrict__ node_s, const int64_t *__restrict__ root_s, int64_t L,
                                     int64_t *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  flag[i] = (i == 0 || node_s[i] != node_s[i - 1] || root_s[i] != root_s[i - 1]) ? 1 : 0;
}

// rid = exclusive scan of the head flags (L + 1 entries): sorted position i belongs to run rid[i + 1] - 1
__global__ void temporal_runfirst_kernel(const int64_t *__restrict__ rid, const int64_t *__restrict__ perm, int64_t L,
                                         int64_t *__restrict__ run_first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  if (rid[i + 1] != rid[i]) run_first[rid[i]] = perm[i];  // stable sort: the head of a run came first
}

__global__ void temporal_open_kernel(const int64_t *__restrict__ rid, const int64_t *__restrict__ perm,
                                     const int64_t *__restrict__ run_first, int64_t L, int64_t n,
                                     const int64_t *__restrict__ keep, int64_t *__restrict__ first_of,
                                     int64_t *__restrict__ open) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const int64_t p = perm[i], f = run_first[rid[i + 1] - 1];
  first_of[p] = f;
  if (p >= n) open[p - n] = (f == p && keep[p - n] != 0) ? 1 : 0;
}

__global__ void temporal_local_kernel(const int64_t *__restrict__ first_of, const int64_t *__restrict__ open_rank,
                                      int64_t n, int64_t T, int64_t *__restrict__ local) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t f = first_of[n + t];
  local[t] = f < n ? f : n + open_rank[f - n];
}

__global__ void temporal_emit_kernel(const int64_t *__restrict__ nbr, const int64_t *__restrict__ e,
                                     const int64_t *__restrict__ seg, const int64_t *__restrict__ f_root,
                                     const int64_t *__restrict__ f_time, const int64_t *__restrict__ keep_rank,
                                     const int64_t *__restrict__ open_rank, const int64_t *__restrict__ local,
                                     int64_t T, int64_t begin, int64_t *__restrict__ rows, int64_t *__restrict__ cols,
                                     int64_t *__restrict__ edges, int64_t *__restrict__ node_out,
                                     int64_t *__restrict__ root_out, int64_t *__restrict__ time_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t kr = keep_rank[t], orank = open_rank[t];
  if (keep_rank[t + 1] != kr) {
    rows[kr] = local[t];
    cols[kr] = seg[t] + begin;
    edges[kr] = e[t];
  }
  if (open_rank[t + 1] != orank) {
    const int64_t s = seg[t];
    node_out[orank] = nbr[t];
    root_out[orank] = f_root[s];
    time_out[orank] = f_time[s];
  }
}

// assoc[idx[i]] = i  (node -> position in the subset, -1 elsewhere; SAINT sub-graphs)
__global__ void assoc_kernel(const int64_t *__restrict__ idx, int64_t n, int64_t M,
                             int64_t *__restrict__ assoc, unsigned long long *err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  if (v < 0 || v >= M) {
    atomicAdd(err, 1ull);
    return;
  }
  atomicMax(reinterpret_cast<long long *>(&assoc[v]), (long long)i);  // duplicates: last position wins
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" int tsamd_random_walk(const int64_t *rowptr, const int64_t *col, const int64_t *start,
                                 const float *rand, int64_t n, int64_t walk_length, int64_t *out,
                                 void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || walk_length < 0) return TSAMD_ERR_INVALID;
  if (n == 0) return TSAMD_OK;
  if (!rowptr || !start || !out || (walk_length > 0 && (!rand || !col))) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(random_walk_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                     rowptr, col, start, rand, n, walk_length, out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_sample_workspace_bytes(int64_t n) { return scan_workspace_bytes(n + 1); }

extern "C" int tsamd_sample_plan(const int64_t *rowptr, int64_t M, const int64_t *idx, int64_t n,
                                 int64_t num_neighbors, int replace, int64_t *out_ptr, int64_t *info,
                                 void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || n < 0 || !rowptr || !out_ptr || !info || (n > 0 && !idx)) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_sample_workspace_bytes(n)) return TSAMD_ERR_WORKSPACE;
  TSAMD_HIP_TRY(hipMemsetAsync(info, 0, 2 * sizeof(int64_t), stream));
  TSAMD_HIP_TRY(hipMemsetAsync(out_ptr + n, 0, sizeof(int64_t), stream));
  if (n > 0) {
    hipLaunchKernelGGL(sample_count_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                       rowptr, M, idx, n, num_neighbors, replace, out_ptr,
                       reinterpret_cast<unsigned long long *>(info + 1));
    TSAMD_LAUNCH_CHECK();
  }
  return exclusive_scan_i64(out_ptr, out_ptr, n + 1, info, workspace, stream);
}

extern "C" int tsamd_sample_draw(const int64_t *rowptr, const int64_t *col, const int64_t *idx,
                                 int64_t n, int64_t num_neighbors, int replace, uint64_t seed,
                                 const int64_t *out_ptr, int64_t *e_id, int64_t *nbr, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || num_neighbors < 0) return TSAMD_ERR_INVALID;  // "all neighbours" is tsamd_select_fill
  const int64_t total = n * num_neighbors;
  if (total == 0) return TSAMD_OK;
  if (!rowptr || !col || !idx || !out_ptr || !e_id || !nbr) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(sample_draw_kernel, dim3((unsigned int)ceil_div(total, 256)), dim3(256), 0, stream,
                     rowptr, col, idx, n, num_neighbors, replace, seed, out_ptr, e_id, nbr);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_relabel_workspace_bytes(int64_t T) { return scan_workspace_bytes(T + 1); }

extern "C" int tsamd_relabel_plan(const int64_t *idx, int64_t n, const int64_t *nbr, int64_t T,
                                  int64_t M, int64_t *slot, int64_t *rank, int64_t *info,
                                  void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || T < 0 || M < 0 || !rank || !info) return TSAMD_ERR_INVALID;
  if ((M > 0 && !slot) || (n > 0 && !idx) || (T > 0 && !nbr)) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_relabel_workspace_bytes(T)) return TSAMD_ERR_WORKSPACE;
  TSAMD_HIP_TRY(hipMemsetAsync(info, 0, 2 * sizeof(int64_t), stream));
  // every slot = 0x7f7f... : "not seen", larger than any draw position
  if (M > 0) TSAMD_HIP_TRY(hipMemsetAsync(slot, 0x7f, sizeof(int64_t) * (size_t)M, stream));
  TSAMD_HIP_TRY(hipMemsetAsync(rank + T, 0, sizeof(int64_t), stream));
  unsigned long long *err = reinterpret_cast<unsigned long long *>(info + 1);
  if (n > 0) {
    hipLaunchKernelGGL(relabel_seed_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream,
                       idx, n, M, slot, err);
    TSAMD_LAUNCH_CHECK();
  }
  if (T > 0) {
    const unsigned int blocks = (unsigned int)ceil_div(T, 256);
    hipLaunchKernelGGL(relabel_first_kernel, dim3(blocks), dim3(256), 0, stream, nbr, T, M, slot, err);
    TSAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(relabel_flag_kernel, dim3(blocks), dim3(256), 0, stream, nbr, T, M,
                       (const int64_t *)slot, rank);
    TSAMD_LAUNCH_CHECK();
  }
  return exclusive_scan_i64(rank, rank, T + 1, info, workspace, stream);
}

extern "C" int tsamd_relabel_apply(const int64_t *idx, int64_t n, const int64_t *nbr, int64_t T,
                                   int64_t M, const int64_t *slot, const int64_t *rank,
                                   int64_t *local, int64_t *n_id, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || T < 0 || M < 0) return TSAMD_ERR_INVALID;
  const int64_t work = n > T ? n : T;
  if (work == 0) return TSAMD_OK;
  if ((T > 0 && (!nbr || !slot || !rank)) || (n > 0 && n_id && !idx)) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(relabel_apply_kernel, dim3((unsigned int)ceil_div(work, 256)), dim3(256), 0, stream,
                     idx, n, nbr, T, M, slot, rank, local, n_id);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_relabel_seed(const int64_t *idx, int64_t n, int64_t M, int64_t *slot, int64_t *count,
                                  int64_t *err, int first_wins, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || M < 0 || !count || !err || (M > 0 && !slot) || (n > 0 && !idx)) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(relabel_count_set_kernel, dim3(1), dim3(1), 0, stream, count, n, err);
  TSAMD_LAUNCH_CHECK();
  if (M > 0) TSAMD_HIP_TRY(hipMemsetAsync(slot, 0x7f, sizeof(int64_t) * (size_t)M, stream));
  if (n > 0) {
    hipLaunchKernelGGL(relabel_seed_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream, idx, n, M,
                       slot, reinterpret_cast<unsigned long long *>(err));
    TSAMD_LAUNCH_CHECK();
    if (first_wins) {
      hipLaunchKernelGGL(relabel_seed_first_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream, idx, n, M,
                         slot);
      TSAMD_LAUNCH_CHECK();
    }
  }
  return TSAMD_OK;
}

extern "C" int tsamd_relabel_extend(const int64_t *nbr, int64_t T, int64_t M, int64_t *slot, int64_t *rank,
                                    int64_t *count, int64_t *local, int64_t *n_id, int64_t capacity, int64_t *err,
                                    void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (T < 0 || M < 0 || capacity < 0 || !count || !err) return TSAMD_ERR_INVALID;
  if (T == 0) return TSAMD_OK;
  if (!nbr || !rank || !n_id || (M > 0 && !slot)) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_relabel_workspace_bytes(T)) return TSAMD_ERR_WORKSPACE;
  unsigned long long *e = reinterpret_cast<unsigned long long *>(err);
  const unsigned int blocks = (unsigned int)ceil_div(T, 256);
  TSAMD_HIP_TRY(hipMemsetAsync(rank + T, 0, sizeof(int64_t), stream));
  hipLaunchKernelGGL(relabel_first_kernel, dim3(blocks), dim3(256), 0, stream, nbr, T, M, slot, e);
  TSAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(relabel_flag_kernel, dim3(blocks), dim3(256), 0, stream, nbr, T, M, (const int64_t *)slot, rank);
  TSAMD_LAUNCH_CHECK();
  const int st = exclusive_scan_i64(rank, rank, T + 1, nullptr, workspace, stream);  // rank[T] = number of new nodes
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL(relabel_extend_kernel, dim3(blocks), dim3(256), 0, stream, nbr, T, M, slot, (const int64_t *)rank,
                     (const int64_t *)count, local, n_id, capacity, e);
  TSAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(relabel_count_add_kernel, dim3(1), dim3(1), 0, stream, count, (const int64_t *)(rank + T));
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_temporal_mark(const int64_t *nbr, const int64_t *seg, int64_t T, const int64_t *src_time,
                                   const int64_t *f_time, int64_t *keep, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (T < 0) return TSAMD_ERR_INVALID;
  if (T == 0) return TSAMD_OK;
  if (!keep || (src_time && (!nbr || !seg || !f_time))) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(temporal_mark_kernel, dim3((unsigned int)ceil_div(T, 256)), dim3(256), 0, stream, nbr, seg, T,
                     src_time, f_time, keep);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_temporal_redraw_workspace_bytes(int64_t T) {
  return align_up(sizeof(int64_t) * (size_t)(T + 1), 256) + align_up(sizeof(int64_t) * (size_t)(T > 0 ? T : 1), 256) +
         scan_workspace_bytes(T + 1);
}

extern "C" int tsamd_temporal_redraw(const int64_t *out_ptr, int64_t F, int64_t T, int64_t k, uint64_t seed,
                                     const int64_t *nbr, const int64_t *e, const int64_t *keep, int64_t *nbr2,
                                     int64_t *e2, int64_t *seg2, int64_t *keep2, void *workspace,
                                     size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (F < 0 || T < 0 || k < 0) return TSAMD_ERR_INVALID;
  if (F * k == 0) return TSAMD_OK;
  if (!out_ptr || !nbr2 || !e2 || !seg2 || !keep2 || (T > 0 && (!nbr || !e || !keep))) return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_temporal_redraw_workspace_bytes(T)) return TSAMD_ERR_WORKSPACE;
  char *w = reinterpret_cast<char *>(workspace);
  int64_t *rank = reinterpret_cast<int64_t *>(w);
  w += align_up(sizeof(int64_t) * (size_t)(T + 1), 256);
  int64_t *order = reinterpret_cast<int64_t *>(w);
  w += align_up(sizeof(int64_t) * (size_t)(T > 0 ? T : 1), 256);
  if (T > 0) TSAMD_HIP_TRY(hipMemcpyAsync(rank, keep, sizeof(int64_t) * (size_t)T, hipMemcpyDeviceToDevice, stream));
  TSAMD_HIP_TRY(hipMemsetAsync(rank + T, 0, sizeof(int64_t), stream));
  const int st = exclusive_scan_i64(rank, rank, T + 1, nullptr, w, stream);
  if (st != TSAMD_OK) return st;
  if (T > 0) {
    hipLaunchKernelGGL(temporal_order_kernel, dim3((unsigned int)ceil_div(T, 256)), dim3(256), 0, stream,
                       (const int64_t *)rank, T, order);
    TSAMD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(temporal_redraw_kernel, dim3((unsigned int)ceil_div(F * k, 256)), dim3(256), 0, stream, out_ptr, F, k,
                     seed, (const int64_t *)rank, (const int64_t *)order, nbr, e, nbr2, e2, seg2, keep2);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

namespace {
struct TemporalCarve {
  int64_t *node, *root, *node_s, *root_s, *perm, *rid, *run_first, *first_of;
  void *sort_ws, *scan_ws;
  size_t sort_bytes, total;
};
TemporalCarve temporal_carve(void *workspace, int64_t L) {
  TemporalCarve c;
  const uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  size_t off = 0;
  const size_t a = align_up(sizeof(int64_t) * (size_t)(L + 1), 256);
  int64_t **arr[8] = {&c.node, &c.root, &c.node_s, &c.root_s, &c.perm, &c.rid, &c.run_first, &c.first_of};
  for (auto *q : arr) {
    *q = reinterpret_cast<int64_t *>(base + off);
    off += a;
  }
  c.sort_bytes = tsamd_sort_coo_workspace_bytes(L);
  c.sort_ws = reinterpret_cast<void *>(base + off);
  off += align_up(c.sort_bytes, 256);
  c.scan_ws = reinterpret_cast<void *>(base + off);
  off += scan_workspace_bytes(L + 1);
  c.total = off;
  return c;
}
}  // namespace

extern "C" size_t tsamd_temporal_relabel_workspace_bytes(int64_t n, int64_t T) {
  return temporal_carve(nullptr, n + T).total + 256;
}

extern "C" int tsamd_temporal_relabel(const int64_t *old_node, const int64_t *old_root, int64_t n, const int64_t *nbr,
                                      const int64_t *seg, const int64_t *f_root, const int64_t *keep, int64_t T,
                                      int64_t num_nodes, int64_t num_roots, int64_t *local, int64_t *keep_rank,
                                      int64_t *open_rank, int64_t *info, void *workspace, size_t workspace_bytes,
                                      void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || T < 0 || num_nodes < 0 || num_roots < 0 || !info) return TSAMD_ERR_INVALID;
  if (T == 0) {
    TSAMD_HIP_TRY(hipMemsetAsync(info, 0, 2 * sizeof(int64_t), stream));
    return TSAMD_OK;
  }
  if (!nbr || !seg || !f_root || !keep || !local || !keep_rank || !open_rank || (n > 0 && (!old_node || !old_root)))
    return TSAMD_ERR_INVALID;
  if (!workspace || workspace_bytes < tsamd_temporal_relabel_workspace_bytes(n, T)) return TSAMD_ERR_WORKSPACE;
  const int64_t L = n + T;
  const TemporalCarve c = temporal_carve(workspace, L);
  const unsigned int bl = (unsigned int)ceil_div(L, 256), bt = (unsigned int)ceil_div(T, 256);
  hipLaunchKernelGGL(temporal_keys_kernel, dim3(bl), dim3(256), 0, stream, old_node, old_root, n, nbr, seg, f_root, keep, T,
                     num_nodes, c.node, c.root);
  TSAMD_LAUNCH_CHECK();
  int st = tsamd_sort_coo(c.node, c.root, L, num_nodes + 1, num_roots > 0 ? num_roots : 1, c.node_s, c.root_s, c.perm,
                          c.sort_ws, c.sort_bytes, stream_);
  if (st != TSAMD_OK) return st;
  TSAMD_HIP_TRY(hipMemsetAsync(c.rid + L, 0, sizeof(int64_t), stream));
  hipLaunchKernelGGL(temporal_head_kernel, dim3(bl), dim3(256), 0, stream, (const int64_t *)c.node_s,
                     (const int64_t *)c.root_s, L, c.rid);
  TSAMD_LAUNCH_CHECK();
  st = exclusive_scan_i64(c.rid, c.rid, L + 1, nullptr, c.scan_ws, stream);
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL(temporal_runfirst_kernel, dim3(bl), dim3(256), 0, stream, (const int64_t *)c.rid,
                     (const int64_t *)c.perm, L, c.run_first);
  TSAMD_LAUNCH_CHECK();
  TSAMD_HIP_TRY(hipMemsetAsync(open_rank + T, 0, sizeof(int64_t), stream));
  hipLaunchKernelGGL(temporal_open_kernel, dim3(bl), dim3(256), 0, stream, (const int64_t *)c.rid, (const int64_t *)c.perm,
                     (const int64_t *)c.run_first, L, n, keep, c.first_of, open_rank);
  TSAMD_LAUNCH_CHECK();
  st = exclusive_scan_i64(open_rank, open_rank, T + 1, info + 1, c.scan_ws, stream);
  if (st != TSAMD_OK) return st;
  TSAMD_HIP_TRY(hipMemcpyAsync(keep_rank, keep, sizeof(int64_t) * (size_t)T, hipMemcpyDeviceToDevice, stream));
  TSAMD_HIP_TRY(hipMemsetAsync(keep_rank + T, 0, sizeof(int64_t), stream));
  st = exclusive_scan_i64(keep_rank, keep_rank, T + 1, info, c.scan_ws, stream);
  if (st != TSAMD_OK) return st;
  hipLaunchKernelGGL(temporal_local_kernel, dim3(bt), dim3(256), 0, stream, (const int64_t *)c.first_of,
                     (const int64_t *)open_rank, n, T, local);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_temporal_emit(const int64_t *nbr, const int64_t *e, const int64_t *seg, const int64_t *f_root,
                                   const int64_t *f_time, const int64_t *keep_rank, const int64_t *open_rank,
                                   const int64_t *local, int64_t T, int64_t begin, int64_t *rows, int64_t *cols,
                                   int64_t *edges, int64_t *node_out, int64_t *root_out, int64_t *time_out,
                                   void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (T < 0) return TSAMD_ERR_INVALID;
  if (T == 0) return TSAMD_OK;
  if (!nbr || !e || !seg || !f_root || !f_time || !keep_rank || !open_rank || !local) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(temporal_emit_kernel, dim3((unsigned int)ceil_div(T, 256)), dim3(256), 0, stream, nbr, e, seg, f_root,
                     f_time, keep_rank, open_rank, local, T, begin, rows, cols, edges, node_out, root_out, time_out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" int tsamd_subset_assoc(const int64_t *idx, int64_t n, int64_t M, int64_t *assoc,
                                  int64_t *err, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || M < 0 || !err || (M > 0 && !assoc) || (n > 0 && !idx)) return TSAMD_ERR_INVALID;
  TSAMD_HIP_TRY(hipMemsetAsync(err, 0, sizeof(int64_t), stream));
  if (M > 0) TSAMD_HIP_TRY(hipMemsetAsync(assoc, 0xff, sizeof(int64_t) * (size_t)M, stream));  // -1
  if (n > 0) {
    hipLaunchKernelGGL(assoc_kernel, dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0, stream, idx, n,
                       M, assoc, reinterpret_cast<unsigned long long *>(err));
    TSAMD_LAUNCH_CHECK();
  }
  return TSAMD_OK;
}
