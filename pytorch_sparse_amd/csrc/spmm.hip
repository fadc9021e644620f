// CSR SpMM forward for gfx950 (MI355X): merge-path balanced, wave64 row-split.
//
// Replaces spmm_cuda / spmm_cpu of the reference (csrc/cuda/spmm_cuda.cu:92-155,
// csrc/cpu/spmm_cpu.cpp:8-101).  The arithmetic contract (init values, strict
// compares, first-occurrence ties, empty-row handling, mean divisor) follows
// csrc/cpu/reducer.h:43-84.
//
// Why not "one wave per row" (the reference's mapping): on power-law graphs the
// row degree is correlated with the row index bits, the hardware deals
// workgroups to the 8 XCDs round-robin, and one XCD ends up with ~44 % of the
// edges of an R-MAT matrix (measured: 2x slowdown).  Here the
// work list "M row ends + E edges" is cut into P equal pieces along the merge
// path (Merrill & Garland's SpMV decomposition), so every wavefront gets the
// same number of (row, edge) items whatever the degree distribution, hub rows
// are split over many waves, and no atomics are needed:
//
//   1. spmm_partition_kernel   P+1 diagonal binary searches -> (row, edge) table
//   2. spmm_merge_kernel       wave p walks its rows/edges:
//        * (col, value) arrive in 64-edge windows, one coalesced load each,
//          the next window is requested before the current one is consumed;
//        * the 64 lanes form G = 64/LPR groups of LPR lanes x VEC features
//          (16 B per lane), so one vector-memory instruction gathers G rows of
//          `mat`, each as one contiguous LPR*16-byte read; U such gathers are
//          issued back to back (G*U rows in flight per wave);
//        * window entries reach the groups through ds_bpermute (no LDS);
//        * a row that ends inside the piece is reduced across groups
//          (bpermute butterfly) and stored once; the pieces of a row that is
//          cut by a partition boundary go to carry records (accumulator
//          precision, plus the winner's offset in the partition for min/max);
//          the wave also leaves the ids of its unfinished last row and of a
//          cut first row that ended in it (tail_row / head_row);
//   3. spmm_fixup_kernel       the wave of the partition in which a cut row
//        ends (head_row) folds that row's carry records (ties -> smaller edge
//        id) and writes the final value (mean divide / empty handling happen
//        here); everything it needs is fetched in one round trip.
//
// Deterministic: the partition only depends on rowptr, every combine order is fixed.
#include "common.h"
#include "spmm_internal.h"

#include <cstdlib>
#include <type_traits>

namespace tsamd {
namespace {

// This file is compiled twice: as itself (every entry point but tsamd_spmm_partial; kPartial = false, the kernels
// carry no trace of the partial-product mode -- a run-time switch cost the north-star instantiation a wave per SIMD)
// and through spmm_partial.hip (TSAMD_SPMM_PARTIAL_BUILD = 1: only tsamd_spmm_partial, floating-point types).
#ifndef TSAMD_SPMM_PARTIAL_BUILD
#define TSAMD_SPMM_PARTIAL_BUILD 0
#endif
constexpr bool kPartial = TSAMD_SPMM_PARTIAL_BUILD != 0;
// ... and its ~200 merge-kernel instantiations are split over two translation units that compile in parallel (176 s
// in one): TSAMD_SPMM_TU 1 = this file (sum / mean and the masked sum, every entry point; min / max are launched through
// spmm_min_bridge / spmm_max_bridge), 2 / 3 = spmm_min.hip / spmm_max.hip (the min / the max instantiations and their
// bridge), 0 = everything in one unit (the partial build).
#ifndef TSAMD_SPMM_TU
#if TSAMD_SPMM_PARTIAL_BUILD
#define TSAMD_SPMM_TU 0
#else
#define TSAMD_SPMM_TU 1
#endif
#endif

[[maybe_unused]] constexpr int RED_ADD = 0;  // sum and mean
[[maybe_unused]] constexpr int RED_MIN = 1;
[[maybe_unused]] constexpr int RED_MAX = 2;

// tuning knobs (overridable with -D for A/B experiments, see scripts/variants.py)
#ifndef TSAMD_UNROLL
#define TSAMD_UNROLL 4
#endif
#ifndef TSAMD_WPB
#define TSAMD_WPB 4
#endif
#ifndef TSAMD_ITEMS_MAX
#define TSAMD_ITEMS_MAX 1024
#endif
#ifndef TSAMD_ITEMS_MIN
#define TSAMD_ITEMS_MIN 128
#endif
#ifndef TSAMD_TARGET_WAVES
#define TSAMD_TARGET_WAVES 32768
#endif
#ifndef TSAMD_MINMAX_UNROLL
#define TSAMD_MINMAX_UNROLL 2
#endif
#ifndef TSAMD_MASKED_SEGMENT_SKIP
#define TSAMD_MASKED_SEGMENT_SKIP 1  // 0: the masked sum gathers every entry's whole row (round 3), for A/B builds
#endif
#ifndef TSAMD_MINMAX_SPLIT_LOOP
#define TSAMD_MINMAX_SPLIT_LOOP 1  // 0: the round-3 loop (value-mode branch inside the step loop), kept for A/B builds
#endif
constexpr int kUnroll = TSAMD_UNROLL;      // gathers in flight per group
constexpr int kMinMaxUnroll = TSAMD_MINMAX_UNROLL;  // min / max carry (value, arg) per element: fewer
constexpr int kWavesPerBlock = TSAMD_WPB;  // 256-thread workgroups
constexpr int64_t kNoArg = 0x7fffffffffffffffLL;

struct Coord {  // a point on the merge path: rows [0,row) done, edges [0,edge) consumed
  int64_t row;
  int64_t edge;
};

struct Workspace {
  Coord *table;       // [P+1]
  int64_t *tail_row;  // [P]   row id of the unfinished row's partial, or -1
  int64_t *head_row;  // [P]   row id of the cut row that ends in the partition (its head record is there), or -1
  void *head_val;     // [B][P][K] acc_t : piece of the first row (it started earlier)
  void *tail_val;     // [B][P][K] acc_t : piece of the last row (it continues later)
  uint32_t *head_arg;  // min/max only: winners as 32-bit offsets from the partition's first edge (table[p].edge),
  uint32_t *tail_arg;  // kNoArg32 = none -- as the merge kernel holds them; the fix-up kernel widens them
  int64_t P;
  int64_t items;  // (row, edge) items per partition
  // channel-camping avoidance (see "relabel" below)
  int relabel_mode;   // host decision: 0 off, 1 on, 2 = decide on the device from the sample
  int *relabel_flag;  // device int[4]: sample counters (see use_relabel)
  void *xperm;        // [B][N][K] copy of mat with rows at hashed positions
  uint32_t hash_bits, hash_mul, hash_shift;
  // relabelled layout end to end (tsamd_spmm_relabelled): output row m is stored at position
  // hash_row(m, M, ...), `col` already holds hashed ids and `mat` is already in hashed row order
  int out_relabel;
  uint32_t ohash_bits, ohash_shift;
  // entries taken through a permutation (tsamd_spmm_permuted): entry e of the CSR is
  // (col[perm[e]], value[perm[e]]) -- the CSC view of a matrix without materialising it
  const int64_t *perm;
  // masked sum (spmm_masked_sum, the pull formulation of the min/max backward): one record of
  // `rec_stride` 32-bit words per (batch, entry) -- see WinRecord in spmm_internal.h: the mask words
  // (bit k of word k / 32 = "feature k of this entry contributes"), then the entry's column id and
  // value, so that one 32-byte line serves every random access an entry needs
  const uint32_t *wmask;
  uint32_t rec_stride, rec_meta;  // words per record; word offset of (id, value lo, value hi)
  // word rec_meta + 3 of a record, when the padding leaves one (rec_has_z): bit s = "mask word s is non-zero", i.e.
  // the 64-byte segment s of the gathered row (32 two-byte features; 128 bytes of 4-byte ones) has a winner in
  // this entry at all -- segments without one are neither gathered nor is their mask word read
  int rec_has_z;
  // operand cache (tsamd_spmm_cached): xperm / relabel_flag live in a caller-owned buffer that survives the
  // call; when both pointers are set the copy kernel returns at once if the two fingerprints agree
  const unsigned long long *fp_stored, *fp_new;
  unsigned long long *cache_fp;  // host side: [2][kFingerprintWords] stored | new, inside the cache buffer
  int cache_state;               // host side: 0 no cache, 1 fill it, 2 reuse it if the fingerprint still matches
  // partial product of one COLUMN BLOCK of a matrix (tsamd_spmm_partial: the stages of the overlapped all-gather,
  // pytorch_sparse_amd/parallel.py): the CSR holds the block's entries only;
  //   accumulate   combine with what out / arg_out hold from the earlier blocks instead of overwriting them
  //   arg_map      min / max: block entry id -> entry id of the whole matrix (what arg_out reports; ties between
  //                blocks go to the smaller id, i.e. to the first occurrence in the whole row as reducer.h:63-67)
  //   arg_none     the whole matrix's "no winner" id (its E)
  //   deg_rowptr   mean: the divisor is the length of the WHOLE row, deg_rowptr[r + 1] - deg_rowptr[r]
  int partial, accumulate;
  const int64_t *arg_map;
  int64_t arg_none;
  const int64_t *deg_rowptr;
  // min / max: the winners are stored as 32-bit entry ids (tsamd_spmm_minmax_arg32: callers that keep them only
  // for their own backward -- half the bytes of the API's int64 arg_out in the forward store and the backward read)
  int arg32;
  // winner records written by the forward (tsamd_spmm_minmax_records: rows of 97..128 features, 4-byte accumulators,
  // int32 winners): at the end of every row that lies inside ONE partition the wave writes the 32-byte record (WinRecord
  // in spmm_internal.h, what minmax_winrec_kernel derives from arg_out) of each of its entries -- the winners are in
  // registers there.  Rows cut between partitions: every piece gets records without winners from the wave that
  // holds it, and the fix-up kernel -- it learns the winners -- writes the whole records of a short row again
  // (<= kFixupRecordMax entries) or enters the winners into the at most K records of a long one that have any.  No ids
  // are stored anywhere.
  uint32_t *rec_out;      // [B][E][8] or nullptr
  const void *rec_value;  // the matrix's values (or nullptr) for the fix-up kernel's records
  int64_t snap;           // see spmm_partition_kernel
};

// ---------------------------------------------------------------------------
// 0. relabel: Kronecker / R-MAT style graphs put their hub columns at indices with few set
//    bits; with a 512-byte row pitch those rows share their low address bits and camp on a few
//    memory channels (measured on MI355X: the same graph runs 1.42x faster when the column ids
//    are relabelled at random).  When a sample of `col` shows that skew, `mat` is copied once
//    with its rows at hashed positions and the gather uses the hashed ids.  The hash is a
//    bijection on [0, N): multiply by an odd constant and fold the high half into the low half
//    on ceil(log2 N) bits, cycle-walking until the value is < N.  Sums are bit-identical with
//    and without it (only addresses change).  TSAMD_SPMM_RELABEL=0|1 forces it off|on.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash_row(uint32_t c, uint32_t N, uint32_t bits, uint32_t mul,
                                             uint32_t shift) {
  const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  do {
    c = (c * mul) & mask;
    c ^= c >> shift;
  } while (c >= N);
  return c;
}

__device__ __forceinline__ uint64_t out_position(const Workspace &ws, int64_t r, int64_t M) {
  return ws.out_relabel ? (uint64_t)hash_row((uint32_t)r, (uint32_t)M, ws.ohash_bits, ws.hash_mul, ws.ohash_shift)
                        : (uint64_t)r;
}

// counters: [1] #sampled ids with 3 low zero
// bits, [2] with 6 low zero bits, [3] #samples.  Uniform ids give 1/8 and 1/64 of the samples.
__device__ __forceinline__ bool use_relabel(int mode, const int *f) {
  if (mode != 2) return mode == 1;
  const int n = f[3];
  return n >= 4096 && (f[1] * 4 > n || f[2] * 16 > n);
}

constexpr int kProbeBlocks = 64;
#ifndef TSAMD_PERMUTE_BLOCKS
#define TSAMD_PERMUTE_BLOCKS 8192
#endif
#ifndef TSAMD_PERMUTE_NT
#define TSAMD_PERMUTE_NT 1
#endif

// Fingerprint of a dense operand for the operand cache: 64 x 256 sixteen-byte packets spread evenly over
// the matrix, mixed with their sample index and summed per block (wrap-around, order independent).  Any
// dense update (x += ..., a new epoch's activations) changes it with certainty for all practical purposes;
// it is the second line of defence behind the tensor's version counter (ops_spmm.cpp), for writes that
// bypass it.
constexpr int kFingerprintWords = 64;
__global__ __launch_bounds__(256) void spmm_fingerprint_kernel(const uint4 *__restrict__ mat, uint64_t npackets,
                                                               unsigned long long *__restrict__ out) {
  __shared__ unsigned long long part[4];
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t pos = (i * npackets) / ((uint64_t)kFingerprintWords * 256);
  const uint4 v = mat[pos];
  unsigned long long h = ((unsigned long long)v.x | ((unsigned long long)v.y << 32)) * 0x9E3779B97F4A7C15ull +
                         ((unsigned long long)v.z | ((unsigned long long)v.w << 32)) * 0xC2B2AE3D27D4EB4Full;
  h ^= h >> 29;
  h *= (2 * i + 1);
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = lane_read_u32((uint32_t)h, (int)((threadIdx.x & 63) ^ off));
    const uint32_t hi = lane_read_u32((uint32_t)(h >> 32), (int)((threadIdx.x & 63) ^ off));
    h += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void spmm_probe_kernel(const int64_t *__restrict__ col, int64_t E, int *__restrict__ flag) {
  const int64_t samples = (int64_t)kProbeBlocks * blockDim.x;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t pos = (i * E) / samples;  // evenly spread over the edge list
  const uint64_t c = (uint64_t)col[pos];
  const unsigned long long m3 = __ballot((c & 7u) == 0), m6 = __ballot((c & 63u) == 0);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&flag[1], __popcll(m3));
    atomicAdd(&flag[2], __popcll(m6));
    atomicAdd(&flag[3], 64);
  }
}

// lgL = log2(lanes per row); a 256-thread block copies 256 >> lgL rows, 16 bytes per lane and
// step (no integer division on the packet index).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void spmm_permute_rows_kernel(const T *__restrict__ mat,
                                                               T *__restrict__ xperm, int64_t BN,
                                                               uint32_t N, uint32_t K, int lgL,
                                                               Workspace ws) {
  if (!use_relabel(ws.relabel_mode, ws.relabel_flag)) return;
  if (ws.fp_stored != nullptr) {  // cached copy still matches the operand's fingerprint: nothing to do
    const int i = (int)(threadIdx.x & (kFingerprintWords - 1));
    if (__syncthreads_and(ws.fp_stored[i] == ws.fp_new[i])) return;
  }
  using P = Pack<T, VEC>;
  const uint32_t slots = K / VEC;
  const uint32_t lanes = 1u << lgL;
  const uint32_t sl0 = threadIdx.x & (lanes - 1);
  const int64_t rows_per_block = 256 >> lgL;
  for (int64_t r = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> lgL); r < BN;
       r += (int64_t)gridDim.x * rows_per_block) {
    const int64_t b = BN == (int64_t)N ? 0 : r / N;
    const uint32_t i = (uint32_t)(r - b * N);
    const uint32_t j = hash_row(i, N, ws.hash_bits, ws.hash_mul, ws.hash_shift);
    const P *src = reinterpret_cast<const P *>(mat) + (uint64_t)r * slots;
    P *dst = reinterpret_cast<P *>(xperm) + ((uint64_t)b * N + j) * slots;
    for (uint32_t sl = sl0; sl < slots; sl += lanes) {
      // the source is streamed once: a non-temporal load keeps it from evicting lines of the copy that the merge
      // kernel is about to gather (same-box A/B, north star: copy + probe + partition 0.44 -> 0.40-0.42 ms;
      // a non-temporal store on top changed nothing; TSAMD_PERMUTE_NT=0 restores plain loads)
#if TSAMD_PERMUTE_NT
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      static_assert(sizeof(P) == 16, "16-byte packets");
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + sl));
#if TSAMD_PERMUTE_NT >= 2
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst + sl));
#else
      *reinterpret_cast<u32x4 *>(dst + sl) = v;
#endif
#else
      dst[sl] = src[sl];
#endif
    }
  }
}

// ---------------------------------------------------------------------------
// 1. merge-path partition: list A = row ends rowptr[1..M], list B = edge ids
// ---------------------------------------------------------------------------
__global__ void spmm_partition_kernel(const int64_t *__restrict__ rowptr, int64_t M, int64_t E,
                                      Workspace ws) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > ws.P) return;
  int64_t d = p * ws.items;
  if (d > M + E) d = M + E;
  int64_t lo = d > E ? d - E : 0;
  int64_t hi = d < M ? d : M;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid + 1] <= d - mid - 1) lo = mid + 1;
    else hi = mid;
  }
  int64_t e = d - lo;
  // record-writing forward (Workspace::rec_out): a split that falls into the first `snap` entries of a row moves back to
  // the row's start, so that rows of up to `snap` entries are never cut (a cut row's records cost a second pass);
  // partitions then hold items .. items + snap items.  snap <= items / 2 keeps the table monotonic.
  if (ws.snap > 0 && lo < M) {
    const int64_t rs = rowptr[lo];
    if (e > rs && e - rs <= ws.snap) e = rs;
  }
  ws.table[p] = Coord{lo, e};
}

// ---------------------------------------------------------------------------
// accumulation helpers
// ---------------------------------------------------------------------------
constexpr uint32_t kNoArg32 = 0xFFFFFFFFu;  // in-kernel args are 32-bit offsets from the partition's first edge

template <typename T, int VEC, int RED, typename ARG>
__device__ __forceinline__ void init_acc(typename Traits<T>::acc_t (&val)[VEC], ARG (&arg)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if constexpr (RED == RED_ADD) val[j] = 0;
    else if constexpr (RED == RED_MIN) val[j] = Traits<T>::max_init();
    else val[j] = Traits<T>::lowest_init();
    arg[j] = (ARG)(sizeof(ARG) == 4 ? (int64_t)kNoArg32 : kNoArg);
  }
}

// The elements of a gathered packet as accumulator values.  bf16 packets are taken apart dword by dword (low half:
// one shift, high half: one AND): left to itself the compiler treats an 8-byte packet as ONE 64-bit integer and spends
// v_alignbit + v_and on the element that starts at bit 32.
template <typename T, int VEC>
__device__ __forceinline__ void unpack_packet(const Pack<T, VEC> &x, typename Traits<T>::acc_t (&out)[VEC]) {
  if constexpr (std::is_same<T, bf16_t>::value && VEC % 2 == 0) {
#pragma unroll
    for (int d = 0; d < VEC / 2; ++d) {
      uint32_t word;
      __builtin_memcpy(&word, reinterpret_cast<const char *>(&x) + 4 * d, 4);
      asm volatile("" : "+v"(word));
      const uint32_t lo = word << 16, hi = word & 0xFFFF0000u;
      __builtin_memcpy(&out[2 * d], &lo, 4);
      __builtin_memcpy(&out[2 * d + 1], &hi, 4);
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = Traits<T>::to_acc(x.v[j]);
  }
}

// w * x rounded to the element type, for the VEC elements of one packet (what `value * mat` is before the reducer
// sees it, reducer.h:63-67).  bf16: two products per v_cvt_pk_bf16_f32, taken apart again by one shift / one AND.
template <typename T, int VEC>
__device__ __forceinline__ void round_products(typename Traits<T>::acc_t w, typename Traits<T>::acc_t (&xv)[VEC]) {
  if constexpr (std::is_same<T, bf16_t>::value && VEC % 2 == 0) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int d = 0; d < VEC / 2; ++d) {
      f32x2 pr;
      pr.x = w * xv[2 * d];
      pr.y = w * xv[2 * d + 1];
      const bf16x2 h = __builtin_convertvector(pr, bf16x2);
      uint32_t word;
      __builtin_memcpy(&word, &h, 4);
      asm volatile("" : "+v"(word));
      const uint32_t lo = word << 16, hi = word & 0xFFFF0000u;
      __builtin_memcpy(&xv[2 * d], &lo, 4);
      __builtin_memcpy(&xv[2 * d + 1], &hi, 4);
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) xv[j] = Traits<T>::round_acc(w * xv[j]);
  }
}

// Accumulate window entries [lo, hi) (window-relative, 0..64) of one row.  `wrel` is the window's
// offset from the partition's first edge (min/max args are kept as 32-bit offsets).
// c_l / w_l hold the window's column ids / weights, one per lane.  All lanes
// stay active; slots past `hi` re-read the last valid entry (masked for sums, harmless for min / max).
// MASKED (sums only): e_l holds the window's entry ids; feature j of this lane's packet contributes iff bit
// (mask_shift + j) of maskk[entry * mask_words] is set, and the product is rounded to the element type
// before it is added (what value.index_select(0, arg) * grad_out does in SPMMMin/Max::backward).
template <typename T, int VEC, int RED, bool MASKED = false>
__device__ __forceinline__ void accumulate_window(
    int lo, int hi, uint32_t wrel, uint32_t c_l, typename Traits<T>::acc_t w_l, bool has_value,
    const T *__restrict__ matk, uint32_t K, int lgG, int g,
    typename Traits<T>::acc_t (&val)[VEC], uint32_t (&arg)[VEC], uint32_t e_l = 0,
    const uint32_t *__restrict__ maskk = nullptr, uint32_t mask_words = 0, uint32_t mask_shift = 0,
    uint32_t z_l = 0xFFFFFFFFu, uint32_t mask_seg = 0) {
  using A = typename Traits<T>::acc_t;
  using P = Pack<T, VEC>;
  // min/max carry (value, arg) per element: fewer gathers in flight keep the VGPR count down
  constexpr int kU = RED == RED_ADD ? kUnroll : kMinMaxUnroll;
  const int n = hi - lo;
  const int nsteps = (n + (1 << lgG) - 1) >> lgG;
#if TSAMD_MINMAX_SPLIT_LOOP
  if constexpr (RED != RED_ADD) {
    // min / max: one step loop PER value mode (the wave-uniform `has_value` test sits outside the loop).  The
    // 2-byte instantiations are bound by VALU issue (SQ counters, round 3: ~80 % of the slots at config 3); with the
    // branch inside the loop the two paths left their results in different registers (6 v_mov at every back edge),
    // both fetched the window's weights (2 ds_bpermute the value-less path never reads), and the gather address took
    // a multiply + a shift-add (now one v_mad_u64_u32 on byte units).  Slots past `hi` re-read the row's last entry:
    // min / max are idempotent, the duplicate carries the same (value, edge id) as the original, so nothing has to be
    // masked (strict compares: an equal candidate with a larger or equal id never replaces).  Without values the
    // candidate is the stored element itself: no product, no rounding.
    const char *matb = reinterpret_cast<const char *>(matk);
    const uint32_t kbytes = K * (uint32_t)sizeof(T);
    auto run = [&](auto with_value) __attribute__((always_inline)) {
      constexpr bool kWV = decltype(with_value)::value;
      int pos = lo + g;
      for (int s = 0; s < nsteps; s += kU) {
        P x[kU];
        A w[kU];
        uint32_t id[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int at = pos + (u << lgG);
          const int src = at < hi ? at : hi - 1;
          id[u] = wrel + (uint32_t)src;
          const uint32_t c = lane_read(c_l, src);
          if constexpr (kWV) w[u] = lane_read(w_l, src);
          x[u] = *reinterpret_cast<const P *>(matb + (uint64_t)c * kbytes);
        }
        pos += kU << lgG;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          A xv[VEC];
          unpack_packet<T, VEC>(x[u], xv);
          if constexpr (kWV) round_products<T, VEC>(w[u], xv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const A p = xv[j];
            const bool better = RED == RED_MIN ? (p < val[j]) : (p > val[j]);
            val[j] = better ? p : val[j];
            arg[j] = better ? id[u] : arg[j];
          }
        }
      }
    };
    if (has_value) run(std::true_type{});  // wave-uniform
    else run(std::false_type{});
    return;
  }
#endif
  for (int s = 0; s < nsteps; s += kU) {
    P x[kU];
    A w[kU];
    int idx[kU];
    uint32_t mb[MASKED ? kU : 1];
    [[maybe_unused]] uint32_t cm[MASKED ? kU : 1], em[MASKED ? kU : 1], on[MASKED ? kU : 1];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      idx[u] = lo + ((s + u) << lgG) + g;
      const int src = idx[u] < hi ? idx[u] : hi - 1;
      if constexpr (RED != RED_ADD) idx[u] = src;  // see below: min / max need no mask
      const uint32_t c = lane_read(c_l, src);
      w[u] = lane_read(w_l, src);
      if constexpr (MASKED) {
#if TSAMD_MASKED_SEGMENT_SKIP
        cm[u] = c;
        em[u] = lane_read(e_l, src);
        on[u] = (lane_read(z_l, src) >> mask_seg) & 1u;
#else
        x[u] = *reinterpret_cast<const P *>(matk + (uint64_t)c * K);
        mb[u] = maskk[(uint64_t)lane_read(e_l, src) * mask_words];
#endif
      } else {
        x[u] = *reinterpret_cast<const P *>(matk + (uint64_t)c * K);
      }
    }
#if TSAMD_MASKED_SEGMENT_SKIP
    if constexpr (MASKED) {
      // winners are sparse in the entries of long rows (an entry of a row of degree d wins a feature with
      // probability ~1/d): the record says which of the row's 32-feature segments have one at all, and only
      // those are gathered (the lanes of an empty segment sit the load out; their mask word reads as zero).
      // All cross-lane reads of the step come first, then the loads: one LDS-pipe wait per step, not two per gather.
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        asm volatile("" : "+v"(cm[u]), "+v"(em[u]), "+v"(on[u]));
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        x[u] = P{};
        mb[u] = 0u;
        if (on[u] != 0u) {
          x[u] = *reinterpret_cast<const P *>(matk + (uint64_t)cm[u] * K);
          mb[u] = maskk[(uint64_t)em[u] * mask_words];
        }
      }
    }
#endif
    if constexpr (RED == RED_ADD && MASKED) {
      // the masked sum is bound by instruction issue as much as by its gathers (twice the instructions of the
      // plain sum per row): the packet's predicate bits become all-ones / all-zero words (v_bfe_i32) that are
      // ANDed onto the addend, and the value-less case skips the multiply and the rounding altogether
      auto add_masked = [&](auto with_value) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
#if TSAMD_MASKED_SEGMENT_SKIP
          // no lane of the wave gathered anything for this slot (hub-row entries mostly win nothing): nothing to add
          if (__ballot(on[u] != 0u) == 0ull) continue;  // wave-uniform
#endif
          const uint32_t bits = idx[u] < hi ? (mb[u] >> mask_shift) : 0u;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const A xv = Traits<T>::to_acc(x[u].v[j]);
            A p = xv;
            if constexpr (decltype(with_value)::value) p = Traits<T>::round_acc(w[u] * xv);
            const int32_t m = __builtin_amdgcn_sbfe((int32_t)bits, j, 1);  // 0 or -1
            if constexpr (sizeof(A) == 4) {
              uint32_t pb;
              __builtin_memcpy(&pb, &p, 4);
              pb &= (uint32_t)m;
              A pm;
              __builtin_memcpy(&pm, &pb, 4);
              val[j] += pm;
            } else {
              uint64_t pb;
              __builtin_memcpy(&pb, &p, 8);
              pb &= (uint64_t)(int64_t)m;
              A pm;
              __builtin_memcpy(&pm, &pb, 8);
              val[j] += pm;
            }
          }
        }
      };
      if (has_value) add_masked(std::true_type{});  // wave-uniform
      else add_masked(std::false_type{});
      continue;
    }
    if constexpr (RED != RED_ADD) {
      // Slots past `hi` re-read the row's last entry: min / max are idempotent, the duplicate carries the same
      // (value, edge id) as the original, so nothing has to be masked (strict compares: an equal candidate with
      // a larger or equal id never replaces).  Without values the candidate is the stored element itself: no
      // product, no rounding -- as a wave-uniform branch, not a select: the 2-byte instantiations are bound by
      // VALU issue (84 % of the pipe at config 3, SQ counters), and `has_value ? round(w * x) : x` per element
      // was a third of their instructions.
      auto fold = [&](auto with_value) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const A xv = Traits<T>::to_acc(x[u].v[j]);
            A p = xv;
            if constexpr (decltype(with_value)::value) p = Traits<T>::round_acc(w[u] * xv);
            const bool better = RED == RED_MIN ? (p < val[j]) : (p > val[j]);
            val[j] = better ? p : val[j];
#if !defined(TSAMD_EXP_NO_ARG_TRACK)
            arg[j] = better ? wrel + (uint32_t)idx[u] : arg[j];
#endif
          }
        }
      };
      if (has_value) fold(std::true_type{});  // wave-uniform
      else fold(std::false_type{});
      continue;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool ok = idx[u] < hi;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const A xv = Traits<T>::to_acc(x[u].v[j]);
        if constexpr (RED == RED_ADD && MASKED) {
          const A p = Traits<T>::round_acc(w[u] * xv);
          val[j] += (ok && ((mb[u] >> (mask_shift + (uint32_t)j)) & 1u)) ? p : A(0);
        } else {
          const A p = w[u] * xv;
          val[j] += ok ? p : A(0);
        }
      }
    }
  }
}

// Reduce over the G lane groups towards group 0 (the only one that writes): log2(G) levels, each
// combining a lane with lane + off.  The exchanges are VALU-only (lane_down: DPP / permlane swaps),
// so a row end costs no LDS-pipe round trips; the pairs combined at every level are the ones a
// butterfly would combine, i.e. group 0 ends up with bit-identical results.
template <int OFF, typename A, int VEC, int RED, typename ARG>
__device__ __forceinline__ void reduce_level(A (&val)[VEC], ARG (&arg)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const A o = lane_down<OFF>(val[j]);
    if constexpr (RED == RED_ADD) {
      val[j] += o;
    } else {
      const ARG oa = lane_down<OFF>(arg[j]);
      // bitwise, not short-circuit: as `||` / `&&` this became four exec-mask branches per element (~22
      // instructions; a row end of the 2-byte min / max kernels spent ~90 of its ~110 instructions here)
      const bool better = RED == RED_MIN ? (o < val[j]) : (o > val[j]);
      const bool take = better | ((o == val[j]) & (oa < arg[j]));
      val[j] = take ? o : val[j];
      arg[j] = take ? oa : arg[j];
    }
  }
}

template <typename A, int VEC, int RED, typename ARG>
__device__ __forceinline__ void reduce_groups(int lgG, A (&val)[VEC], ARG (&arg)[VEC]) {
  if (lgG >= 1) reduce_level<32, A, VEC, RED, ARG>(val, arg);
  if (lgG >= 2) reduce_level<16, A, VEC, RED, ARG>(val, arg);
  if (lgG >= 3) reduce_level<8, A, VEC, RED, ARG>(val, arg);
  if (lgG >= 4) reduce_level<4, A, VEC, RED, ARG>(val, arg);
  if (lgG >= 5) reduce_level<2, A, VEC, RED, ARG>(val, arg);
  if (lgG >= 6) reduce_level<1, A, VEC, RED, ARG>(val, arg);
}

// Non-temporal store of a packet (any size that is a multiple of 4 bytes goes out as dwords).
template <typename U, int VEC>
__device__ __forceinline__ void nt_store(U *dst, const Pack<U, VEC> &v) {
  constexpr int kBytes = (int)sizeof(Pack<U, VEC>);
  if constexpr (kBytes % 16 == 0) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < kBytes / 16; ++i)
      __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(&v)[i], reinterpret_cast<u32x4 *>(dst) + i);
  } else if constexpr (kBytes == 8) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(*reinterpret_cast<const u32x2 *>(&v), reinterpret_cast<u32x2 *>(dst));
  } else if constexpr (kBytes == 4) {
    __builtin_nontemporal_store(*reinterpret_cast<const unsigned int *>(&v), reinterpret_cast<unsigned int *>(dst));
  } else {
    *reinterpret_cast<Pack<U, VEC> *>(dst) = v;
  }
}

// Final write of one row (reducer.h:69-83).
// min / max row of a column-block partial product (Workspace::partial): the candidate (val, arg) of this block
// against the (value, arg) the earlier blocks left in out / arg_out.  State between blocks: arg == arg_none means
// "no winner so far", with value 0 (no entry seen yet) or the reduction's init value (entries seen, none beat it).
template <typename T, int VEC, int RED>
__device__ __forceinline__ void write_row_partial(T *__restrict__ outk, int64_t *__restrict__ argk,
                                                  typename Traits<T>::acc_t (&val)[VEC], int64_t (&arg)[VEC],
                                                  int64_t deg, const Workspace &ws) {
  using A = typename Traits<T>::acc_t;
  Pack<T, VEC> o;
  Pack<int64_t, VEC> a;
  if (ws.accumulate) {
    if (deg <= 0) return;  // the block has no entry in this row: the earlier blocks' result stands
    o = *reinterpret_cast<const Pack<T, VEC> *>(outk);
    a = *reinterpret_cast<const Pack<int64_t, VEC> *>(argk);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    int64_t ca = arg[j];
    if (ca == kNoArg) ca = ws.arg_none;
    else if (ws.arg_map != nullptr) ca = ws.arg_map[ca];
    if (!ws.accumulate) {
      o.v[j] = Traits<T>::from_acc(deg > 0 ? val[j] : A(0));
      a.v[j] = deg > 0 ? ca : ws.arg_none;
    } else {
      const A ev = Traits<T>::to_acc(o.v[j]);
      const int64_t ea = a.v[j];
      bool take;
      if (ea == ws.arg_none) take = true;        // nothing won so far: (val, ca) -- or (init, none) -- stands
      else if (ca == ws.arg_none) take = false;  // this block brought no winner
      else take = (RED == RED_MIN ? (val[j] < ev) : (val[j] > ev)) || (val[j] == ev && ca < ea);
      if (take) {
        o.v[j] = Traits<T>::from_acc(val[j]);
        a.v[j] = ca;
      }
    }
  }
  *reinterpret_cast<Pack<T, VEC> *>(outk) = o;  // plain stores: the next block reads them back
  *reinterpret_cast<Pack<int64_t, VEC> *>(argk) = a;
}

// A32: the ids go out as int32 (tsamd_spmm_minmax_arg32) -- a compile-time variant: as a run-time branch the second
// packet of ids spilled the fp32 min / max kernel (63 VGPRs at 8 waves per SIMD)
template <typename T, int VEC, int RED, bool A32 = false, bool STORE_ARG = true>
__device__ __forceinline__ void write_row(T *__restrict__ out_base, int64_t *__restrict__ arg_base, uint64_t arg_off,
                                          typename Traits<T>::acc_t (&val)[VEC],
                                          int64_t (&arg)[VEC], int64_t deg, bool mean,
                                          int64_t E, const Workspace &ws) {
  using A = typename Traits<T>::acc_t;
  Pack<T, VEC> o;
  constexpr bool a32 = A32 && !kPartial && RED != RED_ADD;
  int64_t *argk = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(arg_base) + (arg_off << (a32 ? 2 : 3)));
  T *outk = out_base + arg_off;
  if constexpr (kPartial && RED != RED_ADD) {
    write_row_partial<T, VEC, RED>(outk, argk, val, arg, deg, ws);
    return;
  }
  if constexpr (RED == RED_ADD) {
    if (kPartial && ws.accumulate) {  // wave-uniform: the earlier column blocks' sum
      const Pack<T, VEC> ex = *reinterpret_cast<const Pack<T, VEC> *>(outk);
#pragma unroll
      for (int j = 0; j < VEC; ++j) val[j] += Traits<T>::to_acc(ex.v[j]);
    }
    if (mean) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) val[j] = mean_of<T>(val[j], deg);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      asm volatile("" : "+v"(val[j]));  // keep the mean / non-mean paths from splitting the 16-byte store
      o.v[j] = Traits<T>::from_acc(val[j]);
    }
#if !defined(TSAMD_NO_NT_STORE)
    // output rows are written once and never re-read by this kernel: a non-temporal store keeps
    // them from evicting gathered rows of `mat` from L2 (+1.5-2 % measured)
    if constexpr (sizeof(Pack<T, VEC>) == 16) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(*reinterpret_cast<u32x4 *>(&o), reinterpret_cast<u32x4 *>(outk));
    } else {
      *reinterpret_cast<Pack<T, VEC> *>(outk) = o;
    }
#else
    *reinterpret_cast<Pack<T, VEC> *>(outk) = o;
#endif
  } else {
    Pack<int64_t, VEC> a;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (deg > 0) {
        o.v[j] = Traits<T>::from_acc(val[j]);
        // no entry beat the init value (NaN-only / +-max inputs): the reference
        // leaves a stale index here; we report E ("no winner").
        if constexpr (STORE_ARG) a.v[j] = arg[j] == kNoArg ? E : arg[j];
      } else {
        o.v[j] = Traits<T>::from_acc(A(0));
        if constexpr (STORE_ARG) a.v[j] = E;
      }
    }
    // written once, never re-read here: keep them out of L2 (1.3 GB of arg ids at config-3 size
    // would otherwise evict the gathered rows of `mat`)
#if !defined(TSAMD_EXP_NO_OUT_STORE)
    nt_store(outk, o);
#endif
#if !defined(TSAMD_EXP_NO_ARG_STORE)
    if constexpr (!STORE_ARG) {  // the caller keeps the winners in another form (Workspace::rec_out)
    } else if constexpr (a32) {
      Pack<int32_t, VEC> an;
#pragma unroll
      for (int j = 0; j < VEC; ++j) an.v[j] = (int32_t)a.v[j];
      nt_store(reinterpret_cast<int32_t *>(argk), an);
    } else {
      nt_store(argk, a);
    }
#else
    asm volatile("" ::"v"(a.v[0]), "v"(a.v[VEC - 1]));
#endif
  }
}

template <typename T, int VEC, int RED>
__device__ __forceinline__ void write_carry(void *cval, uint32_t *carg, uint64_t off,
                                            typename Traits<T>::acc_t (&val)[VEC],
                                            uint32_t (&arg)[VEC]) {
  using A = typename Traits<T>::acc_t;
  Pack<A, VEC> v;
#pragma unroll
  for (int j = 0; j < VEC; ++j) v.v[j] = val[j];
  *reinterpret_cast<Pack<A, VEC> *>(reinterpret_cast<A *>(cval) + off) = v;
  if constexpr (RED != RED_ADD) {
    Pack<uint32_t, VEC> a;
#pragma unroll
    for (int j = 0; j < VEC; ++j) a.v[j] = arg[j];
    *reinterpret_cast<Pack<uint32_t, VEC> *>(carg + off) = a;
  }
}

// ---------------------------------------------------------------------------
// 2. main kernel: wave p consumes merge-path items [table[p], table[p+1])
//    grid = (ceil(P / waves per block), B * ktiles)
// ---------------------------------------------------------------------------
// SHORT: instantiate the "short rows side by side" path (launched for rows of <= 128 bytes only: its
// registers would cost the wide-row instantiation two waves per SIMD)
// The min / max instantiations for wide rows need 61-63 VGPRs but 106 SGPRs, one granule more than fits
// 8 waves per SIMD; asking for 8 makes the allocator fit (experiment knob: -DTSAMD_MINMAX_WAVES=0 turns it off).
#ifndef TSAMD_MINMAX_WAVES
#define TSAMD_MINMAX_WAVES 8
#endif
// (partial build: the sum's row sink costs 6 SGPRs -- ask for 8 waves there; the min / max sink needs the VGPRs)
// ---- record-writing forward: the steps the merge and the fix-up kernel share (Workspace::rec_out) ------------------------
// A wave's LDS tile holds the records of up to 64 consecutive entries of one row exactly as they lie in memory (S words
// each: W mask words, row id, value, [segment bitmap], padding): cleared, the winners' bits entered by `ds_or`, completed
// by the entry's own lane, then copied out as one contiguous block in 16-byte packets.
constexpr int kRecTileWords = kWave * 12;  // K <= 256: records of at most 12 words

__device__ __forceinline__ void records_clear(uint32_t *tile, int lane, uint32_t nq, uint32_t S) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  for (uint32_t w = (uint32_t)lane * 4u; w < nq * S; w += (uint32_t)kWave * 4u)
    *reinterpret_cast<u32x4 *>(tile + w) = u32x4{0u, 0u, 0u, 0u};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// entries first .. first + nq - 1 (ids inside the matrix) of row r; rec_b = the batch's records
template <typename T>
__device__ __forceinline__ void records_finish(uint32_t *tile, int lane, uint32_t nq, int64_t first, uint32_t r,
                                               const T *value, uint32_t *rec_b, uint32_t W, uint32_t S, bool has_z) {
  using A = typename Traits<T>::acc_t;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if ((uint32_t)lane < nq) {
    uint32_t *slot = tile + (uint32_t)lane * S;
    // (words past W are still zero here: the slot was cleared and only mask words have been set)
    const u32x4 m0 = *reinterpret_cast<const u32x4 *>(slot);
    const u32x4 m1 = W > 4u ? *reinterpret_cast<const u32x4 *>(slot + 4) : u32x4{0u, 0u, 0u, 0u};
    const uint32_t z = (m0.x != 0u ? 1u : 0u) | (m0.y != 0u ? 2u : 0u) | (m0.z != 0u ? 4u : 0u) | (m0.w != 0u ? 8u : 0u) |
                       (m1.x != 0u ? 16u : 0u) | (m1.y != 0u ? 32u : 0u) | (m1.z != 0u ? 64u : 0u) | (m1.w != 0u ? 128u : 0u);
    A wv = A(1);
    if (value != nullptr) wv = Traits<T>::to_acc(value[first + lane]);
    uint32_t wbits;
    __builtin_memcpy(&wbits, &wv, 4);
    slot[W] = r;
    slot[W + 1u] = wbits;
    if (has_z) slot[W + 3u] = z;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  u32x4 *dst = reinterpret_cast<u32x4 *>(rec_b + (uint64_t)first * S);
  const uint32_t npk = nq * S / 4u;
  for (uint32_t pk = (uint32_t)lane; pk < npk; pk += (uint32_t)kWave) dst[pk] = *reinterpret_cast<const u32x4 *>(tile + pk * 4u);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// a record without winners (pieces of long cut rows): S / 4 packets, the row id and the value in their places
__device__ __forceinline__ void record_blank(uint32_t *dst, uint32_t r, uint32_t wbits, uint32_t W, uint32_t S) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  for (uint32_t j = 0; j < S; j += 4u) {
    u32x4 pk;
    pk.x = j == W ? r : (j == W + 1u ? wbits : 0u);
    pk.y = j + 1u == W ? r : (j == W ? wbits : 0u);
    pk.z = j + 2u == W ? r : (j + 1u == W ? wbits : 0u);
    pk.w = j + 3u == W ? r : (j + 2u == W ? wbits : 0u);
    *reinterpret_cast<u32x4 *>(dst + j) = pk;
  }
}

// cut rows of at most this many entries get their whole records from the fix-up wave (64 entries per step through an LDS
// tile); in longer ones it only enters the winners into the records the merge kernel left without any
#ifndef TSAMD_RECORD_SNAP
#define TSAMD_RECORD_SNAP 128
#endif
#ifndef TSAMD_FIXUP_RECORD_MAX
#define TSAMD_FIXUP_RECORD_MAX 1024
#endif
constexpr int64_t kFixupRecordMax = TSAMD_FIXUP_RECORD_MAX;

#ifndef TSAMD_RECORD_WAVES
#define TSAMD_RECORD_WAVES 7
#endif
template <int RED, bool SHORT, bool MASKED, int REC = 0>
constexpr int kMinWavesPerEU = kPartial ? ((RED == RED_ADD && !SHORT && !MASKED) ? 8 : 0)
                                        : ((RED != RED_ADD && !SHORT && !MASKED) ? (REC == 1 ? TSAMD_RECORD_WAVES : (REC == 2 ? TSAMD_RECORD_WAVES - 1 : TSAMD_MINMAX_WAVES)) : 0);

// REC = 1 | 2 (with A32): the kernel writes the winner records of the rows it finishes instead of their ids
// (Workspace::rec_out); 1 = the 32-byte records of 97..128 features (one 16-byte mask per entry in the LDS tile, two direct
// stores per lane), 2 = any record shape up to 256 features (the tile holds whole records, copied out in packets: 0.11 ms
// slower at 128 features -- more live registers, more LDS traffic -- which is why the common case keeps its own code)
// -- its own instantiation: as a run-time branch of the A32 kernel the record writer cost that kernel 52 bytes of
// scratch per lane (it sits at its register limit), here the int64 ids of the row store are gone instead
template <typename T, int VEC, int RED, bool SHORT, bool MASKED = false, bool A32 = false, int REC = 0>
__global__ __launch_bounds__(kWavesPerBlock *kWave, (kMinWavesPerEU<RED, SHORT, MASKED, REC>)) void spmm_merge_kernel(
    const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
    const T *__restrict__ value, const T *__restrict__ mat, T *__restrict__ out,
    int64_t *__restrict__ arg_out, int64_t M, int64_t N, uint32_t K, int64_t E,
    uint32_t ktiles, int lgG, bool mean, Workspace ws) {
  using A = typename Traits<T>::acc_t;
  // (see Workspace::rec_out) one feature tile of four-element packets: the host asks for records only when K <= 256
  constexpr bool kEmitRecords = REC != 0 && A32 && RED != RED_ADD && !MASKED && !SHORT && VEC == 4 && sizeof(A) == 4 && !kPartial;
  constexpr bool kRec32 = REC == 1;  // 32-byte records (97..128 features): straight from the tile's 16-byte masks
  static_assert(REC == 0 || kEmitRecords, "record-writing merge kernel: int32 ids, min / max, four-element packets, 4-byte accumulators");
  __shared__ alignas(16) uint32_t rec_tile_[REC == 1 ? kWavesPerBlock * kWave * 4 : (REC == 2 ? kWavesPerBlock * kRecTileWords : 4)];
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t p = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  if (p >= ws.P) return;
  const uint32_t y = blockIdx.y;
  const uint32_t b = y / ktiles;
  const uint32_t kt = y - b * ktiles;

  const Coord c0 = ws.table[p];
  const Coord c1 = ws.table[p + 1];
  const int64_t r0 = c0.row, e0 = c0.edge, r1 = c1.row, e1 = c1.edge;

  const int lpr = 64 >> lgG;
  const int g = lane >> (6 - lgG);
  const int kl = lane & (lpr - 1);
  const uint32_t k0 = (kt * 64u + (uint32_t)kl) * VEC;
  const bool kok = k0 < K;
  const bool relabel = use_relabel(ws.relabel_mode, ws.relabel_flag);  // wave-uniform
  const T *src = relabel ? reinterpret_cast<const T *>(ws.xperm) : mat;
  const T *matk = src + (uint64_t)b * N * K + (kok ? k0 : 0u);
  const uint64_t out_b = (uint64_t)b * M * K + k0;
  const bool writer = g == 0 && kok;
  const uint64_t carry_off = ((uint64_t)b * ws.P + (uint64_t)p) * K + k0;  // [b][p][K]
  // masked sums: this lane's VEC features sit in one 32-bit word of every entry's mask (VEC divides 32)
  const uint32_t *maskk = nullptr;
  uint32_t mask_shift = 0, mask_seg = 0;
  if constexpr (MASKED) {
    maskk = ws.wmask + (uint64_t)b * (uint64_t)E * ws.rec_stride + (kok ? (k0 >> 5) : 0u);
    mask_shift = kok ? (k0 & 31u) : 0u;
    mask_seg = kok ? (k0 >> 5) : 0u;
  }

  // the first row may have been started by an earlier partition
  const bool incoming = r0 < M && e0 > rowptr[r0];
  // record-writing forward: are the partition's first / last row long ones (blank_records below)?  Asked for HERE: at the
  // end of the partition the two dependent scalar loads were a round trip on every wave's critical path (+0.09 ms)
  int rec_long = 0;
  if constexpr (kEmitRecords) {
    const int64_t ra = r0 < M ? r0 : M - 1, rb = r1 < M ? r1 : M - 1;
    const int64_t da = rowptr[ra + 1] - rowptr[ra], db = rowptr[rb + 1] - rowptr[rb];
    rec_long = (da > kFixupRecordMax ? 1 : 0) | (db > kFixupRecordMax ? 2 : 0);
    asm volatile("" : "+v"(rec_long));
  }

  // (col, value) windows: [wbase, wbase+64) current, the next one in flight
  int64_t wbase = e0;
  uint32_t c_cur, c_nxt;
  A w_cur, w_nxt;
  uint32_t e_cur = 0, e_nxt = 0;  // MASKED only: the entries' ids (index of their mask)
  uint32_t z_cur = 0xFFFFFFFFu, z_nxt = 0xFFFFFFFFu;  // MASKED only: which mask words of the entry are non-zero
  auto load_window = [&](int64_t base, uint32_t &c_l, A &w_l, uint32_t &e_l, uint32_t &z_l) {
    const int64_t e = base + lane;
    c_l = 0;
    w_l = A(1);
    if constexpr (MASKED) {
      e_l = 0;
      z_l = 0xFFFFFFFFu;
    }
    if (e < e1) {
      const int64_t src_e = ws.perm != nullptr ? ws.perm[e] : e;  // windows are fetched two ahead:
      if constexpr (MASKED) {                                      // the indirection is off the critical path
        // column id and value come from the entry's record: the line the mask gathers will hit again
        const uint32_t *rec = ws.wmask + ((uint64_t)b * (uint64_t)E + (uint64_t)src_e) * ws.rec_stride + ws.rec_meta;
        c_l = rec[0];
        if (value != nullptr) {
          if constexpr (sizeof(A) == 8) {
            const uint64_t bits = (uint64_t)rec[1] | ((uint64_t)rec[2] << 32);
            __builtin_memcpy(&w_l, &bits, 8);
          } else {
            const uint32_t bits = rec[1];
            __builtin_memcpy(&w_l, &bits, 4);
          }
        }
        e_l = (uint32_t)src_e;
        if (ws.rec_has_z) z_l = rec[3];
      } else {
        c_l = (uint32_t)col[src_e];
        if (value != nullptr) w_l = Traits<T>::to_acc(value[src_e]);
      }
      if (relabel) c_l = hash_row(c_l, (uint32_t)N, ws.hash_bits, ws.hash_mul, ws.hash_shift);
    }
  };
  load_window(wbase, c_cur, w_cur, e_cur, z_cur);
  load_window(wbase + kWave, c_nxt, w_nxt, e_nxt, z_nxt);

  // row ends: lane j holds rowptr[rp_base + 1 + j]
  int64_t rp_base = r0;
  auto load_rowends = [&](int64_t base) -> int64_t {
    const int64_t r = base + 1 + lane;
    int64_t v = rowptr[r <= M ? r : M];
    // Consume the value here: otherwise the compiler keeps it "pending" across the row loop and
    // puts an s_waitcnt vmcnt(0) at the top of EVERY row iteration (draining the previous row's
    // store and the prefetched window) instead of once per 64 rows.
    uint32_t lo = (uint32_t)(uint64_t)v, hi = (uint32_t)((uint64_t)v >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (int64_t)(((uint64_t)hi << 32) | lo);
  };
  int64_t rp_l = load_rowends(rp_base);

  int64_t e = e0;
  A val[VEC];
  uint32_t arg[VEC];    // offsets from e0
  int64_t arg64[VEC];   // absolute edge ids, only materialised when a row is written
  init_acc<T, VEC, RED>(val, arg);
  const bool has_value = value != nullptr;
  auto widen_args = [&]() {
#pragma unroll
    for (int j = 0; j < VEC; ++j) arg64[j] = arg[j] == kNoArg32 ? kNoArg : e0 + (int64_t)arg[j];
  };

  // Windows alternate between two register sets (A = c_cur/w_cur, B = c_nxt/w_nxt): window k is
  // consumed from one set while window k+1 is already in flight into the other, and the set just
  // consumed is refilled with window k+2.  No register is ever renamed, so the compiler waits
  // for outstanding memory operations once per window -- not once per row, which would also
  // drain the previous row's store (measured ~10 % on short-row graphs).
  int64_t r = r0;
  int64_t estart = e0;  // first edge of the current row that belongs to this partition
  int64_t trow = -1;
  // ---- short rows side by side (narrow feature matrices) ------------------------------------
  // With G >= 8 lane groups (rows of <= 128 bytes) a row of ~20 entries fills one partly used batch
  // of gathers, and the wave pays one global-memory round trip per ROW (measured: F = 4 / 8 / 16 all
  // take ~0.75 ms on the north-star graph).  When the next rows are all short, up to G of them are
  // therefore processed at once, one row per lane group: the group's lanes fetch lpr consecutive
  // entries of their row (prefetched one step ahead), every lane of the group consumes them, and
  // each group writes its own row -- no reduction across groups, ~len / lpr round trips for G rows.
  // Rows longer than kShortFactor * lpr entries, the row that was started by an earlier partition
  // and the unfinished last row keep the cooperative path below.
  constexpr int kShortFactor = 8;
  auto short_rows = [&]() __attribute__((always_inline)) -> bool {
    if constexpr (!SHORT) return false;
    if (lgG < 3 || r >= r1) return false;
    int j = (int)(r - rp_base);
    if (j >= kWave) {
      rp_base = r;
      rp_l = load_rowends(rp_base);
      j = 0;
    }
    const int G = 1 << lgG;
    int navail = (int)(r1 - r < (int64_t)G ? r1 - r : (int64_t)G);
    if (navail > kWave - j) navail = kWave - j;
    if (navail < 2) return false;
    const uint32_t rel_l = (uint32_t)(uint64_t)(rp_l - e0);  // row ends relative to e0 (rows < r1: < 2^31)
    const uint32_t end_g = lane_read(rel_l, j + (g < navail ? g : navail - 1));
    const uint32_t prev_g = lane_read(rel_l, j + (g > 0 ? (g <= navail ? g - 1 : navail - 1) : 0));
    const uint32_t beg_g = g == 0 ? (uint32_t)(e - e0) : prev_g;
    bool mine = g < navail;
    const uint32_t len = mine ? end_g - beg_g : 0u;
    const unsigned long long too_long = __ballot(mine && len > (uint32_t)(kShortFactor * lpr));
    const int n = too_long ? (int)(__builtin_ctzll(too_long) >> (6 - lgG)) : navail;
    if (n < 2) return false;
    mine = g < n;
    const uint32_t stop_g = mine ? end_g : beg_g;
    const int grp0 = lane & ~(lpr - 1);
    auto fetch = [&](uint32_t q, uint32_t &c_l, A &w_l) {
      c_l = 0;
      w_l = A(1);
      if (q < stop_g) {
        const int64_t src_e = ws.perm != nullptr ? ws.perm[e0 + q] : e0 + (int64_t)q;
        c_l = (uint32_t)col[src_e];
        if (relabel) c_l = hash_row(c_l, (uint32_t)N, ws.hash_bits, ws.hash_mul, ws.hash_shift);
        if (value != nullptr) w_l = Traits<T>::to_acc(value[src_e]);
      }
    };
    uint32_t pos = beg_g;
    uint32_t c_l, c_n;
    A w_l, w_n;
    fetch(pos + (uint32_t)kl, c_l, w_l);
    while (__any(pos < stop_g)) {
      fetch(pos + (uint32_t)lpr + (uint32_t)kl, c_n, w_n);  // the next step's entries are on their way
#pragma unroll
      for (int u0 = 0; u0 < 8; u0 += 4) {
        if (u0 >= lpr) break;  // wave-uniform
        Pack<T, VEC> x[4];
        A w[4];
        bool ok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int u = u0 + t;
          const int srcl = grp0 + (u < lpr ? u : 0);
          const uint32_t c = lane_read(c_l, srcl);
          w[t] = lane_read(w_l, srcl);
          ok[t] = u < lpr && pos + (uint32_t)u < stop_g;
          x[t] = *reinterpret_cast<const Pack<T, VEC> *>(matk + (uint64_t)(ok[t] ? c : 0u) * K);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int jj = 0; jj < VEC; ++jj) {
            const A xv = Traits<T>::to_acc(x[t].v[jj]);
            if constexpr (RED == RED_ADD) {
              const A pr = w[t] * xv;
              val[jj] += ok[t] ? pr : A(0);
            } else {
              const A pr = has_value ? Traits<T>::round_acc(w[t] * xv) : xv;
              const bool better = ok[t] && (RED == RED_MIN ? (pr < val[jj]) : (pr > val[jj]));
              val[jj] = better ? pr : val[jj];
              arg[jj] = better ? pos + (uint32_t)(u0 + t) : arg[jj];
            }
          }
        }
      }
      pos += (uint32_t)lpr;
      c_l = c_n;
      w_l = w_n;
    }
    if (mine && kok) {
      if constexpr (RED != RED_ADD) widen_args();
      const uint64_t o = out_b + out_position(ws, r + g, M) * K;
      int64_t deg_w = (int64_t)len;
      if constexpr (kPartial && RED == RED_ADD) {
        if (mean && ws.deg_rowptr != nullptr) deg_w = ws.deg_rowptr[r + g + 1] - ws.deg_rowptr[r + g];
      }
      write_row<T, VEC, RED, A32>(out, arg_out, o, val, arg64, deg_w, mean, E, ws);
    }
    init_acc<T, VEC, RED>(val, arg);
    const uint32_t done_rel = (uint32_t)__builtin_amdgcn_readlane((int)rel_l, j + n - 1);
    r += n;
    e = e0 + (int64_t)done_rel;
    estart = e;
    return true;
  };
  // as many batches of short rows as there are; then the cooperative path continues behind them:
  // refill both window register sets
  auto short_row_batches = [&]() __attribute__((always_inline)) -> bool {
    if (!short_rows()) return false;
    while (short_rows()) {
    }
    wbase = e;
    load_window(wbase, c_cur, w_cur, e_cur, z_cur);
    load_window(wbase + kWave, c_nxt, w_nxt, e_nxt, z_nxt);
    return true;
  };

  // record-writing forward: the entries [from, to) of a piece of the cut row `rr` get records WITHOUT winners (row id,
  // value, empty masks); the fix-up kernel, which learns the row's winners, sets them in the few records that have any
  auto blank_records = [&](int64_t from, int64_t to, int64_t rr) __attribute__((always_inline)) {
    if constexpr (kEmitRecords) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      if (!(rec_long & (rr == r0 ? 1 : 2))) return;  // a short row: the fix-up wave writes its records whole
      if constexpr (kRec32) {
        for (int64_t qb = from + lane; qb < to; qb += kWave) {
          A wv = A(1);
          if (has_value) wv = Traits<T>::to_acc(value[qb]);
          uint32_t wbits;
          __builtin_memcpy(&wbits, &wv, 4);
          u32x4 *dst = reinterpret_cast<u32x4 *>(ws.rec_out + ((uint64_t)b * (uint64_t)E + (uint64_t)qb) * 8u);
          dst[0] = u32x4{0u, 0u, 0u, 0u};
          dst[1] = u32x4{(uint32_t)rr, wbits, 0u, 0u};
        }
      } else {
        for (int64_t qb = from + lane; qb < to; qb += kWave) {
          A wv = A(1);
          if (has_value) wv = Traits<T>::to_acc(value[qb]);
          uint32_t wbits;
          __builtin_memcpy(&wbits, &wv, 4);
          record_blank(ws.rec_out + ((uint64_t)b * (uint64_t)E + (uint64_t)qb) * ws.rec_stride, (uint32_t)rr, wbits, ws.rec_meta,
                       ws.rec_stride);
        }
      }
    }
  };

  // rows (or row pieces) inside the window [wbase, wbase + 64): 0 = window exhausted, 1 = partition done,
  // 2 = a batch of short rows was processed side by side and the windows were re-based (start over)
  auto process_window = [&](const uint32_t c_w, const A w_w, const uint32_t e_w, const uint32_t z_w) __attribute__((always_inline)) -> int {
    const int64_t wend_raw = wbase + kWave;
    const int64_t wend = wend_raw < e1 ? wend_raw : e1;
    for (;;) {
      const bool tail = r >= r1;  // the unfinished last row (or nothing, if r1 == M)
      int64_t rend = e1;
      if (!tail) {
        int j = (int)(r - rp_base);
        if (j == kWave) {
          rp_base = r;
          rp_l = load_rowends(rp_base);
          j = 0;
        }
        // j is wave-uniform: v_readlane keeps the row end (and the loop control) in SGPRs
        rend = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)rp_l >> 32), j) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)rp_l, j));
      }
      const int64_t stop = rend < wend ? rend : wend;
      if (e < stop) {
        accumulate_window<T, VEC, RED, MASKED>((int)(e - wbase), (int)(stop - wbase), (uint32_t)(wbase - e0),
                                               c_w, w_w, has_value, matk, K, lgG, g, val, arg, e_w, maskk,
                                               ws.rec_stride, mask_shift, z_w, mask_seg);
        e = stop;
      }
      if (e < rend) return 0;  // window exhausted inside the row
      if (tail) return 1;
      // row r ends here
      if (estart < rend) reduce_groups<A, VEC, RED>(lgG, val, arg);
      if (writer) {
        if (incoming && r == r0) {  // head of a cut row: the fix-up kernel finishes it
          write_carry<T, VEC, RED>(ws.head_val, ws.head_arg, carry_off, val, arg);
        } else {
          if constexpr (RED != RED_ADD && !kEmitRecords) widen_args();
          const uint64_t o = out_b + out_position(ws, r, M) * K;
          int64_t deg_w = rend - estart;
          if constexpr (kPartial && RED == RED_ADD) {
            if (mean && ws.deg_rowptr != nullptr) deg_w = ws.deg_rowptr[r + 1] - ws.deg_rowptr[r];
          }
          write_row<T, VEC, RED, A32, !kEmitRecords>(out, arg_out, o, val, arg64, deg_w, mean, E, ws);
        }
      }
      if constexpr (kEmitRecords) {
        if (incoming && r == r0) {
          blank_records(estart, rend, r);  // the last piece of a cut row: the fix-up kernel enters the winners
        } else if (estart < rend) {
          // the row lies inside this partition: its winners are final -- every entry gets its record, 64 entries per step:
          // the group-0 lanes (they hold the reduced winners of features k0 .. k0 + 3) set their four bits in the LDS
          // tile of the winning entries, then lane u writes the record of the step's u-th entry
          if constexpr (kRec32) {
            uint32_t *tile = rec_tile_ + wib * (kWave * 4);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            for (int64_t qb = estart; qb < rend; qb += kWave) {
              const uint32_t nq = (uint32_t)(rend - qb < (int64_t)kWave ? rend - qb : (int64_t)kWave);
              if ((uint32_t)lane < nq) *reinterpret_cast<u32x4 *>(tile + lane * 4) = u32x4{0u, 0u, 0u, 0u};
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (writer) {
                const uint32_t q0 = (uint32_t)(qb - e0), word = k0 >> 5, sh = k0 & 31u;
  #pragma unroll
                for (int j = 0; j < VEC; ++j) {
                  const uint32_t rel = arg[j] - q0;  // (kNoArg32 and earlier / later steps' entries fall outside [0, nq))
                  if (arg[j] != kNoArg32 && rel < nq && k0 + (uint32_t)j < K) atomicOr(tile + rel * 4 + word, 1u << (sh + (uint32_t)j));
                }
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if ((uint32_t)lane < nq) {
                const int64_t eid = qb + lane;
                const u32x4 m = *reinterpret_cast<const u32x4 *>(tile + lane * 4);
                const uint32_t z = (m.x != 0u ? 1u : 0u) | (m.y != 0u ? 2u : 0u) | (m.z != 0u ? 4u : 0u) | (m.w != 0u ? 8u : 0u);
                A wv = A(1);
                if (has_value) wv = Traits<T>::to_acc(value[eid]);
                uint32_t wbits;
                __builtin_memcpy(&wbits, &wv, 4);
                u32x4 *dst = reinterpret_cast<u32x4 *>(ws.rec_out + ((uint64_t)b * (uint64_t)E + (uint64_t)eid) * 8u);
                dst[0] = m;
                dst[1] = u32x4{(uint32_t)r, wbits, 0u, z};
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          } else {
            uint32_t *tile = rec_tile_ + wib * kRecTileWords;
            const uint32_t W = ws.rec_meta, S = ws.rec_stride;
            uint32_t *rec_b = ws.rec_out + (uint64_t)b * (uint64_t)E * S;
            for (int64_t qb = estart; qb < rend; qb += kWave) {
              const uint32_t nq = (uint32_t)(rend - qb < (int64_t)kWave ? rend - qb : (int64_t)kWave);
              records_clear(tile, lane, nq, S);
              if (writer) {
                const uint32_t q0 = (uint32_t)(qb - e0), word = k0 >> 5, sh = k0 & 31u;
  #pragma unroll
                for (int j = 0; j < VEC; ++j) {
                  const uint32_t rel = arg[j] - q0;  // (kNoArg32 and earlier / later steps' entries fall outside [0, nq))
                  if (arg[j] != kNoArg32 && rel < nq && k0 + (uint32_t)j < K) atomicOr(tile + rel * S + word, 1u << (sh + (uint32_t)j));
                }
              }
              records_finish<T>(tile, lane, nq, qb, (uint32_t)r, has_value ? value : nullptr, rec_b, W, S, ws.rec_has_z != 0);
            }
          }
        }
      }
      init_acc<T, VEC, RED>(val, arg);
      ++r;
      estart = e;
      if (short_row_batches()) return 2;
    }
  };
  if (!incoming) short_row_batches();  // the partition starts at a row start
  for (;;) {
    int st = process_window(c_cur, w_cur, e_cur, z_cur);
    if (st == 1) break;
    if (st == 2) continue;
    load_window(wbase + 2 * kWave, c_cur, w_cur, e_cur, z_cur);
    wbase += kWave;
    st = process_window(c_nxt, w_nxt, e_nxt, z_nxt);
    if (st == 1) break;
    if (st == 2) continue;
    load_window(wbase + 2 * kWave, c_nxt, w_nxt, e_nxt, z_nxt);
    wbase += kWave;
  }
  // tail: the piece of the unfinished row r1 that falls into this partition
  if (r1 < M && estart < e1) {
    reduce_groups<A, VEC, RED>(lgG, val, arg);
    if (writer) write_carry<T, VEC, RED>(ws.tail_val, ws.tail_arg, carry_off, val, arg);
    trow = r1;
    blank_records(estart, e1, r1);
  }
  if (y == 0 && lane == 0) {
    ws.tail_row[p] = trow;
    // a cut first row that ends here (rows [r0, r1) end in this partition): the fix-up kernel finds its id in
    // head_row[p]
    ws.head_row[p] = (incoming && r1 > r0) ? r0 : -1;
  }
}

// ---------------------------------------------------------------------------
// 3. fix-up: partition q in which a cut row ends (the merge kernel left its id in head_row[q]) folds
//    that row's tail records q-1, q-2, ... and writes the final value.
//    One wave per (q, b); lanes stride over K.
//    The kernel is a chain of dependent round trips, not a stream (165 k waves of a few hundred bytes
//    each): it used to take five of them (table -> rowptr -> tail_row -> records -> store).  Now the
//    row id comes from one word, and everything else -- the tail ids of the 64 partitions before q, the
//    row's degree, the head record and the FIRST tail record (a cut row always has one, in q-1) --
//    is requested at once and waited for once.
// ---------------------------------------------------------------------------

template <typename T, int RED, bool A32 = false, int REC = 0>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_fixup_kernel(
    const int64_t *__restrict__ rowptr, T *__restrict__ out, int64_t *__restrict__ arg_out,
    int64_t M, uint32_t K, int64_t E, bool mean, Workspace ws) {
  using A = typename Traits<T>::acc_t;
  constexpr bool kEmitRecords = REC != 0 && A32 && RED != RED_ADD && sizeof(A) == 4 && !kPartial;
  static_assert(REC == 0 || kEmitRecords, "record-writing fix-up kernel: int32 ids, min / max, 4-byte accumulators");
  __shared__ alignas(16) uint32_t rec_tile_[REC == 1 ? kWavesPerBlock * kWave * 4 : (REC == 2 ? kWavesPerBlock * kRecTileWords : 4)];
  // REC: the row's winners (offsets from its first entry) of features lane, lane + 64, lane + 128, lane + 192
  uint32_t rec_rel[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t q = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  if (q >= ws.P || q == 0) return;  // partition 0 starts at a row start
  const uint32_t b = blockIdx.y;
  const int64_t R = ws.head_row[q];
  if (R < 0) return;  // no row ends here that started earlier

  const A *head_val = reinterpret_cast<const A *>(ws.head_val);
  const A *tail_val = reinterpret_cast<const A *>(ws.tail_val);
  const uint64_t plane = (uint64_t)b * ws.P;
  const uint64_t hbase = (plane + (uint64_t)q) * K, tbase = (plane + (uint64_t)q - 1) * K;
  constexpr int kCols = 2;  // feature columns per lane and step
  A hv[kCols], tv[kCols];
  int64_t ha[kCols], ta[kCols];
  // min / max: the records hold the winners as offsets from their partition's first edge
  auto widen = [](uint32_t a, int64_t first) -> int64_t { return a == kNoArg32 ? kNoArg : first + (int64_t)a; };
  int64_t e_head = 0, e_tail = 0;
  if constexpr (RED != RED_ADD) {
    e_head = ws.table[q].edge;
    e_tail = ws.table[q - 1].edge;
  }
  auto fetch = [&](uint32_t kb) {
#pragma unroll
    for (int u = 0; u < kCols; ++u) {
      const uint32_t k = kb + (uint32_t)(u * kWave + lane);
      hv[u] = tv[u] = A(0);
      ha[u] = ta[u] = kNoArg;
      if (k < K) {
        hv[u] = head_val[hbase + k];
        tv[u] = tail_val[tbase + k];
        if constexpr (RED != RED_ADD) {
          ha[u] = widen(ws.head_arg[hbase + k], e_head);
          ta[u] = widen(ws.tail_arg[tbase + k], e_tail);
        }
      }
    }
  };
  auto pin = [&]() {  // one wait for the whole batch (and no sinking of the loads behind the run count)
#pragma unroll
    for (int u = 0; u < kCols; ++u) {
      asm volatile("" : "+v"(hv[u]), "+v"(tv[u]));
      if constexpr (RED != RED_ADD) asm volatile("" : "+v"(ha[u]), "+v"(ta[u]));
    }
  };

  // ---- the one round trip ----
  const int64_t idx0 = q - 1 - lane;
  int64_t t_l = idx0 >= 0 ? ws.tail_row[idx0] : -1;
  const int64_t rs = rowptr[R];
  const int64_t deg = rowptr[R + 1] - rs;
  fetch(0);
  asm volatile("" : "+v"(t_l));
  pin();

  // tail records of row R sit in the partitions right before q: count the leading matches
  int64_t run;
  {
    const unsigned long long m = __ballot(t_l == R);
    run = m == ~0ull ? 64 : (int64_t)__builtin_ctzll(~m);
  }
  if (run == 64) {  // a hub row cut into more than 64 pieces
    for (;;) {
      const int64_t idx = q - 1 - run - lane;
      const bool ok = idx >= 0 && ws.tail_row[idx] == R;
      const unsigned long long m = __ballot(ok);
      const int c = m == ~0ull ? 64 : (int)__builtin_ctzll(~m);
      run += c;
      if (c < 64) break;
    }
  }

  for (uint32_t kb = 0;;) {
#pragma unroll
    for (int u = 0; u < kCols; ++u) {
      const uint32_t k = kb + (uint32_t)(u * kWave + lane);
      if (k >= K) continue;
      A val[1];
      int64_t arg[1];
      val[0] = hv[u];
      arg[0] = ha[u];
      // A hub row is cut into hundreds of pieces; folding their fp32 partial sums in fp64 keeps the
      // error of a long row at that of one piece (costs nothing: a few records per cut row).
      constexpr bool kWideFold = RED == RED_ADD && std::is_same<A, float>::value;
      double wide = kWideFold ? (double)val[0] : 0.0;
      auto fold = [&](A v, int64_t a) {
        if constexpr (kWideFold) {
          wide += (double)v;
        } else if constexpr (RED == RED_ADD) {
          val[0] += v;
        } else {
          const bool better = RED == RED_MIN ? (v < val[0]) : (v > val[0]);
          if (better || (v == val[0] && a < arg[0])) {
            val[0] = v;
            arg[0] = a;
          }
        }
      };
      if (run >= 1) fold(tv[u], ta[u]);
      // the further records of a hub row are fetched kFold at a time: the loads of a batch are
      // independent, only the fold itself is sequential
      constexpr int kFold = 8;
      int64_t i = 1;
      for (; i + kFold <= run; i += kFold) {
        A v[kFold];
        int64_t a[kFold];
#pragma unroll
        for (int f = 0; f < kFold; ++f) {
          const uint64_t o = (plane + (uint64_t)(q - 1 - i - f)) * K + k;
          v[f] = tail_val[o];
          a[f] = kNoArg;
          if constexpr (RED != RED_ADD) a[f] = widen(ws.tail_arg[o], ws.table[q - 1 - i - f].edge);
        }
#pragma unroll
        for (int f = 0; f < kFold; ++f) fold(v[f], a[f]);
      }
      for (; i < run; ++i) {
        const uint64_t o = (plane + (uint64_t)(q - 1 - i)) * K + k;
        int64_t a = kNoArg;
        if constexpr (RED != RED_ADD) a = widen(ws.tail_arg[o], ws.table[q - 1 - i].edge);
        fold(tail_val[o], a);
      }
      if constexpr (kWideFold) val[0] = (A)wide;
      const uint64_t o = ((uint64_t)b * M + out_position(ws, R, M)) * K + k;
      int64_t deg_w = deg;
      if constexpr (kPartial && RED == RED_ADD) {
        if (mean && ws.deg_rowptr != nullptr) deg_w = ws.deg_rowptr[R + 1] - ws.deg_rowptr[R];
      }
      if constexpr (kEmitRecords) {
        const uint32_t rl = (arg[0] == kNoArg || deg <= 0) ? 0xFFFFFFFFu : (uint32_t)(arg[0] - rs);
        if (kb == 0) rec_rel[u] = rl;
        else rec_rel[2 + u] = rl;
      }
      write_row<T, 1, RED, A32, !kEmitRecords>(out, arg_out, o, val, arg, deg_w, mean, E, ws);
    }
    kb += (uint32_t)(kCols * kWave);
    if (kb >= K) break;
    fetch(kb);
    pin();
  }
  if constexpr (REC == 1) {  // 32-byte records, K <= 128: the loop above ran once, rec_rel[0..1] hold every feature's winner
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (deg > kFixupRecordMax) {
      // a long row: at most K of its entries win anything.  Every piece of the row already has records without winners
      // (merge kernel); the wave walks the DISTINCT winners -- the lanes (features) that share one are found by a ballot,
      // which is that entry's mask -- and rewrites only those records' masks: cost independent of the row's length
      const bool v0 = (uint32_t)lane < K && rec_rel[0] != 0xFFFFFFFFu, v1 = (uint32_t)(kWave + lane) < K && rec_rel[1] != 0xFFFFFFFFu;
      unsigned long long todo0 = __ballot(v0), todo1 = __ballot(v1);
      while ((todo0 | todo1) != 0ull) {
        // up to 64 distinct winners per round: lane i keeps the i-th one's entry and mask, then all of them store at once
        uint32_t my_w = 0, my_m0 = 0, my_m1 = 0, my_m2 = 0, my_m3 = 0;
        int cnt = 0;
        while ((todo0 | todo1) != 0ull && cnt < kWave) {
          uint32_t w;
          if (todo0 != 0ull) w = (uint32_t)__builtin_amdgcn_readlane((int)rec_rel[0], (int)__builtin_ctzll(todo0));
          else w = (uint32_t)__builtin_amdgcn_readlane((int)rec_rel[1], (int)__builtin_ctzll(todo1));
          const unsigned long long m0 = __ballot(v0 && rec_rel[0] == w), m1 = __ballot(v1 && rec_rel[1] == w);
          todo0 &= ~m0;
          todo1 &= ~m1;
          const bool me = lane == cnt;
          my_w = me ? w : my_w;
          my_m0 = me ? (uint32_t)m0 : my_m0;
          my_m1 = me ? (uint32_t)(m0 >> 32) : my_m1;
          my_m2 = me ? (uint32_t)m1 : my_m2;
          my_m3 = me ? (uint32_t)(m1 >> 32) : my_m3;
          ++cnt;
        }
        if (lane < cnt) {
          const uint32_t z = (my_m0 != 0u ? 1u : 0u) | (my_m1 != 0u ? 2u : 0u) | (my_m2 != 0u ? 4u : 0u) | (my_m3 != 0u ? 8u : 0u);
          uint32_t *dst = ws.rec_out + ((uint64_t)b * (uint64_t)E + (uint64_t)(rs + (int64_t)my_w)) * 8u;
          *reinterpret_cast<u32x4 *>(dst) = u32x4{my_m0, my_m1, my_m2, my_m3};
          dst[7] = z;
        }
      }
      return;
    }
    uint32_t *tile = rec_tile_ + wib * (kWave * 4);
    const T *value = reinterpret_cast<const T *>(ws.rec_value);
    for (int64_t q0 = 0; q0 < deg; q0 += kWave) {
      const uint32_t nq = (uint32_t)(deg - q0 < (int64_t)kWave ? deg - q0 : (int64_t)kWave);
      if ((uint32_t)lane < nq) *reinterpret_cast<u32x4 *>(tile + lane * 4) = u32x4{0u, 0u, 0u, 0u};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < kCols; ++u) {
        const uint32_t k = (uint32_t)(u * kWave + lane), rel = rec_rel[u] - (uint32_t)q0;
        if (k < K && rec_rel[u] != 0xFFFFFFFFu && rel < nq) atomicOr(tile + rel * 4 + (k >> 5), 1u << (k & 31u));
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if ((uint32_t)lane < nq) {
        const int64_t eid = rs + q0 + lane;
        const u32x4 m = *reinterpret_cast<const u32x4 *>(tile + lane * 4);
        const uint32_t z = (m.x != 0u ? 1u : 0u) | (m.y != 0u ? 2u : 0u) | (m.z != 0u ? 4u : 0u) | (m.w != 0u ? 8u : 0u);
        A wv = A(1);
        if (value != nullptr) wv = Traits<T>::to_acc(value[eid]);
        uint32_t wbits;
        __builtin_memcpy(&wbits, &wv, 4);
        u32x4 *dst = reinterpret_cast<u32x4 *>(ws.rec_out + ((uint64_t)b * (uint64_t)E + (uint64_t)eid) * 8u);
        dst[0] = m;
        dst[1] = u32x4{(uint32_t)R, wbits, 0u, z};
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if constexpr (REC == 2) {  // any record shape, K <= 256: rec_rel holds every feature's winner
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t W = ws.rec_meta, S = ws.rec_stride;
    uint32_t *rec_b = ws.rec_out + (uint64_t)b * (uint64_t)E * S;
    if (deg > kFixupRecordMax) {
      // a long row: at most K of its entries win anything.  Every piece of the row already has records without winners
      // (merge kernel); the wave walks the DISTINCT winners -- the lanes (features) that share one are found by a ballot,
      // which is that entry's mask -- and rewrites only those records' masks: cost independent of the row's length
      bool v[4];
      unsigned long long todo[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[c] = (uint32_t)(c * kWave + lane) < K && rec_rel[c] != 0xFFFFFFFFu;
        todo[c] = __ballot(v[c]);
      }
      while ((todo[0] | todo[1] | todo[2] | todo[3]) != 0ull) {
        // up to 64 distinct winners per round: lane i keeps the i-th one's entry and mask, then all of them store at once
        uint32_t my_w = 0, my_m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        int cnt = 0;
        while ((todo[0] | todo[1] | todo[2] | todo[3]) != 0ull && cnt < kWave) {
          uint32_t w = 0;
          if (todo[0] != 0ull) w = (uint32_t)__builtin_amdgcn_readlane((int)rec_rel[0], (int)__builtin_ctzll(todo[0]));
          else if (todo[1] != 0ull) w = (uint32_t)__builtin_amdgcn_readlane((int)rec_rel[1], (int)__builtin_ctzll(todo[1]));
          else if (todo[2] != 0ull) w = (uint32_t)__builtin_amdgcn_readlane((int)rec_rel[2], (int)__builtin_ctzll(todo[2]));
          else w = (uint32_t)__builtin_amdgcn_readlane((int)rec_rel[3], (int)__builtin_ctzll(todo[3]));
          const bool me = lane == cnt;
          my_w = me ? w : my_w;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned long long m = __ballot(v[c] && rec_rel[c] == w);
            todo[c] &= ~m;
            my_m[2 * c] = me ? (uint32_t)m : my_m[2 * c];
            my_m[2 * c + 1] = me ? (uint32_t)(m >> 32) : my_m[2 * c + 1];
          }
          ++cnt;
        }
        if (lane < cnt) {
          uint32_t z = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) z |= my_m[j] != 0u ? (1u << j) : 0u;
          uint32_t *dst = rec_b + (uint64_t)(rs + (int64_t)my_w) * S;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if ((uint32_t)j < W) dst[j] = my_m[j];
          if (ws.rec_has_z) dst[W + 3u] = z;
        }
      }
      return;
    }
    uint32_t *tile = rec_tile_ + wib * kRecTileWords;
    for (int64_t q0 = 0; q0 < deg; q0 += kWave) {
      const uint32_t nq = (uint32_t)(deg - q0 < (int64_t)kWave ? deg - q0 : (int64_t)kWave);
      records_clear(tile, lane, nq, S);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t k = (uint32_t)(c * kWave + lane), rel = rec_rel[c] - (uint32_t)q0;
        if (k < K && rec_rel[c] != 0xFFFFFFFFu && rel < nq) atomicOr(tile + rel * S + (k >> 5), 1u << (k & 31u));
      }
      records_finish<T>(tile, lane, nq, rs + q0, (uint32_t)R, reinterpret_cast<const T *>(ws.rec_value), rec_b, W, S,
                        ws.rec_has_z != 0);
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int ilog2_ceil(uint32_t x) {
  int l = 0;
  while ((1u << l) < x) ++l;
  return l;
}

// Items (row ends + entries) per wave.  Same-box A/B on the north-star graph (scripts/variants.py,
// TSAMD_ITEMS_MAX = 64..2048): 256 items beat 1024 by 5-18 % for rows up to 512 bytes (F = 16..128
// fp32, every f16/bf16 width up to 256: more, shorter waves fill the machine better and the tail is
// shorter), 1024 items beat 256 by 4-8 % for rows of 1 KB and more (F = 256/512 fp32: every partition
// pays a K-wide carry record and a fix-up).  TSAMD_ITEMS_MAX caps both.
constexpr int64_t kShortRowItems = 1024;

void plan_partition(int64_t M, int64_t E, int64_t row_bytes, int64_t *P, int64_t *items) {
  const int64_t total = M + E > 0 ? M + E : 1;
  int64_t cap = row_bytes <= 512 ? 256 : (row_bytes < 1024 ? 512 : 1024);
  if (row_bytes <= 128) cap = kShortRowItems;  // side-by-side short rows: long partitions amortise the batches
  if (const char *env = exp_env("TSAMD_SPMM_ITEMS")) {  // experiments
    const long v = atol(env);
    if (v >= 64 && v <= 4096) cap = v;
  }
  if (cap > TSAMD_ITEMS_MAX) cap = TSAMD_ITEMS_MAX;
  int64_t it = ceil_div(total, (int64_t)TSAMD_TARGET_WAVES);
  if (it < TSAMD_ITEMS_MIN) it = TSAMD_ITEMS_MIN;
  if (it > cap) it = cap;
  *items = it;
  *P = ceil_div(total, it);
}

// The relabelled copy only pays off for big problems whose rows are 16-byte packets.
// Cost: one read + one write of mat (2 * N rows); gain: ~30 % of the time to gather E rows.
// It therefore needs E >= ~7 N; 8 N is used (a row-sharded block with few edges per column of
// the gathered X -- the multi-GPU case -- does not qualify).
//
// Only rows whose byte size is a power of two (>= 128 B) camp on memory channels: with hub ids
// that are multiples of big powers of two, `id * pitch` keeps its low address bits zero only if the
// pitch is a power of two itself.  Same-box A/B on the north-star graph (scripts/bench_fsweep.py,
// TSAMD_SPMM_RELABEL=0/1): F = 32 / 64 / 128 / 256 fp32 gain 8 / 10 / 26 / 30 % from the copy,
// F = 24 / 40 / 48 / 80 / 96 / 112 / 160 / 192 LOSE 13-25 % (they spread by themselves and only pay
// for the copy and the hashing), 64-byte rows lose 2-10 %.  TSAMD_SPMM_RELABEL=1 still forces it.
bool relabel_forced() {
  const char *env = exp_env("TSAMD_SPMM_RELABEL");
  return env != nullptr && env[0] == '1';
}

// Round 3 same-box A/B over reduction x element type x row size on the scale-20 / 21 R-MAT graphs
// (scripts/ab_relabel.py, profiles/r03_ab_relabel.jsonl): the copy LOSES 3-11 % for min / max on f16 / bf16
// at every row size (those kernels are bound by instruction issue, not by the camped channels, and pay
// the copy and the per-entry hashing on top) and 15 % for fp32 min / max on 128-byte rows; 128-byte rows
// of sums are a wash (-5 ... +5 %).  It stays for sums on power-of-two rows >= 256 bytes (3-23 % gain)
// and for fp32 / fp64 min / max on rows >= 256 bytes (2-20 %).
bool relabel_possible(int dtype, int reduce, int64_t N, int64_t K, int64_t E) {
  const int64_t row_bytes = K * (int64_t)dtype_size(dtype);
  const bool size_ok = E >= (1 << 20) && N >= 4096 && N < ((int64_t)1 << 32) && E >= 8 * N &&
                       row_bytes % 16 == 0;
  const bool minmax = reduce == TSAMD_MIN || reduce == TSAMD_MAX;
  const bool camps = row_bytes >= 256 && (row_bytes & (row_bytes - 1)) == 0 && !(minmax && dtype_size(dtype) < 4);
  return size_ok && (camps || relabel_forced());
}

size_t carve(void *base, int dtype, int reduce, int64_t B, int64_t M, int64_t N, int64_t K,
             int64_t E, Workspace *ws, bool relabelled = false) {
  int64_t P, items;
  plan_partition(M, E, K * (int64_t)dtype_size(dtype), &P, &items);
  const bool minmax = reduce == TSAMD_MIN || reduce == TSAMD_MAX;
  char *p = reinterpret_cast<char *>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) -> void * {
    void *r = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return r;
  };
  const size_t plane = (size_t)B * P * K;
  Workspace w;
  w.P = P;
  w.items = items;
  w.table = reinterpret_cast<Coord *>(take(sizeof(Coord) * (P + 1)));
  w.tail_row = reinterpret_cast<int64_t *>(take(sizeof(int64_t) * P));
  w.head_val = take(acc_size(dtype) * plane);
  w.tail_val = take(acc_size(dtype) * plane);
  w.head_arg = reinterpret_cast<uint32_t *>(minmax ? take(sizeof(uint32_t) * plane) : nullptr);
  w.tail_arg = reinterpret_cast<uint32_t *>(minmax ? take(sizeof(uint32_t) * plane) : nullptr);
  w.relabel_mode = 0;
  w.relabel_flag = reinterpret_cast<int *>(take(256));
  if (const char *env = exp_env("TSAMD_SPMM_XPERM_PAD")) {  // experiments: shift the copy of X
    const long v = atol(env);
    if (v > 0 && v <= (64l << 20)) (void)take((size_t)v);
  }
  w.xperm = (!relabelled && relabel_possible(dtype, reduce, N, K, E)) ? take(dtype_size(dtype) * (size_t)B * N * K) : nullptr;
  // (carved behind the copy of X: the position of that copy relative to the start of the workspace decides which
  // of its hot rows share a memory channel -- 3-5 % of the north-star kernel either way, measured by padding --
  // and the layout in front of it is the one the round-2/3 numbers were taken with)
  w.head_row = reinterpret_cast<int64_t *>(take(sizeof(int64_t) * P));
  w.hash_bits = 1;
  while (w.hash_bits < 32 && ((uint64_t)1 << w.hash_bits) < (uint64_t)(N > 1 ? N : 2)) ++w.hash_bits;
  w.hash_mul = 0x9E3779B1u;  // odd (golden-ratio) multiplier
  w.hash_shift = w.hash_bits > 1 ? w.hash_bits / 2 : 1;
  w.out_relabel = 0;
  w.perm = nullptr;
  w.wmask = nullptr;
  w.rec_stride = w.rec_meta = 0;
  w.rec_has_z = 0;
  w.fp_stored = w.fp_new = nullptr;
  w.cache_fp = nullptr;
  w.cache_state = 0;
  w.partial = w.accumulate = 0;
  w.arg_map = nullptr;
  w.arg_none = 0;
  w.deg_rowptr = nullptr;
  w.arg32 = 0;
  w.rec_out = nullptr;
  w.rec_value = nullptr;
  w.snap = 0;
  w.ohash_bits = 1;
  while (w.ohash_bits < 32 && ((uint64_t)1 << w.ohash_bits) < (uint64_t)(M > 1 ? M : 2)) ++w.ohash_bits;
  w.ohash_shift = w.ohash_bits > 1 ? w.ohash_bits / 2 : 1;
  if (ws) *ws = w;
  return off;
}

// operand cache buffer: [probe counters 256 B | fingerprints 2 x 512 B | relabelled copy of mat]
constexpr size_t kOperandCacheHeader = 256 + 2 * 512;
size_t operand_cache_bytes(int dtype, int reduce, int64_t B, int64_t N, int64_t K, int64_t E) {
  if (!relabel_possible(dtype, reduce, N, K, E)) return 0;
  return kOperandCacheHeader + align_up(dtype_size(dtype) * (size_t)B * N * K, 256);
}

template <typename T, int VEC, int RED>
int launch_spmm(const int64_t *rowptr, const int64_t *col, const T *value, const T *mat,
                T *out, int64_t *arg_out, int64_t B, int64_t M, int64_t N, int64_t K,
                int64_t E, bool mean, Workspace ws, hipStream_t stream, hipEvent_t *ev) {
  const uint32_t slots = (uint32_t)((K + VEC - 1) / VEC);  // feature packets per row
  const uint32_t lpr = slots >= 64 ? 64u : (1u << ilog2_ceil(slots));
  const int lgG = 6 - ilog2_ceil(lpr);
  const uint32_t ktiles = (slots + 63) / 64;
  const unsigned int threads = kWavesPerBlock * kWave;

  if (ev) TSAMD_HIP_TRY(hipEventRecord(ev[0], stream));
  {
    int mode = 0;
    if (ws.xperm != nullptr && VEC > 1 && !ws.out_relabel) {
      const char *env = exp_env("TSAMD_SPMM_RELABEL");
      mode = env ? (env[0] == '1' ? 1 : (env[0] == '0' ? 0 : 2)) : 2;
    }
    // masked sums (the pull of the min / max backward): `col` points at winner records, not at column ids, so there
    // is nothing to probe; same-box A/B at configs[2]: 1.76 ms with the copy, 1.93 without (the rows gathered are
    // grad_out rows indexed by the R-MAT ROW ids of the forward, which camp like its column ids)
    if (mode == 2 && ws.wmask != nullptr) mode = 1;
    ws.relabel_mode = mode;
    const bool cached = ws.cache_state != 0 && mode != 0;
    if (mode == 2 && !(cached && ws.cache_state == 2)) {  // (a reused cache keeps the verdict of its first call)
      TSAMD_HIP_TRY(hipMemsetAsync(ws.relabel_flag, 0, 4 * sizeof(int), stream));
      hipLaunchKernelGGL(spmm_probe_kernel, dim3(kProbeBlocks), dim3(256), 0, stream, col, E,
                         ws.relabel_flag);
      TSAMD_LAUNCH_CHECK();
    }
    if (mode != 0) {
      if (cached) {
        hipLaunchKernelGGL(spmm_fingerprint_kernel, dim3(kFingerprintWords), dim3(256), 0, stream,
                           reinterpret_cast<const uint4 *>(mat), (uint64_t)(B * N * K) * sizeof(T) / 16,
                           ws.cache_fp + kFingerprintWords);
        TSAMD_LAUNCH_CHECK();
        if (ws.cache_state == 2) {
          ws.fp_stored = ws.cache_fp;
          ws.fp_new = ws.cache_fp + kFingerprintWords;
        }
      }
      // the copy always moves 16-byte packets, whatever packet size the reduction kernel uses
      constexpr int kPV = 16 / (int)sizeof(T);
      const uint32_t pslots = (uint32_t)(K / kPV);
      int lgL = 0;
      while (lgL < 8 && (1u << lgL) < pslots) ++lgL;
      hipLaunchKernelGGL((spmm_permute_rows_kernel<T, kPV>), dim3(TSAMD_PERMUTE_BLOCKS), dim3(256), 0, stream, mat,
                         reinterpret_cast<T *>(ws.xperm), B * N, (uint32_t)N, (uint32_t)K, lgL, ws);
      TSAMD_LAUNCH_CHECK();
      if (cached)
        TSAMD_HIP_TRY(hipMemcpyAsync(ws.cache_fp, ws.cache_fp + kFingerprintWords,
                                     sizeof(unsigned long long) * kFingerprintWords, hipMemcpyDeviceToDevice, stream));
    }
  }
  hipLaunchKernelGGL(spmm_partition_kernel, dim3((unsigned int)ceil_div(ws.P + 1, 256)), dim3(256),
                     0, stream, rowptr, M, E, ws);
  TSAMD_LAUNCH_CHECK();
  if (ev) TSAMD_HIP_TRY(hipEventRecord(ev[1], stream));
  const unsigned int gx = (unsigned int)ceil_div(ws.P, kWavesPerBlock);
  // int32 winner ids (tsamd_spmm_minmax_arg32): min / max of the floating types (what autograd differentiates)
  constexpr bool kArg32able = !kPartial && RED != RED_ADD &&
                              (std::is_same<T, float>::value || std::is_same<T, double>::value ||
                               std::is_same<T, f16_t>::value || std::is_same<T, bf16_t>::value);
  constexpr bool kMaskable = RED == RED_ADD && (std::is_same<T, float>::value || std::is_same<T, double>::value ||
                                                std::is_same<T, f16_t>::value || std::is_same<T, bf16_t>::value);
  if (ws.wmask != nullptr) {
    if constexpr (kMaskable)
      hipLaunchKernelGGL((spmm_merge_kernel<T, VEC, RED, false, true>), dim3(gx, (unsigned int)(B * ktiles), 1),
                         dim3(threads), 0, stream, rowptr, col, value, mat, out, arg_out, M, N, (uint32_t)K, E,
                         ktiles, lgG, mean, ws);
    else
      return TSAMD_ERR_UNSUPPORTED;
  } else if (kArg32able && ws.arg32) {
    if constexpr (kArg32able) {
      constexpr bool kRecordable = VEC == 4 && sizeof(typename Traits<T>::acc_t) == 4;
      if (ws.rec_out != nullptr) {
        if constexpr (kRecordable) {
          if (lgG > 2 || ktiles != 1) return TSAMD_ERR_UNSUPPORTED;  // (spmm_emits_records: 33..256 features)
          if (ws.rec_meta == 4u && ws.rec_stride == 8u)
            hipLaunchKernelGGL((spmm_merge_kernel<T, VEC, RED, false, false, true, 1>), dim3(gx, (unsigned int)(B * ktiles), 1),
                               dim3(threads), 0, stream, rowptr, col, value, mat, out, arg_out, M, N, (uint32_t)K, E,
                               ktiles, lgG, mean, ws);
          else
            hipLaunchKernelGGL((spmm_merge_kernel<T, VEC, RED, false, false, true, 2>), dim3(gx, (unsigned int)(B * ktiles), 1),
                               dim3(threads), 0, stream, rowptr, col, value, mat, out, arg_out, M, N, (uint32_t)K, E,
                               ktiles, lgG, mean, ws);
        } else {
          return TSAMD_ERR_UNSUPPORTED;
        }
      } else if (lgG >= 3)
        hipLaunchKernelGGL((spmm_merge_kernel<T, VEC, RED, true, false, true>), dim3(gx, (unsigned int)(B * ktiles), 1),
                           dim3(threads), 0, stream, rowptr, col, value, mat, out, arg_out, M, N, (uint32_t)K, E,
                           ktiles, lgG, mean, ws);
      else
        hipLaunchKernelGGL((spmm_merge_kernel<T, VEC, RED, false, false, true>), dim3(gx, (unsigned int)(B * ktiles), 1),
                           dim3(threads), 0, stream, rowptr, col, value, mat, out, arg_out, M, N, (uint32_t)K, E,
                           ktiles, lgG, mean, ws);
    }
  } else if (ws.arg32) {
    return TSAMD_ERR_UNSUPPORTED;
  } else if (lgG >= 3)
    hipLaunchKernelGGL((spmm_merge_kernel<T, VEC, RED, true>), dim3(gx, (unsigned int)(B * ktiles), 1),
                       dim3(threads), 0, stream, rowptr, col, value, mat, out, arg_out, M, N,
                       (uint32_t)K, E, ktiles, lgG, mean, ws);
  else
    hipLaunchKernelGGL((spmm_merge_kernel<T, VEC, RED, false>), dim3(gx, (unsigned int)(B * ktiles), 1),
                       dim3(threads), 0, stream, rowptr, col, value, mat, out, arg_out, M, N,
                       (uint32_t)K, E, ktiles, lgG, mean, ws);
  TSAMD_LAUNCH_CHECK();
  if (ev) TSAMD_HIP_TRY(hipEventRecord(ev[2], stream));
  if (ws.P > 1) {
    if (kArg32able && ws.arg32) {
      if constexpr (kArg32able) {
        constexpr bool kRecordable = VEC == 4 && sizeof(typename Traits<T>::acc_t) == 4;
        if (ws.rec_out != nullptr) {
          if constexpr (kRecordable) {
            if (ws.rec_meta == 4u && ws.rec_stride == 8u)
              hipLaunchKernelGGL((spmm_fixup_kernel<T, RED, true, 1>), dim3(gx, (unsigned int)B, 1), dim3(threads), 0,
                                 stream, rowptr, out, arg_out, M, (uint32_t)K, E, mean, ws);
            else
              hipLaunchKernelGGL((spmm_fixup_kernel<T, RED, true, 2>), dim3(gx, (unsigned int)B, 1), dim3(threads), 0,
                                 stream, rowptr, out, arg_out, M, (uint32_t)K, E, mean, ws);
          }
        } else {
          hipLaunchKernelGGL((spmm_fixup_kernel<T, RED, true>), dim3(gx, (unsigned int)B, 1), dim3(threads), 0,
                             stream, rowptr, out, arg_out, M, (uint32_t)K, E, mean, ws);
        }
      }
    } else {
      hipLaunchKernelGGL((spmm_fixup_kernel<T, RED>), dim3(gx, (unsigned int)B, 1), dim3(threads), 0,
                         stream, rowptr, out, arg_out, M, (uint32_t)K, E, mean, ws);
    }
    TSAMD_LAUNCH_CHECK();
  }
  if (ev) TSAMD_HIP_TRY(hipEventRecord(ev[3], stream));
  return TSAMD_OK;
}

template <typename T, int VEC>
int dispatch_reduce(int reduce, const int64_t *rowptr, const int64_t *col, const T *v, const T *x,
                    T *o, int64_t *arg_out, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                    Workspace ws, hipStream_t stream, hipEvent_t *ev) {
  const bool mean = reduce == TSAMD_MEAN;
#if TSAMD_SPMM_TU == 0 || TSAMD_SPMM_TU == 2
  if (reduce == TSAMD_MIN)
    return launch_spmm<T, VEC, RED_MIN>(rowptr, col, v, x, o, arg_out, B, M, N, K, E, mean, ws, stream, ev);
#endif
#if TSAMD_SPMM_TU == 0 || TSAMD_SPMM_TU == 3
  if (reduce == TSAMD_MAX)
    return launch_spmm<T, VEC, RED_MAX>(rowptr, col, v, x, o, arg_out, B, M, N, K, E, mean, ws, stream, ev);
#endif
#if TSAMD_SPMM_TU <= 1
  if (reduce == TSAMD_SUM || reduce == TSAMD_MEAN)
    return launch_spmm<T, VEC, RED_ADD>(rowptr, col, v, x, o, arg_out, B, M, N, K, E, mean, ws, stream, ev);
#endif
  return TSAMD_ERR_UNSUPPORTED;  // (the other translation unit's reductions: spmm_entry never sends them here)
}

// `vec` = elements per lane packet, chosen by the caller: the largest power of two <= kMaxVec<T>
// that divides K and matches the pointers' alignment.
template <typename T>
constexpr int kMaxVec = sizeof(T) <= 2 ? 4 : 16 / (int)sizeof(T);

template <typename T>
int dispatch_spmm(int reduce, int vec, const int64_t *rowptr, const int64_t *col,
                  const void *value, const void *mat, void *out, int64_t *arg_out, int64_t B,
                  int64_t M, int64_t N, int64_t K, int64_t E, Workspace ws, hipStream_t stream,
                  hipEvent_t *ev) {
  // Packet per lane: up to 16 bytes for 4/8-byte types; up to 8 bytes (4 elements) for f16/bf16 --
  // with 8 narrow elements per lane the per-element state (fp32 accumulator, and the arg for
  // min/max) costs 80-84 VGPRs = 5 waves/SIMD, with 4 it is 8 waves/SIMD (measured +5...27 %).
  // Row pitches that are not a multiple of 16 bytes fall back to 8- or 4-byte packets
  // (e.g. F = 602 fp32 -> 8 bytes), odd pitches to single elements.
  const T *v = reinterpret_cast<const T *>(value);
  const T *x = reinterpret_cast<const T *>(mat);
  T *o = reinterpret_cast<T *>(out);
#define TSAMD_SPMM_GO(VEC) \
  return dispatch_reduce<T, VEC>(reduce, rowptr, col, v, x, o, arg_out, B, M, N, K, E, ws, stream, ev)
  if constexpr (kMaxVec<T> >= 4) {
    if (vec >= 4) TSAMD_SPMM_GO(4);
  }
  if constexpr (kMaxVec<T> >= 2) {
    if (vec >= 2) TSAMD_SPMM_GO(2);
  }
  TSAMD_SPMM_GO(1);
#undef TSAMD_SPMM_GO
}

}  // namespace

// min / max launches of the other translation units (spmm_min.hip / spmm_max.hip); `ws` = the caller's Workspace, byte
// for byte
#define TSAMD_BRIDGE_ARGS                                                                                             \
  int dtype, int reduce, int vec, const int64_t *rowptr, const int64_t *col, const void *value, const void *mat,      \
      void *out, int64_t *arg_out, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E, const void *ws_blob,        \
      size_t ws_bytes, hipStream_t stream, hipEvent_t *ev
int spmm_min_bridge(TSAMD_BRIDGE_ARGS);
int spmm_max_bridge(TSAMD_BRIDGE_ARGS);
#if TSAMD_SPMM_TU >= 2
#if TSAMD_SPMM_TU == 2
int spmm_min_bridge(TSAMD_BRIDGE_ARGS) {
  if (reduce != TSAMD_MIN) return TSAMD_ERR_INVALID;
#else
int spmm_max_bridge(TSAMD_BRIDGE_ARGS) {
  if (reduce != TSAMD_MAX) return TSAMD_ERR_INVALID;
#endif
  if (ws_bytes != sizeof(Workspace)) return TSAMD_ERR_INVALID;
  Workspace ws;
  __builtin_memcpy(&ws, ws_blob, sizeof(Workspace));
  return TSAMD_DISPATCH_DTYPE_ALL(dtype, [&]() -> int {
    return dispatch_spmm<scalar_t>(reduce, vec, rowptr, col, value, mat, out, arg_out, B, M, N, K, E, ws, stream, ev);
  });
}
#endif
#undef TSAMD_BRIDGE_ARGS
}  // namespace tsamd

#if TSAMD_SPMM_TU <= 1  // every entry point lives in the main unit
using namespace tsamd;

#if !TSAMD_SPMM_PARTIAL_BUILD
extern "C" size_t tsamd_spmm_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M,
                                             int64_t N, int64_t K, int64_t E) {
  if (dtype_size(dtype) == 0 || B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return 0;
  return carve(nullptr, dtype, reduce, B, M, N, K, E, nullptr);
}

#endif

struct PartialOpts {  // tsamd_spmm_partial (see Workspace)
  int accumulate;
  const int64_t *arg_map;
  int64_t arg_none;
  const int64_t *deg_rowptr;
};

static int spmm_entry(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                      const void *value, const void *mat, void *out, int64_t *arg_out, int64_t B,
                      int64_t M, int64_t N, int64_t K, int64_t E, void *workspace,
                      size_t workspace_bytes_given, hipStream_t stream, hipEvent_t *ev,
                      bool relabelled = false, const int64_t *perm = nullptr,
                      const uint32_t *wmask = nullptr, void *cache = nullptr, size_t cache_bytes = 0,
                      int cache_valid = 0, const PartialOpts *partial = nullptr, bool arg32 = false,
                      uint32_t *rec_out = nullptr) {
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (reduce < TSAMD_SUM || reduce > TSAMD_MAX) return TSAMD_ERR_UNSUPPORTED;
  if (dtype_size(dtype) == 0) return TSAMD_ERR_UNSUPPORTED;
  if (N >= (int64_t)1 << 32 || K >= (int64_t)1 << 31 || B >= 65536) return TSAMD_ERR_UNSUPPORTED;
  if (B * ceil_div(K, 64) >= 65536) return TSAMD_ERR_UNSUPPORTED;  // gridDim.y = B * feature tiles
  const bool minmax = reduce == TSAMD_MIN || reduce == TSAMD_MAX;
  if (B * M * K == 0) return TSAMD_OK;  // nothing to write
  if (!rowptr || !out || (E > 0 && (!col || !mat)) || (minmax && !arg_out))
    return TSAMD_ERR_INVALID;
  // verification mode (tsamd_spmm_reference_order): the plain product in the reference's order of operations
  if (spmm_reference_order_on() && !relabelled && perm == nullptr && wmask == nullptr && partial == nullptr)
    return spmm_reference_order_run(dtype, reduce, rowptr, col, value, mat, out, arg_out, arg32, B, M, N, K, E, stream);
  // operand cache: the relabelled copy and the probe verdict live in the caller's buffer, not in the workspace
  const size_t cache_need = operand_cache_bytes(dtype, reduce, B, N, K, E);
  const bool use_cache = cache != nullptr && !relabelled && cache_need > 0 && cache_bytes >= cache_need &&
                         ((uintptr_t)cache % 256) == 0 && ((uintptr_t)mat % 16) == 0;
  const bool no_xperm_in_ws = relabelled || use_cache || partial != nullptr;
  const size_t need = carve(nullptr, dtype, reduce, B, M, N, K, E, nullptr, no_xperm_in_ws);
  if (!workspace || workspace_bytes_given < need) return TSAMD_ERR_WORKSPACE;
  if ((uintptr_t)workspace % 256 != 0) return TSAMD_ERR_WORKSPACE;
  Workspace ws;
  carve(workspace, dtype, reduce, B, M, N, K, E, &ws, no_xperm_in_ws);
  if (use_cache) {
    char *cb = reinterpret_cast<char *>(cache);
    ws.relabel_flag = reinterpret_cast<int *>(cb);
    ws.cache_fp = reinterpret_cast<unsigned long long *>(cb + 256);
    ws.xperm = cb + kOperandCacheHeader;
    ws.cache_state = cache_valid ? 2 : 1;
  }
  if (relabelled) {
    if (M >= (int64_t)1 << 32) return TSAMD_ERR_UNSUPPORTED;
    ws.out_relabel = 1;
  }
  ws.perm = perm;
  if (arg32) {  // E itself ("no winner") must fit a non-negative int32
    if (!minmax || partial != nullptr || E >= (int64_t)1 << 31) return TSAMD_ERR_UNSUPPORTED;
    ws.arg32 = 1;
    if (rec_out != nullptr) {  // (tsamd_spmm_minmax_records checked the shape: spmm_emits_records)
      ws.rec_out = rec_out;
      ws.rec_value = value;
      ws.rec_stride = win_record_stride(K);
      ws.rec_meta = (uint32_t)ceil_div(K, 32);
      ws.rec_has_z = ((ws.rec_meta + 3u) & 3u) != 0u ? 1 : 0;
      ws.snap = TSAMD_RECORD_SNAP < ws.items / 2 ? TSAMD_RECORD_SNAP : ws.items / 2;
    }
  }
  // min / max: partition boundaries snap to row starts (spmm_partition_kernel) -- fewer cut rows, fewer carry records and
  // fix-up waves: configs[2] forward 1.024-1.029 -> 1.000 ms (bf16), 1.57 -> 1.55 (fp32), same box
  // (profiles/r06_ab_minmax_snap.jsonl).  The result does not depend on where a row is cut (no rounding in min / max);
  // sums keep their partition: theirs does, in the last bits.
  if (minmax && ws.snap == 0) ws.snap = TSAMD_RECORD_SNAP < ws.items / 2 ? TSAMD_RECORD_SNAP : ws.items / 2;
  if (partial != nullptr) {
    if (reduce == TSAMD_MEAN && partial->deg_rowptr == nullptr) return TSAMD_ERR_INVALID;
    ws.partial = 1;
    ws.accumulate = partial->accumulate ? 1 : 0;
    ws.arg_map = partial->arg_map;
    ws.arg_none = partial->arg_none;
    ws.deg_rowptr = reduce == TSAMD_MEAN ? partial->deg_rowptr : nullptr;
  }
  if (wmask != nullptr) {
    if (E >= (int64_t)1 << 32 || reduce != TSAMD_SUM) return TSAMD_ERR_UNSUPPORTED;  // 32-bit entry ids in the windows
    ws.wmask = wmask;
    ws.rec_stride = win_record_stride(K);
    ws.rec_meta = (uint32_t)ceil_div(K, 32);
    ws.rec_has_z = (ws.rec_meta <= 32u && ((ws.rec_meta + 3u) & 3u) != 0u) ? 1 : 0;  // a padding word behind (id, value)
  }
  const size_t es = dtype_size(dtype);
  int vec = es <= 2 ? 4 : (int)(16 / es);  // widest packet for the type (see dispatch_spmm)
  while (vec > 1 && !((K % vec) == 0 && ((uintptr_t)mat % (vec * es)) == 0 &&
                      ((uintptr_t)out % (vec * es)) == 0 &&
                      (!minmax || ((uintptr_t)arg_out % (vec * (arg32 ? 4 : 8))) == 0)))
    vec >>= 1;

#if TSAMD_SPMM_PARTIAL_BUILD
  // partial products are the stages of the sharded SpMM over dense FEATURE matrices: floating point only
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16) return TSAMD_ERR_UNSUPPORTED;
#endif
#if TSAMD_SPMM_TU == 1
  if (reduce == TSAMD_MIN)  // instantiated in spmm_min.hip / spmm_max.hip
    return tsamd::spmm_min_bridge(dtype, reduce, vec, rowptr, col, value, mat, out, arg_out, B, M, N, K, E, &ws,
                                  sizeof(ws), stream, ev);
  if (reduce == TSAMD_MAX)
    return tsamd::spmm_max_bridge(dtype, reduce, vec, rowptr, col, value, mat, out, arg_out, B, M, N, K, E, &ws,
                                  sizeof(ws), stream, ev);
#endif
  return TSAMD_DISPATCH_DTYPE_ALL(dtype, [&]() -> int {
#if TSAMD_SPMM_PARTIAL_BUILD
    if constexpr (!(std::is_same<scalar_t, float>::value || std::is_same<scalar_t, double>::value ||
                    std::is_same<scalar_t, f16_t>::value || std::is_same<scalar_t, bf16_t>::value))
      return (int)TSAMD_ERR_UNSUPPORTED;
    else
#endif
    return dispatch_spmm<scalar_t>(reduce, vec, rowptr, col, value, mat, out, arg_out, B, M, N,
                                   K, E, ws, stream, ev);
  });
}

#if !TSAMD_SPMM_PARTIAL_BUILD
extern "C" int tsamd_spmm(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                          const void *value, const void *mat, void *out, int64_t *arg_out,
                          int64_t B, int64_t M, int64_t N, int64_t K, int64_t E, void *workspace,
                          size_t workspace_bytes_given, void *stream_) {
  return spmm_entry(dtype, reduce, rowptr, col, value, mat, out, arg_out, B, M, N, K, E, workspace,
                    workspace_bytes_given, reinterpret_cast<hipStream_t>(stream_), nullptr);
}

// ---------------------------------------------------------------------------
// operand cache: see include/tsamd.h
// ---------------------------------------------------------------------------
extern "C" size_t tsamd_spmm_operand_cache_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N,
                                                 int64_t K, int64_t E) {
  (void)M;
  if (dtype_size(dtype) == 0 || B < 0 || N < 0 || K < 0 || E < 0) return 0;
  return operand_cache_bytes(dtype, reduce, B, N, K, E);
}

extern "C" size_t tsamd_spmm_cached_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N,
                                                    int64_t K, int64_t E) {
  if (dtype_size(dtype) == 0 || B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return 0;
  return carve(nullptr, dtype, reduce, B, M, N, K, E, nullptr, operand_cache_bytes(dtype, reduce, B, N, K, E) > 0);
}

extern "C" int tsamd_spmm_cached(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                                 const void *value, const void *mat, void *out, int64_t *arg_out, int64_t B,
                                 int64_t M, int64_t N, int64_t K, int64_t E, void *workspace,
                                 size_t workspace_bytes_given, void *cache, size_t cache_bytes, int cache_valid,
                                 void *stream_) {
  return spmm_entry(dtype, reduce, rowptr, col, value, mat, out, arg_out, B, M, N, K, E, workspace,
                    workspace_bytes_given, reinterpret_cast<hipStream_t>(stream_), nullptr, false, nullptr, nullptr,
                    cache, cache_bytes, cache_valid);
}

// min / max with the winners as 32-bit entry ids (include/tsamd.h); cache == nullptr: stateless
extern "C" int tsamd_spmm_minmax_arg32(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                                       const void *value, const void *mat, void *out, int32_t *arg_out32, int64_t B,
                                       int64_t M, int64_t N, int64_t K, int64_t E, void *workspace,
                                       size_t workspace_bytes_given, void *cache, size_t cache_bytes, int cache_valid,
                                       void *stream_) {
  if (reduce != TSAMD_MIN && reduce != TSAMD_MAX) return TSAMD_ERR_UNSUPPORTED;
  return spmm_entry(dtype, reduce, rowptr, col, value, mat, out, reinterpret_cast<int64_t *>(arg_out32), B, M, N, K,
                    E, workspace, workspace_bytes_given, reinterpret_cast<hipStream_t>(stream_), nullptr, false,
                    nullptr, nullptr, cache, cache_bytes, cache_valid, nullptr, true);
}

// min / max whose forward leaves the winner RECORDS of the pull backward instead of the winner ids (include/tsamd.h).
// The merge kernel writes the records of the rows it finishes by itself (Workspace::rec_out) when the row shape allows
// it; the rows cut between partitions -- or every row, when it does not -- get theirs from the ids
// (minmax_winrec_kernel, csrc/spmm_bw.hip), which only live in the workspace.
static bool spmm_emits_records(int dtype, int64_t B, int64_t M, int64_t K, int64_t E, const void *mat, const void *out) {
  if (dtype != TSAMD_F32 && dtype != TSAMD_F16 && dtype != TSAMD_BF16) return false;
  if (K <= 32 || K > 256 || K % 4 != 0 || E >= (int64_t)1 << 31 || M >= (int64_t)1 << 32 || B < 1) return false;
  if (dtype == TSAMD_F32 && K <= 64) return false;  // measured even (profiles/r06_ab_fwd_winrec.md): the ids stay
  const size_t packet = 4 * dtype_size(dtype);  // the four-element packets the record writer's lane layout assumes
  if (mat != nullptr && (((uintptr_t)mat % packet) != 0 || ((uintptr_t)out % packet) != 0)) return false;
  return !spmm_reference_order_on();
}

static size_t records_arg_bytes(int64_t B, int64_t M, int64_t K) { return align_up(sizeof(int32_t) * (size_t)(B * M * K), 256); }

extern "C" int tsamd_spmm_minmax_records_in_forward(int dtype, int64_t B, int64_t M, int64_t K, int64_t E) {
  return spmm_emits_records(dtype, B, M, K, E, nullptr, nullptr) ? 1 : 0;
}

extern "C" size_t tsamd_spmm_minmax_records_bytes(int64_t B, int64_t K, int64_t E) {
  if (B < 0 || K < 0 || E < 0) return 0;
  return align_up(sizeof(uint32_t) * (size_t)(B * E) * win_record_stride(K), 256);
}

// (the ids only exist -- in the workspace -- for the shapes whose records the forward does not write itself)
extern "C" size_t tsamd_spmm_minmax_records_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N,
                                                            int64_t K, int64_t E) {
  if (dtype_size(dtype) == 0 || B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return 0;
  return records_arg_bytes(B, M, K) + carve(nullptr, dtype, reduce, B, M, N, K, E, nullptr);
}

extern "C" int tsamd_spmm_minmax_records(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                                         const void *value, const void *mat, void *out, const int64_t *row,
                                         uint32_t *records, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                                         void *workspace, size_t workspace_bytes_given, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (reduce != TSAMD_MIN && reduce != TSAMD_MAX) return TSAMD_ERR_UNSUPPORTED;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16) return TSAMD_ERR_UNSUPPORTED;
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (E >= (int64_t)1 << 31 || M >= (int64_t)1 << 32) return TSAMD_ERR_UNSUPPORTED;
  if (B * M * K == 0) return TSAMD_OK;
  if (E > 0 && (!records || !row)) return TSAMD_ERR_INVALID;
  const size_t arg_b = records_arg_bytes(B, M, K);
  if (!workspace || workspace_bytes_given < tsamd_spmm_minmax_records_workspace_bytes(dtype, reduce, B, M, N, K, E) ||
      (uintptr_t)workspace % 256 != 0)
    return TSAMD_ERR_WORKSPACE;
  char *w = reinterpret_cast<char *>(workspace);
  int32_t *arg32 = reinterpret_cast<int32_t *>(w);
  void *inner = w + arg_b;
  const bool emit = E > 0 && spmm_emits_records(dtype, B, M, K, E, mat, out);
  int st = spmm_entry(dtype, reduce, rowptr, col, value, mat, out, reinterpret_cast<int64_t *>(arg32), B, M, N, K, E,
                      inner, workspace_bytes_given - arg_b, stream, nullptr, false, nullptr, nullptr, nullptr, 0, 0,
                      nullptr, true, emit ? records : nullptr);
  if (st != TSAMD_OK || E == 0 || emit) return st;
  return minmax_winrec_from_ids(dtype, row, value, arg32, records, B, M, K, E, stream);
}

// ---------------------------------------------------------------------------
// relabelled ("camping-free") layout end to end: see include/tsamd.h
// ---------------------------------------------------------------------------
namespace tsamd {
namespace {
struct RelabelParams {
  uint32_t bits, mul, shift;
};
RelabelParams relabel_params(int64_t n) {
  RelabelParams p;
  p.bits = 1;
  while (p.bits < 32 && ((uint64_t)1 << p.bits) < (uint64_t)(n > 1 ? n : 2)) ++p.bits;
  p.mul = 0x9E3779B1u;
  p.shift = p.bits > 1 ? p.bits / 2 : 1;
  return p;
}
__global__ void relabel_ids_kernel(const int64_t *__restrict__ ids, int64_t count, uint32_t n,
                                   RelabelParams p, int64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t v = ids ? ids[i] : i;
  out[i] = (v < 0 || v >= (int64_t)n) ? v : (int64_t)hash_row((uint32_t)v, n, p.bits, p.mul, p.shift);
}
}  // namespace
}  // namespace tsamd

// ---------------------------------------------------------------------------
// dst[i, :] = src[idx[i], :] for row-major matrices of `row_bytes`-byte rows: the pack step in front
// of a row exchange (pytorch_sparse_amd/parallel.py).  One packet (16 / 8 / 4 / 2 / 1 bytes, the
// widest the pitch and the pointers allow) per lane, 2^lgL lanes per row.
// ---------------------------------------------------------------------------
namespace tsamd {
namespace {
template <typename P>
__global__ __launch_bounds__(256) void gather_rows_kernel(const P *__restrict__ src,
                                                          const int64_t *__restrict__ idx,
                                                          P *__restrict__ dst, int64_t n, int64_t n_src,
                                                          uint32_t slots, int lgL) {
  const uint32_t lanes = 1u << lgL;
  const uint32_t sl0 = threadIdx.x & (lanes - 1);
  const int64_t rows_per_block = 256 >> lgL;
  for (int64_t r = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> lgL); r < n;
       r += (int64_t)gridDim.x * rows_per_block) {
    int64_t j = idx[r];
    if (j < 0) j += n_src;  // torch indexing convention; the caller guarantees the range
    const P *s = src + (uint64_t)j * slots;
    P *d = dst + (uint64_t)r * slots;
    for (uint32_t sl = sl0; sl < slots; sl += lanes) d[sl] = s[sl];
  }
}

template <typename P>
int launch_gather_rows(const void *src, const int64_t *idx, void *dst, int64_t n, int64_t n_src,
                       int64_t row_bytes, hipStream_t stream) {
  const uint32_t slots = (uint32_t)(row_bytes / (int64_t)sizeof(P));
  int lgL = 0;
  while (lgL < 8 && (1u << lgL) < slots) ++lgL;
  const int64_t rows_per_block = 256 >> lgL;
  int64_t blocks = ceil_div(n, rows_per_block);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL((gather_rows_kernel<P>), dim3((unsigned int)blocks), dim3(256), 0, stream,
                     reinterpret_cast<const P *>(src), idx, reinterpret_cast<P *>(dst), n, n_src, slots, lgL);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}
}  // namespace
}  // namespace tsamd

extern "C" int tsamd_gather_rows(const void *src, const int64_t *idx, void *dst, int64_t n,
                                 int64_t n_src, int64_t row_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || n_src < 0 || row_bytes < 0 || row_bytes >= (int64_t)1 << 32) return TSAMD_ERR_INVALID;
  if (n == 0 || row_bytes == 0) return TSAMD_OK;
  if (!src || !idx || !dst) return TSAMD_ERR_INVALID;
  const uintptr_t a = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)row_bytes;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  if (a % 16 == 0) return launch_gather_rows<u32x4>(src, idx, dst, n, n_src, row_bytes, stream);
  if (a % 8 == 0) return launch_gather_rows<u32x2>(src, idx, dst, n, n_src, row_bytes, stream);
  if (a % 4 == 0) return launch_gather_rows<uint32_t>(src, idx, dst, n, n_src, row_bytes, stream);
  if (a % 2 == 0) return launch_gather_rows<uint16_t>(src, idx, dst, n, n_src, row_bytes, stream);
  return launch_gather_rows<uint8_t>(src, idx, dst, n, n_src, row_bytes, stream);
}

extern "C" int tsamd_relabel_ids(const int64_t *ids, int64_t count, int64_t n, int64_t *out,
                                 void *stream_) {
  if (count < 0 || n < 0 || n >= (int64_t)1 << 32) return TSAMD_ERR_UNSUPPORTED;
  if (count == 0) return TSAMD_OK;
  if (!out) return TSAMD_ERR_INVALID;
  hipLaunchKernelGGL(relabel_ids_kernel, dim3((unsigned int)ceil_div(count, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream_), ids, count, (uint32_t)n, relabel_params(n), out);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

extern "C" size_t tsamd_spmm_relabelled_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M,
                                                        int64_t N, int64_t K, int64_t E) {
  if (dtype_size(dtype) == 0 || B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return 0;
  return carve(nullptr, dtype, reduce, B, M, N, K, E, nullptr, true);
}

extern "C" int tsamd_spmm_relabelled(int dtype, int reduce, const int64_t *rowptr,
                                     const int64_t *col_h, const void *value, const void *mat_h,
                                     void *out_h, int64_t *arg_out_h, int64_t B, int64_t M, int64_t N,
                                     int64_t K, int64_t E, void *workspace,
                                     size_t workspace_bytes_given, void *stream_) {
  return spmm_entry(dtype, reduce, rowptr, col_h, value, mat_h, out_h, arg_out_h, B, M, N, K, E, workspace,
                    workspace_bytes_given, reinterpret_cast<hipStream_t>(stream_), nullptr, true);
}

#else  // TSAMD_SPMM_PARTIAL_BUILD
extern "C" size_t tsamd_spmm_partial_workspace_bytes(int dtype, int reduce, int64_t B, int64_t M, int64_t N,
                                                     int64_t K, int64_t E) {
  if (dtype_size(dtype) == 0 || B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return 0;
  return carve(nullptr, dtype, reduce, B, M, N, K, E, nullptr, true);
}

extern "C" int tsamd_spmm_partial(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                                  const void *value, const void *mat, void *out, int64_t *arg_out, int64_t B,
                                  int64_t M, int64_t N, int64_t K, int64_t E, const int64_t *arg_map,
                                  int64_t arg_none, int accumulate, const int64_t *deg_rowptr, void *workspace,
                                  size_t workspace_bytes_given, void *stream_) {
  PartialOpts po{accumulate, arg_map, arg_none, deg_rowptr};
  // E == 0 with accumulate: nothing to add; without: the "empty so far" state has to be written
  if (accumulate && E == 0 && reduce != TSAMD_MEAN) return TSAMD_OK;
  return spmm_entry(dtype, reduce, rowptr, col, value, mat, out, arg_out, B, M, N, K, E, workspace,
                    workspace_bytes_given, reinterpret_cast<hipStream_t>(stream_), nullptr, false, nullptr, nullptr,
                    nullptr, 0, 0, &po);
}

#endif

#if !TSAMD_SPMM_PARTIAL_BUILD
extern "C" int tsamd_spmm_permuted(int dtype, int reduce, const int64_t *rowptr, const int64_t *col,
                                   const void *value, const int64_t *perm, const void *mat, void *out,
                                   int64_t *arg_out, int64_t B, int64_t M, int64_t N, int64_t K,
                                   int64_t E, void *workspace, size_t workspace_bytes_given,
                                   void *stream_) {
  if (E > 0 && !perm) return TSAMD_ERR_INVALID;
  return spmm_entry(dtype, reduce, rowptr, col, value, mat, out, arg_out, B, M, N, K, E, workspace,
                    workspace_bytes_given, reinterpret_cast<hipStream_t>(stream_), nullptr, false, perm);
}

// internal (spmm_internal.h): the masked sum behind tsamd_spmm_minmax_bw_csc
namespace tsamd {
size_t spmm_masked_sum_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E) {
  return carve(nullptr, dtype, TSAMD_SUM, B, M, N, K, E, nullptr);
}
int spmm_masked_sum(int dtype, const int64_t *rowptr, bool has_value, const int64_t *perm,
                    const uint32_t *records, const void *mat, void *out, int64_t B, int64_t M, int64_t N,
                    int64_t K, int64_t E, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  if (E > 0 && !records) return TSAMD_ERR_INVALID;
  // `col` / `value` are only tested against NULL in the masked kernel (their contents come from the records)
  const int64_t *col = reinterpret_cast<const int64_t *>(records);
  const void *value = has_value ? reinterpret_cast<const void *>(records) : nullptr;
  return spmm_entry(dtype, TSAMD_SUM, rowptr, col, value, mat, out, nullptr, B, M, N, K, E, workspace,
                    workspace_bytes, stream, nullptr, false, perm, records);
}
}  // namespace tsamd

extern "C" int tsamd_spmm_profiled(int dtype, int reduce, const int64_t *rowptr,
                                   const int64_t *col, const void *value, const void *mat,
                                   void *out, int64_t *arg_out, int64_t B, int64_t M, int64_t N,
                                   int64_t K, int64_t E, void *workspace,
                                   size_t workspace_bytes_given, void *stream_,
                                   float *kernel_ms_host) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!kernel_ms_host) return TSAMD_ERR_INVALID;
  hipEvent_t ev[4];
  for (int i = 0; i < 4; ++i) TSAMD_HIP_TRY(hipEventCreate(&ev[i]));
  kernel_ms_host[0] = kernel_ms_host[1] = kernel_ms_host[2] = 0.f;
  int st = spmm_entry(dtype, reduce, rowptr, col, value, mat, out, arg_out, B, M, N, K, E,
                      workspace, workspace_bytes_given, stream, ev);
  if (st == TSAMD_OK && B * M * K > 0) {
    TSAMD_HIP_TRY(hipEventSynchronize(ev[3]));
    for (int i = 0; i < 3; ++i)
      TSAMD_HIP_TRY(hipEventElapsedTime(&kernel_ms_host[i], ev[i], ev[i + 1]));
  }
  for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
  return st;
}
#endif  // !TSAMD_SPMM_PARTIAL_BUILD
#endif  // TSAMD_SPMM_TU <= 1
