// Backward kernels of CSR SpMM for gfx950 (MI355X).
//
//  * tsamd_spmm_value_bw  -- gradient of SUM/MEAN SpMM w.r.t. the sparse values, an SDDMM
//    over the pattern.  Replaces spmm_value_bw_cuda / spmm_value_bw_cpu of the reference
//    (csrc/cuda/spmm_cuda.cu:157-237, csrc/cpu/spmm_cpu.cpp:103-152).
//  * tsamd_spmm_minmax_bw -- backward of MIN/MAX SpMM.  Replaces the ATen composition
//    (masked_fill / index_select / gather / scatter_add_) in SPMMMin/SPMMMax::backward
//    (csrc/spmm.cpp:204-242, 264-302) with one fused pass.
#include "common.h"
#include "spmm_internal.h"

#include <cstdlib>
#include <type_traits>

#ifndef TSAMD_MASKED_SEGMENT_SKIP
#define TSAMD_MASKED_SEGMENT_SKIP 1  // 0: the masked SDDMM / masked sum gather every entry's whole rows (round 3)
#endif

namespace tsamd {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kUnroll = 2;

// ---------------------------------------------------------------------------
// value gradient: edge-parallel (perfectly balanced whatever the degrees).
// A wave owns 64 consecutive edges; (row, col) ids are read once, coalesced.
// G = 64/LPR groups of LPR lanes x VEC features each take one edge per step:
// two 16-byte gathers per lane (mat[col], grad[row]), a VEC-wide dot, then a
// butterfly over the LPR lanes.  Consecutive edges share `row`, so the grad
// row is served by L1/L2 after its first touch.  Results are moved to the lane
// that owns the edge and stored with one coalesced write per window.
// ---------------------------------------------------------------------------
// MASKED: only the features whose bit is set in the entry's winner record (spmm_internal.h) enter the
// dot product -- grad_value of the min/max backward as an SDDMM, for callers that built the records.
// Masked dot product of one 16-byte packet of 2-byte features (8 of them), fp32 accumulation: the winner bits are
// widened to a 16-bit lane mask per dword (two v_bfe_i32 + one v_bfi), BOTH operands are ANDed with it (a feature
// that did not win must not contribute at all: 0 * Inf would be a NaN) and gfx950's v_dot2c_f32_{bf16,f16} takes the
// pair -- conversion, two multiplies and two adds in one instruction.  24 VALU instructions per packet instead of
// ~45 (two conversions + select + FMA per element): the masked SDDMM ran at 0.94 of the VALU slots
// (profiles/r04_sq_counters.md).  -DTSAMD_MASKED_DOT2=0 keeps the per-element form.
#ifndef TSAMD_MASKED_DOT2
#define TSAMD_MASKED_DOT2 1
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ float masked_dot8(const u32x4_t &x, const u32x4_t &y, uint32_t bits, float acc) {
  static_assert(sizeof(T) == 2, "2-byte features");
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * d, 1);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * d + 1, 1);
    const uint32_t m = (lo & 0xFFFFu) | (hi & 0xFFFF0000u);
    const uint32_t xm = x[d] & m, ym = y[d] & m;
    if constexpr (std::is_same<T, bf16_t>::value) {
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      bf16x2_t a, b;
      __builtin_memcpy(&a, &xm, 4);
      __builtin_memcpy(&b, &ym, 4);
      acc = __builtin_amdgcn_fdot2_f32_bf16(a, b, acc, false);
    } else {
      typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
      f16x2_t a, b;
      __builtin_memcpy(&a, &xm, 4);
      __builtin_memcpy(&b, &ym, 4);
      acc = __builtin_amdgcn_fdot2(a, b, acc, false);
    }
  }
  return acc;
}

template <typename T, int VEC, bool MASKED = false>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_value_bw_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ rowptr,
    const int64_t *__restrict__ col, const T *__restrict__ mat, const T *__restrict__ grad,
    T *__restrict__ out, int64_t B, int64_t M, int64_t N, uint32_t K, int64_t E, int lgG,
    bool mean, const uint32_t *__restrict__ rec = nullptr, uint32_t rec_stride = 0) {
  using A = typename Traits<T>::acc_t;
  using P = Pack<T, VEC>;
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wib) * kWave;
  if (base >= E) return;
  const int64_t rem = E - base;
  const int n = rem < kWave ? (int)rem : kWave;

  // this lane's edge
  uint32_t c_l = 0, r_l = 0;
  int64_t deg_l = 1;
  if (lane < n) {
    const int64_t e = base + lane;
    c_l = (uint32_t)col[e];
    int64_t r;
    if (row != nullptr) {
      r = row[e];
    } else {  // last r with rowptr[r] <= e
      int64_t lo = 0, hi = M;
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid;
      }
      r = lo;
    }
    r_l = (uint32_t)r;
    if (mean) {
      deg_l = rowptr[r + 1] - rowptr[r];
      if (deg_l < 1) deg_l = 1;
    }
  }

  const int G = 1 << lgG;
  const int lpr = 64 >> lgG;
  const int g = lane >> (6 - lgG);
  const int kl = lane & (lpr - 1);
  const uint32_t slots = (K + VEC - 1) / VEC;
  const int nsteps = (n + G - 1) >> lgG;

  A mine = A(0);  // result of the edge owned by this lane
  for (int s = 0; s < nsteps; s += kUnroll) {
    A acc[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int idx = ((s + u) << lgG) + g;
      const int src = idx < n ? idx : n - 1;
      const uint32_t c = lane_read(c_l, src);
      const uint32_t r = lane_read(r_l, src);
      acc[u] = A(0);
      for (int64_t b = 0; b < B; ++b) {
        const T *mrow = mat + ((uint64_t)b * N + c) * K;
        const T *grow = grad + ((uint64_t)b * M + r) * K;
        for (uint32_t sl = kl; sl < slots; sl += lpr) {
#if TSAMD_MASKED_SEGMENT_SKIP
          if constexpr (MASKED) {
            // The record word first: an entry of a row of degree d wins a feature with probability ~1/d, and the
            // entries of a hub row are consecutive in this kernel's (CSR) order -- most slots of most steps hold no
            // winner at all.  A lane whose VEC features have none sits out its two 16-byte gathers, and a slot in
            // which NO lane of the wave has one is skipped altogether (the kernel is bound by VALU issue: 0.94 of
            // the slots at configs[2], profiles/r04_sq_counters.md).  Adding nothing == adding the masked zeros.
            const uint32_t f0 = sl * VEC;
            const uint32_t word = rec[((uint64_t)b * (uint64_t)E + (uint64_t)(base + src)) * rec_stride + (f0 >> 5)];
            const uint32_t bits = (word >> (f0 & 31u)) & (VEC >= 32 ? 0xFFFFFFFFu : ((1u << (VEC & 31)) - 1u));
            if (__ballot(bits != 0u) == 0ull) continue;  // wave-uniform (every lane runs the same sl sequence)
            P x{}, y{};
            if (bits != 0u) {
              x = *reinterpret_cast<const P *>(mrow + (uint64_t)sl * VEC);
              y = *reinterpret_cast<const P *>(grow + (uint64_t)sl * VEC);
            }
            {
              typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
              static_assert(sizeof(P) == 16, "masked SDDMM works on 16-byte packets");
              u32x4 xb, yb;
              __builtin_memcpy(&xb, &x, 16);
              __builtin_memcpy(&yb, &y, 16);
              asm volatile("" : "+v"(xb), "+v"(yb));
#if TSAMD_MASKED_DOT2
              if constexpr (sizeof(T) == 2 && VEC == 8 && std::is_same<A, float>::value) {
                acc[u] = masked_dot8<T>(xb, yb, bits, acc[u]);
                continue;
              }
#endif
              __builtin_memcpy(&x, &xb, 16);
              __builtin_memcpy(&y, &yb, 16);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j)
              acc[u] += ((bits >> j) & 1u) ? Traits<T>::to_acc(x.v[j]) * Traits<T>::to_acc(y.v[j]) : A(0);
            continue;
          }
#endif
          P x = *reinterpret_cast<const P *>(mrow + (uint64_t)sl * VEC);
          P y = *reinterpret_cast<const P *>(grow + (uint64_t)sl * VEC);
          if constexpr (MASKED) {
            // keep the two 16-byte loads whole: without the barrier the compiler sinks them into the
            // per-element selects below as 2-byte loads (measured 1.8 ms instead of 0.7 ms)
            {
              typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
              static_assert(sizeof(P) == 16, "masked SDDMM works on 16-byte packets");
              u32x4 xb, yb;
              __builtin_memcpy(&xb, &x, 16);
              __builtin_memcpy(&yb, &y, 16);
              asm volatile("" : "+v"(xb), "+v"(yb));
              __builtin_memcpy(&x, &xb, 16);
              __builtin_memcpy(&y, &yb, 16);
            }
            // VEC divides 32: the packet's bits sit in one word of the record
            const uint32_t f0 = sl * VEC;
            const uint32_t bits = rec[((uint64_t)b * (uint64_t)E + (uint64_t)(base + src)) * rec_stride + (f0 >> 5)] >> (f0 & 31u);
#pragma unroll
            for (int j = 0; j < VEC; ++j)
              acc[u] += ((bits >> j) & 1u) ? Traits<T>::to_acc(x.v[j]) * Traits<T>::to_acc(y.v[j]) : A(0);
          } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
              acc[u] += Traits<T>::to_acc(x.v[j]) * Traits<T>::to_acc(y.v[j]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      A v = acc[u];
      for (int off = lpr >> 1; off > 0; off >>= 1) v += lane_xor(v, off);
      // slot (s+u)*G + g' was computed by group g'; its owner lane fetches it
      const int owner_step = lane >> lgG;          // step in which this lane's edge is handled
      const int owner_group = lane & (G - 1);
      const A got = lane_read(v, owner_group << (6 - lgG));
      if (owner_step == s + u) mine = got;
    }
  }
  if (lane < n) {
    if (mean) mine = mine / (A)deg_l;
    out[base + lane] = Traits<T>::from_acc(mine);
  }
}

// ---------------------------------------------------------------------------
// Masked SDDMM, pipelined (round 5): grad_value of the min / max pull backward from the winner records.
// The kernel above takes a step as "record word -> (wait) -> the two 16-byte gathers -> (wait) -> dot product": two
// dependent round trips per step and lane group, and with the skip branches in between the compiler cannot overlap
// consecutive steps -- at configs[2] a wave spent ~1.4 us per step, i.e. it was bound by those latencies (halving
// the VALU work of the dot product with v_dot2c moved the op by 2 %: 2.24 -> 2.20 ms).  Here a lane fetches the
// record words of kMaskedChunk steps in ONE round trip, then takes the steps in pairs: both steps' gathers are
// issued before the first dot product.  Same arithmetic as the kernel above (fp32 / fp64 accumulation, masked
// elements contribute nothing), one packet per lane and step, so it needs K / VEC <= 64 packets per row
// (K <= 512 two-byte, 256 fp32, 128 fp64 features); wider rows keep the kernel above.
// Sum over a group of lpr consecutive lanes, result in every lane of the group: DPP inside a row of 16 lanes
// (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror), ds_bpermute above.
// ---------------------------------------------------------------------------
constexpr int kMaskedChunk = 8;

__device__ __forceinline__ float group_sum(float v, int lpr) {
  if (lpr >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if (lpr >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if (lpr >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  if (lpr >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)); // row_mirror
  if (lpr >= 32) v += lane_xor(v, 16);
  if (lpr >= 64) v += lane_xor(v, 32);
  return v;
}
__device__ __forceinline__ double group_sum(double v, int lpr) {
  for (int off = lpr >> 1; off > 0; off >>= 1) v += lane_xor(v, off);
  return v;
}

template <typename T>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_value_bw_masked_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
    const T *__restrict__ mat, const T *__restrict__ grad, T *__restrict__ out, int64_t B, int64_t M, int64_t N,
    uint32_t K, int64_t E, int lgG, const uint32_t *__restrict__ rec, uint32_t rec_stride) {
  using A = typename Traits<T>::acc_t;
  constexpr int VEC = 16 / (int)sizeof(T);
  using P = Pack<T, VEC>;
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wib) * kWave;
  if (base >= E) return;
  const int64_t rem = E - base;
  const int n = rem < kWave ? (int)rem : kWave;
  uint32_t c_l = 0, r_l = 0;
  if (lane < n) {
    const int64_t e = base + lane;
    c_l = (uint32_t)col[e];
    int64_t r;
    if (row != nullptr) {
      r = row[e];
    } else {  // last r with rowptr[r] <= e
      int64_t lo = 0, hi = M;
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid;
      }
      r = lo;
    }
    r_l = (uint32_t)r;
  }
  const int G = 1 << lgG, lpr = 64 >> lgG;
  const int g = lane >> (6 - lgG), kl = lane & (lpr - 1);
  const uint32_t slots = K / VEC;  // the launcher guarantees K % VEC == 0 and slots <= lpr
  const bool slot_ok = (uint32_t)kl < slots;
  const uint32_t f0 = (uint32_t)kl * VEC;
  const uint32_t wsel = f0 >> 5, sh = f0 & 31u;
  constexpr uint32_t kBitsMask = VEC >= 32 ? 0xFFFFFFFFu : ((1u << (VEC & 31)) - 1u);
  const int nsteps = (n + G - 1) >> lgG;
  const int owner_step = lane >> lgG, owner_src = (lane & (G - 1)) << (6 - lgG);
  A mine = A(0);  // result of the edge owned by this lane
  for (int64_t b = 0; b < B; ++b) {
    const T *mat_b = mat + (uint64_t)b * (uint64_t)N * K + f0;
    const T *grad_b = grad + (uint64_t)b * (uint64_t)M * K + f0;
    const uint32_t *rec_b = rec + ((uint64_t)b * (uint64_t)E + (uint64_t)base) * rec_stride + wsel;
    for (int t0 = 0; t0 < nsteps; t0 += kMaskedChunk) {
      uint32_t bits[kMaskedChunk];
#pragma unroll
      for (int u = 0; u < kMaskedChunk; ++u) {  // the chunk's record words: one round trip
        const int idx = ((t0 + u) << lgG) + g;
        uint32_t w = 0;
        if (slot_ok && idx < n) w = rec_b[(uint32_t)idx * rec_stride];
        bits[u] = w;
      }
#pragma unroll
      for (int u = 0; u < kMaskedChunk; ++u) bits[u] = (bits[u] >> sh) & kBitsMask;
#pragma unroll
      for (int u = 0; u < kMaskedChunk; u += 2) {
        if (t0 + u >= nsteps) break;                                           // wave-uniform
        if (__ballot((bits[u] | bits[u + 1]) != 0u) == 0ull) continue;         // no winner in either step: adds nothing
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4 xb[2], yb[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int idx = ((t0 + u + v) << lgG) + g;
          const int src = idx < n ? idx : n - 1;
          const uint32_t c = lane_read(c_l, src);
          const uint32_t r = lane_read(r_l, src);
          xb[v] = u32x4{0u, 0u, 0u, 0u};
          yb[v] = u32x4{0u, 0u, 0u, 0u};
          if (bits[u + v] != 0u) {
            xb[v] = *reinterpret_cast<const u32x4 *>(mat_b + (uint64_t)c * K);
            yb[v] = *reinterpret_cast<const u32x4 *>(grad_b + (uint64_t)r * K);
          }
        }
        asm volatile("" : "+v"(xb[0]), "+v"(yb[0]), "+v"(xb[1]), "+v"(yb[1]));  // all four gathers in flight, whole
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          A acc = A(0);
          if constexpr (sizeof(T) == 2 && std::is_same<A, float>::value) {
            acc = masked_dot8<T>(xb[v], yb[v], bits[u + v], acc);
          } else {
            P x, y;
            __builtin_memcpy(&x, &xb[v], 16);
            __builtin_memcpy(&y, &yb[v], 16);
#pragma unroll
            for (int j = 0; j < VEC; ++j)
              acc += ((bits[u + v] >> j) & 1u) ? Traits<T>::to_acc(x.v[j]) * Traits<T>::to_acc(y.v[j]) : A(0);
          }
          acc = group_sum(acc, lpr);
          const A got = lane_read(acc, owner_src);
          if (owner_step == t0 + u + v) mine += got;
        }
      }
    }
  }
  if (lane < n) out[base + lane] = Traits<T>::from_acc(mine);
}

// ---------------------------------------------------------------------------
// min/max backward (csrc/spmm.cpp:204-242, 264-302 of the reference):
//   for every (b, m, k) with a = arg_out[b,m,k] != E:
//     grad_value[a]          += mat[b, col[a], k] * grad_out[b,m,k]
//     grad_mat[b, col[a], k] += value[a] * grad_out[b,m,k]
//
// grad_mat is a scatter into rows of OTHER nodes (the transposed pattern, which the op does not
// receive), so it needs hardware atomics, and on MI355X a device-scope atomic is priced per
// 64-BYTE SEGMENT an instruction touches: 49 ps each (20 G segments/s for the whole device),
// whether 1 or 16 lanes fall into it, fp32 and packed bf16 alike (scripts/ubench/atomics.hip,
// profiles/r02_ubench_atomics.csv).  The kernel is therefore laid out to touch every
// (output row, winning entry, 64-byte segment of the target row) exactly once:
//   * lane l of a wave owns feature k = tile * 64 + l of one output row, so that one atomic
//     instruction covers 64 CONSECUTIVE features (a thread-pair layout (2l, 2l+1) with one
//     instruction per pair member touches every segment twice: measured 3.4 vs 2.3 ms);
//   * f16 / bf16 go straight into the final buffer with global_atomic_pk_add_{f16,bf16} (a
//     256-byte row is 4 segments instead of the 8 of an fp32 shadow, no memset of the shadow, no
//     narrowing pass): the lane adds its value in its half of the aligned 4-byte word and +0 in
//     the other half; when both halves of a word go to the same entry the even lane adds both.
//     This is the reference's own arithmetic class (its scatter_add_ accumulates in the narrow
//     type, one rounding per add) in a non-deterministic order.  TSAMD_MINMAX_BW_SHADOW=1 (and
//     buffers that are not 4-byte aligned) use the fp32 shadow, rounded once;
//   * kBwRows rows x 2 feature tiles are in flight per wave so that the dependent chain
//     arg -> col[arg] -> atomic is overlapped; every row has exactly K elements, so the work is
//     balanced whatever the degrees are.
// grad_value targets are the row's OWN entries: contributions are summed in a per-wave LDS array
// indexed by (arg - rowptr[m]) (ds_add_f32) and every entry of the row is written once with a
// plain store -- no global atomics, no memset.  Rows longer than the LDS array take several
// passes over their (L2-resident) args.
// ---------------------------------------------------------------------------
constexpr int kBwRows = 2;     // row groups in flight per wave
constexpr int kBwTiles = 2;    // 64-feature tiles in flight per row
constexpr int kBwSlots = 512;  // LDS accumulator slots per wave (grad_value)

typedef short short2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void atomic_add_elem(float *p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_elem(double *p, double v) { atomicAdd(p, v); }
// p must be 4-byte aligned; `bits` holds two narrow values (low half = lower address)
__device__ __forceinline__ void atomic_add_pair(bf16_t *p, uint32_t bits) {
  short2v v;
  v.x = (short)(bits & 0xFFFFu);
  v.y = (short)(bits >> 16);
  __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) short2v *)p, v);
}
__device__ __forceinline__ void atomic_add_pair(f16_t *p, uint32_t bits) {
  union {
    uint32_t u;
    half2v h;
  } c;
  c.u = bits;
  __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2v *)p, c.h);
}
__device__ __forceinline__ uint32_t narrow_bits(bf16_t v) { return v.bits; }
__device__ __forceinline__ uint32_t narrow_bits(f16_t v) {
  union {
    f16_t h;
    uint16_t u;
  } c;
  c.h = v;
  return c.u;
}
// value held by the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2], no LDS traffic
__device__ __forceinline__ uint32_t neighbour(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// TM = element type of the grad_mat accumulator: T itself (fp32 / fp64 per-element atomics,
// packed atomics for f16 / bf16) or float (fp32 shadow of a narrow type).
template <typename T, typename TM, bool GMAT, bool GVAL>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void spmm_minmax_bw_kernel(
    const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
    const T *__restrict__ value, const T *__restrict__ mat, const T *__restrict__ grad_out,
    const int64_t *__restrict__ arg_out, T *__restrict__ gval, TM *__restrict__ gmat, int64_t B,
    int64_t M, int64_t N, uint32_t K, int64_t E, int lgG) {
  using A = typename Traits<T>::acc_t;
  constexpr bool kPacked = Traits<T>::kNarrow && std::is_same<T, TM>::value;
  __shared__ A slots_[GVAL ? kWavesPerBlock * kBwSlots : 1];
  const int lane = (int)(threadIdx.x & 63);
  const int wib = (int)(threadIdx.x >> 6);
  const int lpr = 64 >> lgG;  // lanes per row: 64 for K >= 64 (then G = 1 and K is tiled)
  const int g = lane >> (6 - lgG);
  const int kl = lane & (lpr - 1);
  const uint32_t ktiles = (K + (uint32_t)lpr - 1) / (uint32_t)lpr;
  const int cap = kBwSlots / kBwRows >> lgG;  // LDS slots per row in flight
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const bool merge_pairs = kPacked && (K & 1u) == 0;  // word mates = lanes (2j, 2j+1) of one row

  int64_t m[kBwRows], rs[kBwRows], deg[kBwRows];
  bool ok[kBwRows];
#pragma unroll
  for (int r = 0; r < kBwRows; ++r) {
    m[r] = ((unit * kBwRows + r) << lgG) + g;
    ok[r] = m[r] < M;
    rs[r] = 0;
    deg[r] = 0;
    if (GVAL && ok[r]) {
      rs[r] = rowptr[m[r]];
      deg[r] = rowptr[m[r] + 1] - rs[r];
    }
  }

  for (int64_t c0 = 0;; c0 += cap) {  // passes over the row's entries (one unless deg > cap)
    A *slot[kBwRows];
    if constexpr (GVAL) {
#pragma unroll
      for (int r = 0; r < kBwRows; ++r) {
        slot[r] = slots_ + wib * kBwSlots + ((r << lgG) + g) * cap;
        const int64_t n = deg[r] - c0 < cap ? deg[r] - c0 : cap;
        for (int i = kl; i < n; i += lpr) slot[r][i] = A(0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const bool scatter = GMAT && c0 == 0;
    for (int64_t b = 0; b < B; ++b) {
      for (uint32_t t0 = 0; t0 < ktiles; t0 += kBwTiles) {
        int64_t a[kBwRows][kBwTiles];
        T go[kBwRows][kBwTiles];
        uint32_t kk[kBwTiles];
#pragma unroll
        for (int u = 0; u < kBwTiles; ++u) kk[u] = (t0 + u) * (uint32_t)lpr + (uint32_t)kl;
#pragma unroll
        for (int r = 0; r < kBwRows; ++r) {
#pragma unroll
          for (int u = 0; u < kBwTiles; ++u) {
            a[r][u] = E;
            go[r][u] = Traits<T>::from_acc(A(0));
            if (ok[r] && kk[u] < K) {
              const uint64_t off = ((uint64_t)b * M + (uint64_t)m[r]) * K + kk[u];
              a[r][u] = arg_out[off];
              go[r][u] = grad_out[off];
            }
          }
        }
        uint32_t c[kBwRows][kBwTiles];
        A w[kBwRows][kBwTiles];
#pragma unroll
        for (int r = 0; r < kBwRows; ++r) {
#pragma unroll
          for (int u = 0; u < kBwTiles; ++u) {
            const bool valid = a[r][u] != E;  // empty row / no winner: masked out (spmm.cpp:210)
            c[r][u] = valid ? (uint32_t)col[a[r][u]] : 0u;
            w[r][u] = A(1);
            if (scatter && value != nullptr && valid) w[r][u] = Traits<T>::to_acc(value[a[r][u]]);
          }
        }
        if constexpr (GVAL) {
          A x[kBwRows][kBwTiles];
#pragma unroll
          for (int r = 0; r < kBwRows; ++r) {
#pragma unroll
            for (int u = 0; u < kBwTiles; ++u) {
              const bool valid = a[r][u] != E;
              x[r][u] = valid ? Traits<T>::to_acc(mat[((uint64_t)b * N + c[r][u]) * K + kk[u]]) : A(0);
            }
          }
#pragma unroll
          for (int r = 0; r < kBwRows; ++r) {
#pragma unroll
            for (int u = 0; u < kBwTiles; ++u) {
              const int64_t rel = a[r][u] - rs[r] - c0;
              if (a[r][u] != E && rel >= 0 && rel < cap)
                atomicAdd(&slot[r][rel], x[r][u] * Traits<T>::to_acc(go[r][u]));
            }
          }
        }
        if (scatter) {
#pragma unroll
          for (int r = 0; r < kBwRows; ++r) {
#pragma unroll
            for (int u = 0; u < kBwTiles; ++u) {
              const bool valid = a[r][u] != E;
              const uint64_t idx = ((uint64_t)b * N + c[r][u]) * K + kk[u];
              if constexpr (kPacked) {
                // the reference rounds the product to the narrow type before it is accumulated
                uint32_t bits = narrow_bits(Traits<T>::from_acc(w[r][u] * Traits<T>::to_acc(go[r][u])));
                bool mine = valid;
                if (merge_pairs) {  // wave-uniform
                  const uint32_t alo = (uint32_t)(uint64_t)a[r][u], ahi = (uint32_t)((uint64_t)a[r][u] >> 32);
                  const uint32_t nlo = neighbour(alo), nhi = neighbour(ahi), nbits = neighbour(bits);
                  const bool same = valid && nlo == alo && nhi == ahi;  // the mate is valid as well then
                  if (same) {
                    if (lane & 1) mine = false;          // the even lane adds both halves
                    else bits |= nbits << 16;
                  } else if (lane & 1) {
                    bits <<= 16;
                  }
                } else if (idx & 1) {
                  bits <<= 16;
                }
                if (mine) atomic_add_pair(gmat + (idx & ~(uint64_t)1), bits);
              } else {
                if (valid) atomic_add_elem(gmat + idx, (TM)(w[r][u] * Traits<T>::to_acc(go[r][u])));
              }
            }
          }
        }
      }
    }
    if constexpr (!GVAL) {
      break;
    } else {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      bool more = false;
#pragma unroll
      for (int r = 0; r < kBwRows; ++r) {
        const int64_t n = deg[r] - c0 < cap ? deg[r] - c0 : cap;
        for (int i = kl; i < n; i += lpr) gval[rs[r] + c0 + i] = Traits<T>::from_acc(slot[r][i]);
        more = more || deg[r] > c0 + cap;
      }
      if (!__any(more)) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}


// ---------------------------------------------------------------------------
// Pull formulation of the min/max backward (tsamd_spmm_minmax_bw_csc), step 1: the winner records
// (layout: WinRecord in spmm_internal.h -- mask words, row id, value -- `S` words per (batch, entry)).
//   mask bit k of entry e  =  (arg_out[b, row(e), k] == e)
// ENTRY-balanced (a row-parallel version spent 1.8 ms of a 2^20-row R-MAT matrix on its hub rows): a
// wave owns 64 consecutive entries and writes their records exactly once, fully coalesced.  Lane j
// holds row(e0 + j); for every DISTINCT row that intersects the chunk (~4 at 20 entries per row; the
// 1 KB of winners of a hub row is re-read by each of its chunks from L2) lane l loads the winners of
// features t * 64 + l and ORs its bit into the winner's mask in a per-wave LDS tile (ds_or_b32) when
// that entry belongs to the chunk.  Two feature tiles (four mask words) per pass; the owners then read
// their four words back and store them with one 16-byte store -- zeros included, nothing to memset, no
// global atomics.  A winner is only trusted to lie inside the chunk (LDS bounds); a foreign arg_out
// gives a wrong mask, never a wild store.
// ---------------------------------------------------------------------------
#ifndef TSAMD_WINREC_LINE_STORES
#define TSAMD_WINREC_LINE_STORES 1
#endif
#ifndef TSAMD_WINREC_ROWS
#define TSAMD_WINREC_ROWS 2
#endif
// ARG = int64_t (the API's arg_out) or int32_t (tsamd_spmm_minmax_arg32: the same ids in half the bytes)
template <typename T, typename ARG>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void minmax_winrec_kernel(
    const int64_t *__restrict__ row, const T *__restrict__ value, const ARG *__restrict__ arg_out,
    uint32_t *__restrict__ rec, int64_t B, int64_t M, uint32_t K, int64_t E, uint32_t W, uint32_t S) {
  using A = typename Traits<T>::acc_t;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __shared__ uint32_t tile_[kWavesPerBlock][kWave * 4];
  __shared__ uint32_t meta_[kWavesPerBlock][kWave * 4];
  const int lane = (int)(threadIdx.x & 63);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t *tile = tile_[wib];
  uint32_t *meta = meta_[wib];
  const int64_t e0 = ((int64_t)blockIdx.x * kWavesPerBlock + wib) * kWave;
  if (e0 >= E) return;
  const int n = (int)(E - e0 < kWave ? E - e0 : kWave);
  const bool whole32 = TSAMD_WINREC_LINE_STORES && W == 4u && S == 8u && n == kWave;  // wave-uniform
  const uint32_t ntiles = (K + 63u) >> 6;
  const bool mine = lane < n;
  const uint32_t m_l = mine ? (uint32_t)row[e0 + lane] : 0xFFFFFFFFu;
  // heads: lanes whose entry starts a new row inside the chunk (lane 0 always)
  const uint32_t m_prev = lane_read(m_l, lane > 0 ? lane - 1 : 0);
  const unsigned long long heads = __ballot(mine && (lane == 0 || m_l != m_prev));
  uint32_t vlo = 0, vhi = 0;
  if (mine) {
    A v = A(1);
    if (value != nullptr) v = Traits<T>::to_acc(value[e0 + lane]);
    if constexpr (sizeof(A) == 8) {
      uint64_t bits;
      __builtin_memcpy(&bits, &v, 8);
      vlo = (uint32_t)bits;
      vhi = (uint32_t)(bits >> 32);
    } else {
      __builtin_memcpy(&vlo, &v, 4);
    }
  }
  const uint32_t bit = 1u << (lane & 31), half = (uint32_t)lane >> 5;
  for (int64_t b = 0; b < B; ++b) {
    const ARG *a_b = arg_out + (uint64_t)b * M * K;
    uint32_t *rec_l = rec + ((uint64_t)b * (uint64_t)E + (uint64_t)(e0 + lane)) * S;
    uint32_t z = 0;  // bit s: mask word s of this entry is non-zero (words 0..31; the masked sum skips the others' segments)
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 2) {
      if (mine) *reinterpret_cast<u32x4 *>(tile + lane * 4) = u32x4{0u, 0u, 0u, 0u};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t k0 = t0 * 64u + (uint32_t)lane, k1 = k0 + 64u;
      unsigned long long todo = heads;
      while (todo != 0) {  // up to kRows rows per step: their loads are independent -- one round trip for all of them
        // (TSAMD_WINREC_ROWS = 2 / 4 / 8 rows per step measured within 0.5 %: profiles/r05_ab_winrec_rows.log -- the kernel does not wait for these loads)
        constexpr int kRows = TSAMD_WINREC_ROWS;
        uint64_t rbase[kRows];
        bool have[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
          have[u] = todo != 0;
          const int p = have[u] ? (int)__builtin_ctzll(todo) : 0;
          if (have[u]) todo &= todo - 1;
          rbase[u] = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m_l, p) * K;
        }
        // (the ids stay in their own width until all loads are issued: a widening inside the predicated blocks made
        // the compiler wait for every int32 load on its own -- four round trips instead of one, 0.31 -> 0.40 ms)
        ARG l0[kRows], l1[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
          l0[u] = -1;
          l1[u] = -1;
          if (have[u] && k0 < K) l0[u] = a_b[rbase[u] + k0];
          if (have[u] && k1 < K) l1[u] = a_b[rbase[u] + k1];
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u) asm volatile("" : "+v"(l0[u]), "+v"(l1[u]));
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
          const int64_t a0 = (int64_t)l0[u], a1 = (int64_t)l1[u];
          const int64_t r0 = a0 - e0, r1 = a1 - e0;
          if (a0 >= 0 && r0 >= 0 && r0 < n) atomicOr(tile + (uint32_t)r0 * 4 + half, bit);
          if (a1 >= 0 && r1 >= 0 && r1 < n) atomicOr(tile + (uint32_t)r1 * 4 + 2 + half, bit);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (whole32) {
        // 32-byte records (K <= 128: four mask words + row id, value, segment bitmap), a full chunk: the chunk's 2 KB go out
        // as two instructions of 64 x 16 CONTIGUOUS bytes (lane j of half h writes the j-th 16-byte piece of that half:
        // the mask or the meta part of entry (64 h + j) / 2, read back from LDS).  As two 16-byte stores per lane at a
        // 32-byte pitch every instruction touched all 32 segments of the block half-filled -- twice the segment touches.
        const u32x4 v = *reinterpret_cast<const u32x4 *>(tile + lane * 4);
        z = (v.x != 0u ? 1u : 0u) | (v.y != 0u ? 2u : 0u) | (v.z != 0u ? 4u : 0u) | (v.w != 0u ? 8u : 0u);
        *reinterpret_cast<u32x4 *>(meta + lane * 4) = u32x4{m_l, vlo, vhi, z};
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t *blk = rec + ((uint64_t)b * (uint64_t)E + (uint64_t)e0) * 8u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = h * 64 + lane, ent = j >> 1;
          const u32x4 piece = *reinterpret_cast<const u32x4 *>(((j & 1) ? meta : tile) + ent * 4);
          *reinterpret_cast<u32x4 *>(blk + j * 4) = piece;
        }
      } else if (mine) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(tile + lane * 4);
        if (2u * t0 < 32u)  // (words past W stay zero in the tile)
          z |= ((v.x != 0u ? 1u : 0u) | (v.y != 0u ? 2u : 0u) | (v.z != 0u ? 4u : 0u) | (v.w != 0u ? 8u : 0u)) << (2u * t0);
        uint32_t *dst = rec_l + 2u * t0;
        const uint32_t left = W - 2u * t0;  // mask words of this entry from tile t0 on (>= 1)
        if (left >= 4) {
          *reinterpret_cast<u32x4 *>(dst) = v;  // S and 2 * t0 are multiples of 4 words: aligned
        } else {
          dst[0] = v.x;
          if (left > 1) dst[1] = v.y;
          if (left > 2) dst[2] = v.z;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (mine && !whole32) {  // the entry's row id and value behind its mask
      if ((W & 3u) == 0) {
        *reinterpret_cast<u32x4 *>(rec_l + W) = u32x4{m_l, vlo, vhi, z};
      } else {
        rec_l[W] = m_l;
        rec_l[W + 1] = vlo;
        rec_l[W + 2] = vhi;
        if (((W + 3u) & 3u) != 0u) rec_l[W + 3] = z;  // the record's padding word, when it has one
      }
    }
  }
}

template <typename T>
__global__ void narrow_from_f32_kernel(const float *__restrict__ src, T *__restrict__ dst,
                                       int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = Traits<T>::from_acc(src[i]);
}

int ilog2_ceil(uint32_t x) {
  int l = 0;
  while ((1u << l) < x) ++l;
  return l;
}

template <typename T, int VEC>
int launch_value_bw(const int64_t *row, const int64_t *rowptr, const int64_t *col, const T *mat,
                    const T *grad, T *out, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                    bool mean, hipStream_t stream) {
  const uint32_t slots = (uint32_t)((K + VEC - 1) / VEC);
  const uint32_t lpr = slots >= 64 ? 64u : (1u << ilog2_ceil(slots));
  const int lgG = 6 - ilog2_ceil(lpr);
  const unsigned int blocks = (unsigned int)ceil_div(ceil_div(E, kWave), kWavesPerBlock);
  hipLaunchKernelGGL((spmm_value_bw_kernel<T, VEC>), dim3(blocks), dim3(kWavesPerBlock * kWave), 0,
                     stream, row, rowptr, col, mat, grad, out, B, M, N, (uint32_t)K, E, lgG, mean);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

// grad_value of the min/max backward from the winner records (16-byte packets required)
template <typename T>
int launch_value_bw_masked(const int64_t *row, const int64_t *rowptr, const int64_t *col, const T *mat,
                           const T *grad, T *out, const uint32_t *rec, uint32_t rec_stride, int64_t B,
                           int64_t M, int64_t N, int64_t K, int64_t E, hipStream_t stream) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const uint32_t slots = (uint32_t)(K / VEC);
  const uint32_t lpr = slots >= 64 ? 64u : (1u << ilog2_ceil(slots));
  const int lgG = 6 - ilog2_ceil(lpr);
  const unsigned int blocks = (unsigned int)ceil_div(ceil_div(E, kWave), kWavesPerBlock);
  const char *env_pipe = exp_env("TSAMD_MASKED_SDDMM_PIPE");  // experiments: 0 = the round-4 kernel
  const bool pipelined = !(env_pipe != nullptr && env_pipe[0] == '0');
  if (pipelined && slots <= 64u && K % VEC == 0 && (uint64_t)kWave * rec_stride < (1ull << 32)) {
    hipLaunchKernelGGL((spmm_value_bw_masked_kernel<T>), dim3(blocks), dim3(kWavesPerBlock * kWave), 0, stream, row,
                       rowptr, col, mat, grad, out, B, M, N, (uint32_t)K, E, lgG, rec, rec_stride);
  } else {
    hipLaunchKernelGGL((spmm_value_bw_kernel<T, VEC, true>), dim3(blocks), dim3(kWavesPerBlock * kWave), 0,
                       stream, row, rowptr, col, mat, grad, out, B, M, N, (uint32_t)K, E, lgG, false, rec,
                       rec_stride);
  }
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

template <typename T>
int dispatch_value_bw(bool vec_ok, const int64_t *row, const int64_t *rowptr, const int64_t *col,
                      const void *mat, const void *grad, void *out, int64_t B, int64_t M,
                      int64_t N, int64_t K, int64_t E, bool mean, hipStream_t stream) {
  constexpr int kVec = 16 / (int)sizeof(T);
  const T *x = reinterpret_cast<const T *>(mat);
  const T *g = reinterpret_cast<const T *>(grad);
  T *o = reinterpret_cast<T *>(out);
  if (vec_ok) return launch_value_bw<T, kVec>(row, rowptr, col, x, g, o, B, M, N, K, E, mean, stream);
  return launch_value_bw<T, 1>(row, rowptr, col, x, g, o, B, M, N, K, E, mean, stream);
}

template <typename T, typename TM>
int launch_minmax_bw(const int64_t *rowptr, const int64_t *col, const void *value, const void *mat,
                     const void *grad_out, const int64_t *arg_out, void *gval, TM *gmat, int64_t B,
                     int64_t M, int64_t N, int64_t K, int64_t E, hipStream_t stream) {
  if (B * M * K == 0) return TSAMD_OK;
  const uint32_t lpr = K >= 64 ? 64u : (1u << ilog2_ceil((uint32_t)K));
  const int lgG = 6 - ilog2_ceil(lpr);
  const int64_t rows_per_wave = (int64_t)kBwRows << lgG;
  const unsigned int blocks = (unsigned int)ceil_div(ceil_div(M, rows_per_wave), kWavesPerBlock);
  const T *v = reinterpret_cast<const T *>(value), *x = reinterpret_cast<const T *>(mat),
          *g = reinterpret_cast<const T *>(grad_out);
  T *gv = reinterpret_cast<T *>(gval);
#define TSAMD_BW_GO(GMAT, GVAL)                                                                     \
  hipLaunchKernelGGL((spmm_minmax_bw_kernel<T, TM, GMAT, GVAL>), dim3(blocks),                      \
                     dim3(kWavesPerBlock *kWave), 0, stream, rowptr, col, v, x, g, arg_out, gv, gmat, \
                     B, M, N, (uint32_t)K, E, lgG)
  if (gmat != nullptr && gv != nullptr) TSAMD_BW_GO(true, true);
  else if (gmat != nullptr) TSAMD_BW_GO(true, false);
  else if (gv != nullptr) TSAMD_BW_GO(false, true);
#undef TSAMD_BW_GO
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

template <typename T>
int narrow_out(const float *src, void *dst, int64_t n, hipStream_t stream) {
  if (n == 0) return TSAMD_OK;
  hipLaunchKernelGGL((narrow_from_f32_kernel<T>), dim3((unsigned int)ceil_div(n, 256)), dim3(256), 0,
                     stream, src, reinterpret_cast<T *>(dst), n);
  TSAMD_LAUNCH_CHECK();
  return TSAMD_OK;
}

}  // namespace
}  // namespace tsamd

using namespace tsamd;

extern "C" int tsamd_spmm_value_bw(int dtype, int reduce, const int64_t *row,
                                   const int64_t *rowptr, const int64_t *col, const void *mat,
                                   const void *grad, void *out, int64_t B, int64_t M, int64_t N,
                                   int64_t K, int64_t E, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (reduce != TSAMD_SUM && reduce != TSAMD_MEAN) return TSAMD_ERR_UNSUPPORTED;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16)
    return TSAMD_ERR_UNSUPPORTED;
  if (N >= (int64_t)1 << 32 || M >= (int64_t)1 << 32 || K >= (int64_t)1 << 31)
    return TSAMD_ERR_UNSUPPORTED;
  if (E == 0) return TSAMD_OK;
  if (!rowptr || !col || !out || (B * K > 0 && (!mat || !grad))) return TSAMD_ERR_INVALID;
  const size_t es = dtype_size(dtype);
  const bool vec_ok = K > 0 && (K * es) % 16 == 0 && ((uintptr_t)mat % 16 == 0) &&
                      ((uintptr_t)grad % 16 == 0);
  const bool mean = reduce == TSAMD_MEAN;
  return TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    if constexpr (std::is_integral<scalar_t>::value) {
      return (int)TSAMD_ERR_UNSUPPORTED;
    } else {
      return dispatch_value_bw<scalar_t>(vec_ok, row, rowptr, col, mat, grad, out, B, M, N, K, E,
                                         mean, stream);
    }
  });
}

// f16 / bf16 grad_mat: packed atomics on the final buffer unless TSAMD_MINMAX_BW_SHADOW=1 asks for
// the fp32 shadow (round once; costs a memset, fp32 atomics on twice as many 64-byte segments and a
// narrowing pass over [B,N,K]).
static bool minmax_bw_shadow(int dtype) {
  if (dtype != TSAMD_F16 && dtype != TSAMD_BF16) return false;
  const char *env = exp_env("TSAMD_MINMAX_BW_SHADOW");
  return env != nullptr && env[0] == '1';
}

extern "C" size_t tsamd_spmm_minmax_bw_workspace_bytes(int dtype, int64_t B, int64_t N, int64_t K,
                                                       int64_t E) {
  (void)E;
  const bool narrow = dtype == TSAMD_F16 || dtype == TSAMD_BF16;
  // packed atomics need whole 4-byte words inside the buffer: an odd element count takes the shadow
  if (minmax_bw_shadow(dtype) || (narrow && ((B * N * K) % 2) != 0))
    return align_up(sizeof(float) * (size_t)(B * N * K), 256);
  return 0;
}

namespace {
template <typename T>
int run_minmax_bw(const int64_t *rowptr, const int64_t *col, const void *value, const void *mat,
                  const void *grad_out, const int64_t *arg_out, void *grad_value, void *grad_mat,
                  float *shadow_mat, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                  hipStream_t stream) {
  if constexpr (Traits<T>::kNarrow) {
    if (shadow_mat != nullptr || grad_mat == nullptr)
      return launch_minmax_bw<T, float>(rowptr, col, value, mat, grad_out, arg_out, grad_value,
                                        shadow_mat, B, M, N, K, E, stream);
  }
  return launch_minmax_bw<T, T>(rowptr, col, value, mat, grad_out, arg_out, grad_value,
                                reinterpret_cast<T *>(grad_mat), B, M, N, K, E, stream);
}
}  // namespace

extern "C" int tsamd_spmm_minmax_bw(int dtype, const int64_t *rowptr, const int64_t *col,
                                    const void *value, const void *mat, const void *grad_out,
                                    const int64_t *arg_out, void *grad_value, void *grad_mat,
                                    int64_t B, int64_t M, int64_t N, int64_t K, int64_t E,
                                    void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16)
    return TSAMD_ERR_UNSUPPORTED;
  if (N >= (int64_t)1 << 32 || K >= (int64_t)1 << 31) return TSAMD_ERR_UNSUPPORTED;
  const int64_t total = B * M * K;
  if (total > 0 && (!col || !mat || !grad_out || !arg_out)) return TSAMD_ERR_INVALID;
  if (grad_value && E > 0 && !rowptr) return TSAMD_ERR_INVALID;  // grad_value is laid out by rows
  const size_t es = dtype_size(dtype);
  const size_t nmat = (size_t)(B * N * K);
  const bool narrow = dtype == TSAMD_F16 || dtype == TSAMD_BF16;
  bool shadow = narrow && grad_mat && minmax_bw_shadow(dtype);
  // packed atomics work on aligned 4-byte words inside the buffer
  if (narrow && grad_mat && !shadow && (((uintptr_t)grad_mat % 4) != 0 || (nmat % 2) != 0)) shadow = true;
  float *shadow_mat = nullptr;
  if (shadow) {
    const size_t need = align_up(sizeof(float) * nmat, 256);
    if (!workspace || workspace_bytes < need) return TSAMD_ERR_WORKSPACE;
    shadow_mat = reinterpret_cast<float *>(workspace);
    TSAMD_HIP_TRY(hipMemsetAsync(shadow_mat, 0, sizeof(float) * nmat, stream));
  } else if (grad_mat) {
    TSAMD_HIP_TRY(hipMemsetAsync(grad_mat, 0, es * nmat, stream));
  }
  if (total == 0 || (!grad_value && !grad_mat)) {
    if (grad_value && E > 0) TSAMD_HIP_TRY(hipMemsetAsync(grad_value, 0, es * (size_t)E, stream));
    return TSAMD_OK;
  }
  int st = TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    if constexpr (std::is_integral<scalar_t>::value) {
      return (int)TSAMD_ERR_UNSUPPORTED;
    } else {
      return run_minmax_bw<scalar_t>(rowptr, col, value, mat, grad_out, arg_out, grad_value, grad_mat,
                                     shadow_mat, B, M, N, K, E, stream);
    }
  });
  if (st != TSAMD_OK) return st;
  if (shadow_mat) {
    if (dtype == TSAMD_F16) return narrow_out<f16_t>(shadow_mat, grad_mat, (int64_t)nmat, stream);
    return narrow_out<bf16_t>(shadow_mat, grad_mat, (int64_t)nmat, stream);
  }
  return TSAMD_OK;
}

// ---------------------------------------------------------------------------
// min/max backward, pull formulation over the transposed pattern (see include/tsamd.h)
// ---------------------------------------------------------------------------
static size_t winrec_bytes(int64_t B, int64_t K, int64_t E) {
  return align_up(sizeof(uint32_t) * (size_t)(B * E) * win_record_stride(K), 256);
}

// grad_mat route of the pull: winner bit masks + the masked merge-path SpMM (default), or -- TSAMD_MINMAX_BW_LISTS=1,
// K <= 1024 -- compacted winner lists (csrc/spmm_bw_list.hip).  Same-box A/B on the 2^20 R-MAT graph
// (profiles/r04_minmax_bw_routes.md): the lists move a third of the bytes but issue ~26 instructions per entry and
// one LDS add per (row, feature) and end up instruction-bound: 2.08 vs 1.78 ms at configs[2] (bf16, F = 128), ahead
// only for bf16 F = 64 (1.58 vs 1.75) and fp32 F = 256 (4.30 vs 4.43).  Kept as an option, bit-identical results
// for value-less narrow types, tests/test_spmm_gpu.py runs both.
static bool use_lists(int dtype, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E) {
#if defined(TSAMD_EXPERIMENTS)
  const char *env = exp_env("TSAMD_MINMAX_BW_LISTS");
  if (env == nullptr || env[0] != '1') return false;
  return minmax_bw_lists_supported(dtype, B, M, N, K, E);
#else
  (void)dtype; (void)B; (void)M; (void)N; (void)K; (void)E;
  return false;  // (csrc/spmm_bw_list.hip is only compiled into experiment builds)
#endif
}

extern "C" size_t tsamd_spmm_minmax_bw_csc_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N,
                                                           int64_t K, int64_t E) {
  if (dtype_size(dtype) == 0 || B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return 0;
  // the masked sum runs on the transposed matrix: N rows, M columns
  const size_t masked = winrec_bytes(B, K, E) + spmm_masked_sum_workspace_bytes(dtype, B, N, M, K, E);
#if defined(TSAMD_EXPERIMENTS)
  const size_t lists = minmax_bw_lists_supported(dtype, B, M, N, K, E) ? minmax_bw_lists_workspace_bytes(dtype, B, M, N, K, E) : 0;
#else
  const size_t lists = 0;
#endif
  return masked > lists ? masked : lists;
}

// arg32: arg_any holds int32 ids (tsamd_spmm_minmax_arg32); only the record route reads them in that width
// records_in: the winner records are there already (tsamd_spmm_minmax_records wrote them in the forward): no ids, the
// whole workspace belongs to the masked sum; `value` is then only asked for presence
static int minmax_bw_csc_impl(int dtype, const int64_t *rowptr, const int64_t *col, const void *value,
                              const void *mat, const void *grad_out, const void *arg_any, bool arg32,
                              const int64_t *colptr, const int64_t *csr2csc, const int64_t *row, void *grad_value,
                              void *grad_mat, int64_t B, int64_t M, int64_t N, int64_t K, int64_t E, void *workspace,
                              size_t workspace_bytes, void *stream_, const uint32_t *records_in = nullptr) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (records_in != nullptr) {
    arg_any = records_in;  // (never dereferenced as ids: every id-reading route below returns UNSUPPORTED first)
    arg32 = true;
  }
  const int64_t *arg_out = arg32 ? nullptr : reinterpret_cast<const int64_t *>(arg_any);
  if (arg32 && arg_any == nullptr && B * M * K > 0) return TSAMD_ERR_INVALID;
  if (arg32) arg_out = reinterpret_cast<const int64_t *>(arg_any);  // (never dereferenced at this width below)
  if (B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (dtype != TSAMD_F32 && dtype != TSAMD_F64 && dtype != TSAMD_F16 && dtype != TSAMD_BF16)
    return TSAMD_ERR_UNSUPPORTED;
  if (N >= (int64_t)1 << 32 || M >= (int64_t)1 << 32 || E >= (int64_t)1 << 32 || K >= (int64_t)1 << 31)
    return TSAMD_ERR_UNSUPPORTED;
  const int64_t total = B * M * K;
  if (total > 0 && (!rowptr || !col || !mat || !grad_out || !arg_out)) return TSAMD_ERR_INVALID;
  if (grad_mat && E > 0 && (!colptr || !csr2csc || !row)) return TSAMD_ERR_INVALID;
  const size_t es = dtype_size(dtype);
#if defined(TSAMD_EXPERIMENTS)
  if (grad_mat && total > 0 && E > 0 && B * N * K > 0 && use_lists(dtype, B, M, N, K, E)) {
    if (arg32) return TSAMD_ERR_UNSUPPORTED;  // the (opt-in) list route reads int64 ids: the caller widens them
    if (!workspace || workspace_bytes < minmax_bw_lists_workspace_bytes(dtype, B, M, N, K, E) ||
        (uintptr_t)workspace % 256 != 0)
      return TSAMD_ERR_WORKSPACE;
    if (grad_value) {  // the row-parallel kernel (per-row LDS slots, plain stores)
      int st = tsamd_spmm_minmax_bw(dtype, rowptr, col, value, mat, grad_out, arg_out, grad_value, nullptr, B, M, N,
                                    K, E, nullptr, 0, stream_);
      if (st != TSAMD_OK) return st;
    }
    return minmax_bw_lists(dtype, row, col, value, grad_out, arg_out, colptr, csr2csc, grad_mat, B, M, N, K, E,
                           workspace, stream);
  }
#endif
  // grad_value as a masked SDDMM over the records needs 16-byte packets; else the row-parallel LDS kernel
  const bool sddmm_ok = grad_value && row && (K * (int64_t)es) % 16 == 0 && ((uintptr_t)mat % 16) == 0 &&
                        ((uintptr_t)grad_out % 16) == 0;
  if (grad_value && !(sddmm_ok && grad_mat)) {
    if (arg32) return TSAMD_ERR_UNSUPPORTED;  // the row-parallel kernel reads int64 ids: the caller widens them
    int st = tsamd_spmm_minmax_bw(dtype, rowptr, col, value, mat, grad_out, arg_out, grad_value, nullptr, B, M, N,
                                  K, E, nullptr, 0, stream_);
    if (st != TSAMD_OK) return st;
  }
  if (!grad_mat || B * N * K == 0) return TSAMD_OK;
  if (total == 0 || E == 0) {
    TSAMD_HIP_TRY(hipMemsetAsync(grad_mat, 0, es * (size_t)(B * N * K), stream));
    if (grad_value && sddmm_ok && E > 0) TSAMD_HIP_TRY(hipMemsetAsync(grad_value, 0, es * (size_t)E, stream));
    return TSAMD_OK;
  }
  const size_t rec_b = records_in != nullptr ? 0 : winrec_bytes(B, K, E);
  const size_t spmm_b = spmm_masked_sum_workspace_bytes(dtype, B, N, M, K, E);
  if (!workspace || workspace_bytes < rec_b + spmm_b || (uintptr_t)workspace % 256 != 0)
    return TSAMD_ERR_WORKSPACE;
  uint32_t *rec = records_in != nullptr ? const_cast<uint32_t *>(records_in) : reinterpret_cast<uint32_t *>(workspace);
  const uint32_t W = (uint32_t)ceil_div(K, 32), S = win_record_stride(K);
  const unsigned int blocks = (unsigned int)ceil_div(ceil_div(E, kWave), kWavesPerBlock);
  int st = TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    if constexpr (std::is_integral<scalar_t>::value) {
      return (int)TSAMD_ERR_UNSUPPORTED;
    } else {
      if (records_in != nullptr) {
      } else if (arg32)
        hipLaunchKernelGGL((minmax_winrec_kernel<scalar_t, int32_t>), dim3(blocks), dim3(kWavesPerBlock * kWave), 0,
                           stream, row, reinterpret_cast<const scalar_t *>(value),
                           reinterpret_cast<const int32_t *>(arg_any), rec, B, M, (uint32_t)K, E, W, S);
      else
        hipLaunchKernelGGL((minmax_winrec_kernel<scalar_t, int64_t>), dim3(blocks), dim3(kWavesPerBlock * kWave), 0,
                           stream, row, reinterpret_cast<const scalar_t *>(value), arg_out, rec, B, M, (uint32_t)K, E,
                           W, S);
      TSAMD_LAUNCH_CHECK();
      if (grad_value && sddmm_ok)
        return launch_value_bw_masked<scalar_t>(row, rowptr, col, reinterpret_cast<const scalar_t *>(mat),
                                                reinterpret_cast<const scalar_t *>(grad_out),
                                                reinterpret_cast<scalar_t *>(grad_value), rec, S, B, M, N, K, E, stream);
      return (int)TSAMD_OK;
    }
  });
  if (st != TSAMD_OK) return st;
  // grad_mat = masked (A^T) * grad_out: rows of A^T = columns of A, entries through csr2csc
  return spmm_masked_sum(dtype, colptr, value != nullptr, csr2csc, rec, grad_out, grad_mat, B, N, M, K, E,
                         reinterpret_cast<char *>(workspace) + rec_b, workspace_bytes - rec_b, stream);
}

extern "C" int tsamd_spmm_minmax_bw_csc(int dtype, const int64_t *rowptr, const int64_t *col,
                                        const void *value, const void *mat, const void *grad_out,
                                        const int64_t *arg_out, const int64_t *colptr,
                                        const int64_t *csr2csc, const int64_t *row, void *grad_value,
                                        void *grad_mat, int64_t B, int64_t M, int64_t N, int64_t K,
                                        int64_t E, void *workspace, size_t workspace_bytes, void *stream_) {
  return minmax_bw_csc_impl(dtype, rowptr, col, value, mat, grad_out, arg_out, false, colptr, csr2csc, row,
                            grad_value, grad_mat, B, M, N, K, E, workspace, workspace_bytes, stream_);
}

extern "C" int tsamd_spmm_minmax_bw_csc_arg32(int dtype, const int64_t *rowptr, const int64_t *col,
                                              const void *value, const void *mat, const void *grad_out,
                                              const int32_t *arg_out32, const int64_t *colptr,
                                              const int64_t *csr2csc, const int64_t *row, void *grad_value,
                                              void *grad_mat, int64_t B, int64_t M, int64_t N, int64_t K,
                                              int64_t E, void *workspace, size_t workspace_bytes, void *stream_) {
  if (E >= (int64_t)1 << 31) return TSAMD_ERR_UNSUPPORTED;
  return minmax_bw_csc_impl(dtype, rowptr, col, value, mat, grad_out, arg_out32, true, colptr, csr2csc, row,
                            grad_value, grad_mat, B, M, N, K, E, workspace, workspace_bytes, stream_);
}

// ---- the pull backward on records the forward left (tsamd_spmm_minmax_records, include/tsamd.h) ------------------
namespace tsamd {
int minmax_winrec_from_ids(int dtype, const int64_t *row, const void *value, const int32_t *arg32, uint32_t *records,
                           int64_t B, int64_t M, int64_t K, int64_t E, hipStream_t stream) {
  if (E == 0 || B * M * K == 0) return TSAMD_OK;
  if (!row || !arg32 || !records) return TSAMD_ERR_INVALID;
  const uint32_t W = (uint32_t)ceil_div(K, 32), S = win_record_stride(K);
  const unsigned int blocks = (unsigned int)ceil_div(ceil_div(E, kWave), kWavesPerBlock);
  return TSAMD_DISPATCH_DTYPE(dtype, [&]() -> int {
    if constexpr (std::is_integral<scalar_t>::value) {
      return (int)TSAMD_ERR_UNSUPPORTED;
    } else {
      hipLaunchKernelGGL((minmax_winrec_kernel<scalar_t, int32_t>), dim3(blocks), dim3(kWavesPerBlock * kWave), 0, stream,
                         row, reinterpret_cast<const scalar_t *>(value), arg32, records, B, M, (uint32_t)K, E, W, S);
      TSAMD_LAUNCH_CHECK();
      return (int)TSAMD_OK;
    }
  });
}
}  // namespace tsamd

extern "C" int tsamd_spmm_minmax_winrec(int dtype, const int64_t *row, const void *value, const int32_t *arg_out32,
                                        uint32_t *records, int64_t B, int64_t M, int64_t K, int64_t E, void *stream_) {
  if (B < 0 || M < 0 || K < 0 || E < 0) return TSAMD_ERR_INVALID;
  if (E >= (int64_t)1 << 31 || M >= (int64_t)1 << 32) return TSAMD_ERR_UNSUPPORTED;
  return tsamd::minmax_winrec_from_ids(dtype, row, value, arg_out32, records, B, M, K, E,
                                       reinterpret_cast<hipStream_t>(stream_));
}

extern "C" size_t tsamd_spmm_minmax_bw_csc_records_workspace_bytes(int dtype, int64_t B, int64_t M, int64_t N, int64_t K,
                                                                   int64_t E) {
  if (dtype_size(dtype) == 0 || B < 0 || M < 0 || N < 0 || K < 0 || E < 0) return 0;
  return spmm_masked_sum_workspace_bytes(dtype, B, N, M, K, E);
}

extern "C" int tsamd_spmm_minmax_bw_csc_records(int dtype, const int64_t *rowptr, const int64_t *col, int has_value,
                                                const void *mat, const void *grad_out, const uint32_t *records,
                                                const int64_t *colptr, const int64_t *csr2csc, const int64_t *row,
                                                void *grad_value, void *grad_mat, int64_t B, int64_t M, int64_t N,
                                                int64_t K, int64_t E, void *workspace, size_t workspace_bytes,
                                                void *stream_) {
  if (!grad_mat) return TSAMD_ERR_UNSUPPORTED;  // (grad_value alone reads ids: tsamd_spmm_minmax_bw)
  if (E > 0 && B * M * K > 0 && !records) return TSAMD_ERR_INVALID;
  if (E >= (int64_t)1 << 31) return TSAMD_ERR_UNSUPPORTED;
  // `value` is only tested for presence on this route (the records carry the values): any non-null pointer will do
  return minmax_bw_csc_impl(dtype, rowptr, col, has_value ? mat : nullptr, mat, grad_out, records, true, colptr, csr2csc,
                            row, grad_value, grad_mat, B, M, N, K, E, workspace, workspace_bytes, stream_,
                            records ? records : reinterpret_cast<const uint32_t *>(mat));
}
