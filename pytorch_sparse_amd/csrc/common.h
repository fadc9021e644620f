// Shared device/host helpers for the gfx950 kernels (wave64 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <limits>
#include <type_traits>

#include "tsamd.h"

namespace tsamd {

// --------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------
extern thread_local int g_last_hip_error;

#define TSAMD_HIP_TRY(expr)                       \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) {                       \
      ::tsamd::g_last_hip_error = (int)_e;        \
      return TSAMD_ERR_HIP;                       \
    }                                             \
  } while (0)

#define TSAMD_LAUNCH_CHECK() TSAMD_HIP_TRY(hipGetLastError())

// Experiment switches (A/B runs of rejected or alternative variants: scripts/variants.py builds with
// -DTSAMD_EXPERIMENTS=1) read the environment; the shipped library has none of them: exp_env() is a constant there
// and the branches behind it fold away.
#if defined(TSAMD_EXPERIMENTS)
static inline const char *exp_env(const char *name) { return getenv(name); }
#else
static inline const char *exp_env(const char *) { return nullptr; }
#endif

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;

// --------------------------------------------------------------------------
// element types.  bf16 is carried as raw bits; f16 uses the native _Float16.
// --------------------------------------------------------------------------
struct bf16_t {
  uint16_t bits;
};
using f16_t = _Float16;

__host__ __device__ inline float bf16_to_f32(bf16_t h) {
  union {
    uint32_t u;
    float f;
  } c;
  c.u = ((uint32_t)h.bits) << 16;
  return c.f;
}

// round-to-nearest-even, NaN -> quiet NaN (same rule as c10::BFloat16).  On the device the rounding
// is gfx950's v_cvt_pk_bf16_f32 (one instruction instead of five); NaNs are canonicalised to 0x7FC0
// by hand because c10 does (the hardware keeps sign / payload).
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
  bf16_t r;
#if defined(__HIP_DEVICE_COMPILE__)
  const __bf16 h = (__bf16)f;
  uint16_t b;
  __builtin_memcpy(&b, &h, 2);
  r.bits = f != f ? (uint16_t)0x7FC0 : b;
#else
  union {
    uint32_t u;
    float f;
  } c;
  c.f = f;
  if (f != f) {
    r.bits = 0x7FC0;
  } else {
    uint32_t bias = 0x7FFFu + ((c.u >> 16) & 1u);
    r.bits = (uint16_t)((c.u + bias) >> 16);
  }
#endif
  return r;
}

// fp32 -> bf16 -> fp32 for values that are only compared afterwards (NaN payloads do not matter)
__device__ inline float round_through_bf16(float f) {
  const __bf16 h = (__bf16)f;
  uint16_t b;
  __builtin_memcpy(&b, &h, 2);
  union {
    uint32_t u;
    float f;
  } c;
  c.u = ((uint32_t)b) << 16;
  return c.f;
}

// Traits<T>: accumulator type, conversions, and the reducer's init values
// (csrc/cpu/reducer.h:43-52 of the reference: max() for MIN, lowest() for MAX).
template <typename T>
struct Traits;

template <>
struct Traits<float> {
  using acc_t = float;
  static constexpr bool kNarrow = false;
  __device__ static inline acc_t to_acc(float x) { return x; }
  __device__ static inline float from_acc(acc_t a) { return a; }
  __device__ static inline acc_t round_acc(acc_t a) { return a; }
  __device__ static inline acc_t max_init() { return 3.402823466e+38f; }
  __device__ static inline acc_t lowest_init() { return -3.402823466e+38f; }
};

template <>
struct Traits<double> {
  using acc_t = double;
  static constexpr bool kNarrow = false;
  __device__ static inline acc_t to_acc(double x) { return x; }
  __device__ static inline double from_acc(acc_t a) { return a; }
  __device__ static inline acc_t round_acc(acc_t a) { return a; }
  __device__ static inline acc_t max_init() { return 1.7976931348623157e+308; }
  __device__ static inline acc_t lowest_init() { return -1.7976931348623157e+308; }
};

template <>
struct Traits<f16_t> {
  using acc_t = float;
  static constexpr bool kNarrow = true;
  __device__ static inline acc_t to_acc(f16_t x) { return (float)x; }
  __device__ static inline f16_t from_acc(acc_t a) { return (f16_t)a; }
  // The asm barrier keeps LLVM from folding `fptrunc(fmul a, b)` into
  // v_fma_mixlo_f16(a, b, +0), which turns a -0.0 product into +0.0.
  __device__ static inline acc_t round_acc(acc_t a) {
    asm("" : "+v"(a));
    return (float)(f16_t)a;
  }
  __device__ static inline acc_t max_init() { return 65504.0f; }
  __device__ static inline acc_t lowest_init() { return -65504.0f; }
};

template <>
struct Traits<bf16_t> {
  using acc_t = float;
  static constexpr bool kNarrow = true;
  __device__ static inline acc_t to_acc(bf16_t x) { return bf16_to_f32(x); }
  __device__ static inline bf16_t from_acc(acc_t a) { return f32_to_bf16(a); }
  __device__ static inline acc_t round_acc(acc_t a) { return round_through_bf16(a); }
  // 0x7F7F = largest finite bf16
  __device__ static inline acc_t max_init() { return 3.38953139e+38f; }
  __device__ static inline acc_t lowest_init() { return -3.38953139e+38f; }
};

template <>
struct Traits<int32_t> {
  using acc_t = int32_t;
  static constexpr bool kNarrow = false;
  __device__ static inline acc_t to_acc(int32_t x) { return x; }
  __device__ static inline int32_t from_acc(acc_t a) { return a; }
  __device__ static inline acc_t round_acc(acc_t a) { return a; }
  __device__ static inline acc_t max_init() { return 2147483647; }
  __device__ static inline acc_t lowest_init() { return (-2147483647 - 1); }
};

template <>
struct Traits<int64_t> {
  using acc_t = int64_t;
  static constexpr bool kNarrow = false;
  __device__ static inline acc_t to_acc(int64_t x) { return x; }
  __device__ static inline int64_t from_acc(acc_t a) { return a; }
  __device__ static inline acc_t round_acc(acc_t a) { return a; }
  __device__ static inline acc_t max_init() { return 9223372036854775807LL; }
  __device__ static inline acc_t lowest_init() { return (-9223372036854775807LL - 1); }
};

// 8- / 16-bit integers (SpMM forward only): carried in int32, wrapped to the element type where the
// reference's scalar_t arithmetic wraps (every product handed to the reducer, the final sum).
template <typename T>
struct SmallIntTraits {
  using acc_t = int32_t;
  static constexpr bool kNarrow = false;
  __device__ static inline acc_t to_acc(T x) { return (acc_t)x; }
  __device__ static inline T from_acc(acc_t a) { return (T)a; }
  __device__ static inline acc_t round_acc(acc_t a) { return (acc_t)(T)a; }
  __device__ static inline acc_t max_init() { return (acc_t)std::numeric_limits<T>::max(); }
  __device__ static inline acc_t lowest_init() { return (acc_t)std::numeric_limits<T>::lowest(); }
};
template <>
struct Traits<uint8_t> : SmallIntTraits<uint8_t> {};
template <>
struct Traits<int8_t> : SmallIntTraits<int8_t> {};
template <>
struct Traits<int16_t> : SmallIntTraits<int16_t> {};

// mean = sum / (scalar_t)count (reducer.h:74).  Floating types and 32 / 64-bit integers divide the
// accumulator; the small integers divide the WRAPPED sum by the WRAPPED count like the reference
// (a divisor that wraps to zero -- 256 entries in a uint8 row -- traps in the reference, gives 0 here).
template <typename T>
__device__ inline typename Traits<T>::acc_t mean_of(typename Traits<T>::acc_t sum, int64_t deg) {
  using A = typename Traits<T>::acc_t;
  const int64_t cnt = deg > 0 ? deg : 1;
  if constexpr (std::is_same<T, uint8_t>::value || std::is_same<T, int8_t>::value ||
                std::is_same<T, int16_t>::value) {
    const int32_t s = (int32_t)(T)sum, d = (int32_t)(T)cnt;
    return d == 0 ? 0 : s / d;
  } else {
    return sum / (A)cnt;
  }
}

static inline size_t dtype_size(int dtype) {
  switch (dtype) {
    case TSAMD_F32: return 4;
    case TSAMD_F64: return 8;
    case TSAMD_F16: return 2;
    case TSAMD_BF16: return 2;
    case TSAMD_I32: return 4;
    case TSAMD_I64: return 8;
    case TSAMD_U8: case TSAMD_I8: return 1;
    case TSAMD_I16: return 2;
    default: return 0;
  }
}
static inline size_t acc_size(int dtype) {
  switch (dtype) {
    case TSAMD_F32: case TSAMD_F16: case TSAMD_BF16: case TSAMD_I32: return 4;
    case TSAMD_U8: case TSAMD_I8: case TSAMD_I16: return 4;
    case TSAMD_F64: case TSAMD_I64: return 8;
    default: return 0;
  }
}

// Dispatch a tsamd_dtype to a C++ element type.
#define TSAMD_DISPATCH_DTYPE(dtype, ...)                                   \
  [&]() -> int {                                                           \
    switch (dtype) {                                                       \
      case TSAMD_F32: { using scalar_t = float; return __VA_ARGS__(); }    \
      case TSAMD_F64: { using scalar_t = double; return __VA_ARGS__(); }   \
      case TSAMD_F16: { using scalar_t = ::tsamd::f16_t; return __VA_ARGS__(); }  \
      case TSAMD_BF16: { using scalar_t = ::tsamd::bf16_t; return __VA_ARGS__(); } \
      case TSAMD_I32: { using scalar_t = int32_t; return __VA_ARGS__(); }  \
      case TSAMD_I64: { using scalar_t = int64_t; return __VA_ARGS__(); }  \
      default: return (int)TSAMD_ERR_UNSUPPORTED;                          \
    }                                                                      \
  }()

// The same plus the 8- / 16-bit integers (SpMM forward).
#define TSAMD_DISPATCH_DTYPE_ALL(dtype, ...)                                  \
  [&]() -> int {                                                              \
    switch (dtype) {                                                          \
      case TSAMD_U8: { using scalar_t = uint8_t; return __VA_ARGS__(); }      \
      case TSAMD_I8: { using scalar_t = int8_t; return __VA_ARGS__(); }       \
      case TSAMD_I16: { using scalar_t = int16_t; return __VA_ARGS__(); }     \
      default: return TSAMD_DISPATCH_DTYPE(dtype, __VA_ARGS__);               \
    }                                                                         \
  }()

// --------------------------------------------------------------------------
// wave64 cross-lane helpers.  ds_bpermute takes a byte address (lane * 4).
// --------------------------------------------------------------------------
__device__ inline uint32_t lane_read_u32(uint32_t v, int src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ inline float lane_read(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ inline int32_t lane_read(int32_t v, int src_lane) {
  return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
__device__ inline uint32_t lane_read(uint32_t v, int src_lane) {
  return lane_read_u32(v, src_lane);
}
__device__ inline int64_t lane_read(int64_t v, int src_lane) {
  uint32_t lo = lane_read_u32((uint32_t)(uint64_t)v, src_lane);
  uint32_t hi = lane_read_u32((uint32_t)((uint64_t)v >> 32), src_lane);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ inline double lane_read(double v, int src_lane) {
  return __longlong_as_double(lane_read((int64_t)__double_as_longlong(v), src_lane));
}

template <typename A>
__device__ inline A lane_xor(A v, int mask) {
  int lane = (int)(threadIdx.x & 63);
  return lane_read(v, lane ^ mask);
}

// Value held by lane + OFF (OFF = 1, 2, 4, 8, 16, 32), for reductions towards the low lanes: pure
// VALU data movement (DPP row_shl inside a row of 16 lanes, gfx950's v_permlane16_swap /
// v_permlane32_swap across rows), no LDS-pipe round trip like ds_bpermute.  Lanes whose source
// would lie outside the wave (or, for OFF < 16, outside their row of 16) get an unspecified value.
template <int OFF>
__device__ __forceinline__ uint32_t lane_down_u32(uint32_t v) {
  if constexpr (OFF == 32) {
    return (uint32_t)__builtin_amdgcn_permlane32_swap(v, v, false, false)[1];
  } else if constexpr (OFF == 16) {
    return (uint32_t)__builtin_amdgcn_permlane16_swap(v, v, false, false)[1];
  } else {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8, "lane_down: power-of-two offsets only");
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x100 | OFF, 0xF, 0xF, false);  // row_shl:OFF
  }
}
template <int OFF>
__device__ __forceinline__ float lane_down(float v) {
  return __uint_as_float(lane_down_u32<OFF>(__float_as_uint(v)));
}
template <int OFF>
__device__ __forceinline__ int32_t lane_down(int32_t v) {
  return (int32_t)lane_down_u32<OFF>((uint32_t)v);
}
template <int OFF>
__device__ __forceinline__ uint32_t lane_down(uint32_t v) {
  return lane_down_u32<OFF>(v);
}
template <int OFF>
__device__ __forceinline__ int64_t lane_down(int64_t v) {
  const uint32_t lo = lane_down_u32<OFF>((uint32_t)(uint64_t)v);
  const uint32_t hi = lane_down_u32<OFF>((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <int OFF>
__device__ __forceinline__ double lane_down(double v) {
  return __longlong_as_double(lane_down<OFF>((int64_t)__double_as_longlong(v)));
}

// Value held by lane - 1 (wave_shr:1 on the DPP network: VALU data movement, no LDS-pipe round trip like ds_bpermute).
// Lane 0 -- and a lane whose lower neighbour is disabled -- gets 0.
__device__ __forceinline__ uint32_t lane_below_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int64_t lane_below(int64_t v) {
  const uint32_t lo = lane_below_u32((uint32_t)(uint64_t)v), hi = lane_below_u32((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// A 16-byte (or narrower) packet of VEC elements; alignment lets the compiler
// emit one global_load_dwordx4 / global_store_dwordx4 per lane.
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
  T v[VEC];
};

}  // namespace tsamd
