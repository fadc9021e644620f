// Micro-benchmark: what does a device-scope floating-point atomic cost on MI355X as a function of
// how the 64 lanes of one instruction spread over cache lines?  (Design input for the min/max
// backward scatter, csrc/spmm_bw.hip.)
//   hipcc -O3 --offload-arch=gfx950 -o atomics atomics.hip && ./atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ inline uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// KIND: 0 = f32 atomic add, 1 = packed bf16 atomic add, 2 = plain dword store, 3 = f32 atomic with return
// lgL: log2(lanes per target row); the group's lanes hit 2^lgL consecutive 4-byte words
// keep: a lane is active with probability keep/256
template <int KIND>
__global__ __launch_bounds__(256) void bench(uint32_t *buf, uint32_t nrows, uint32_t row_words, int iters,
                                             int lgL, int keep, uint32_t *sink) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t L = 1u << lgL;
  const uint32_t grp = lane >> lgL, kl = lane & (L - 1);
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t h = mix((wave * (uint32_t)iters + it) * 64u + grp);
    const uint32_t row = h % nrows;
    const uint32_t start = ((h >> 20) % (row_words / L)) * L;  // aligned window of L words inside the row
    uint32_t *p = buf + (uint64_t)row * row_words + start + kl;
    const bool on = (mix(h ^ (lane * 0x9E3779B1u)) & 255u) < (uint32_t)keep;
    if (on) {
      if (KIND == 0) atomicAdd(reinterpret_cast<float *>(p), 1.0f);
      else if (KIND == 1) {
        short2v v; v.x = 0x3F80; v.y = 0x3F80;
        __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) short2v *)p, v);
      } else if (KIND == 2) *p = it;
      else acc += __float_as_uint(atomicAdd(reinterpret_cast<float *>(p), 1.0f));
    }
  }
  if (KIND == 3 && acc == 0x12345u) sink[0] = acc;
}

int main() {
  const uint32_t nrows = 1u << 20;
  const int waves = 1 << 16, iters = 32;
  uint32_t *buf, *sink;
  const uint32_t row_words_max = 128;
  CHECK(hipMalloc(&buf, (size_t)nrows * row_words_max * 4));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 0, (size_t)nrows * row_words_max * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char *names[4] = {"f32_atomic", "pk_bf16_atomic", "plain_store", "f32_atomic_ret"};
  printf("kind,row_bytes,lanes_per_row,window_bytes,keep,ms,G_lane_ops_per_s,G_instr_groups_per_s,ps_per_group\n");
  for (int kind = 0; kind < 4; ++kind) {
    for (uint32_t row_words : {128u, 64u}) {
      for (int lgL = 6; lgL >= 0; lgL -= 1) {
        for (int keep : {256, 32}) {
          if (keep == 32 && lgL < 4) continue;
          if (kind == 3 && !(lgL == 6 || lgL == 0)) continue;
          if ((1u << lgL) > row_words) continue;
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            switch (kind) {
              case 0: hipLaunchKernelGGL(bench<0>, dim3(waves / 4), dim3(256), 0, 0, buf, nrows, row_words, iters, lgL, keep, sink); break;
              case 1: hipLaunchKernelGGL(bench<1>, dim3(waves / 4), dim3(256), 0, 0, buf, nrows, row_words, iters, lgL, keep, sink); break;
              case 2: hipLaunchKernelGGL(bench<2>, dim3(waves / 4), dim3(256), 0, 0, buf, nrows, row_words, iters, lgL, keep, sink); break;
              default: hipLaunchKernelGGL(bench<3>, dim3(waves / 4), dim3(256), 0, 0, buf, nrows, row_words, iters, lgL, keep, sink); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const double lane_ops = (double)waves * iters * 64 * keep / 256.0;
          const double groups = (double)waves * iters * (64 >> lgL);
          printf("%s,%u,%d,%d,%d,%.3f,%.2f,%.2f,%.1f\n", names[kind], row_words * 4, 1 << lgL, 4 << lgL, keep, best,
                 lane_ops / best / 1e6, groups / best / 1e6, best * 1e9 / groups);
        }
      }
    }
  }
  return 0;
}
