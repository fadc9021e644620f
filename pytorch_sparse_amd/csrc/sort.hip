// Stable one-sweep LSD radix sort of COO entries on gfx950.
//
// Replaces the generic device sort the reference calls through `index_sort` / `torch.sort` when a
// SparseStorage is built from unsorted COO, when csr2csc is computed and inside coalesce / transpose
// (torch_sparse/utils.py:14-21, called from torch_sparse/storage.py:149-162, 407-429), TOGETHER with the key
// build in front of it and the gathers / divisions behind it.  Unlike the reference's default `torch.sort` the
// order of equal keys is stable.
//
// Design (round 4; the round 1-3 version took three launches per digit -- histogram, device scan, scatter --
// over 16-byte (key, payload) pairs plus a key-build and a decode pass: 1.39 GB of fabric traffic for 300 MB of
// in + out bytes at 7.5 M entries, ~35 dispatches per coalesce):
//   * key = (row << col_bits) | col  -- the same order as row * N + col without a multiply in front and a 64-bit
//     division behind; only row_bits + col_bits bits are sorted, 8 per pass;
//   * the entry's position rides in the low bits of the SAME 64-bit word when row_bits + col_bits + idx_bits <= 64
//     (7.5 M entries of a 500 k x 500 k matrix: 38 + 23): a pass moves 8 + 8 bytes per entry instead of 16 + 16;
//     otherwise a 32-bit payload array travels beside the keys (12 + 12 bytes);
//   * ONE read of (row, col) builds the words and the digit histograms of EVERY pass (and, for the
//     device-decided sort, the order probe);
//   * a pass is ONE kernel: the tile's digit counts are published and the tiles before it are looked back at
//     (decoupled look-back, one status word per (tile, digit)) -- no histogram kernel, no device scan.  Tile =
//     workgroup id: a tile only waits for LOWER tiles, and the dispatcher of every XCD hands its share of the
//     workgroups out in increasing order, so the lowest tile that has not started yet always finds its predecessors
//     running or done and a free slot on its XCD (an atomic ticket per tile cost 10-16 us per pass at 7.5 M
//     entries: 1831 serialised atomics on one word; -DTSAMD_SORT_TICKET=1 brings it back).  A look-back that does not
//     see its predecessor within ~1 s raises an error flag instead of hanging;
//   * the last pass writes the sorted row / col / permutation themselves (shifts and masks).
// 2 + passes launches (memset, build, passes), also for the device-decided variants; a value array can ride along in
// the last pass (dst[o] = src[perm[o]], 4- or 8-byte elements) instead of being gathered through the permutation later.
#include "common.h"
#include "sort.h"

#include <type_traits>

#include <atomic>
#include <mutex>

namespace tsamd {
namespace {

constexpr int kSortThreads = 256;
// Entries per thread.  Same-box A/B at 7.5 M entries (packed words): 8 / 12 / 16 / 20 / 24 / 28 / 32 / 40 items ->
// 0.41 / 0.34 / 0.31 / 0.29 / 0.29 / 0.28 / 0.27 / 0.32 ms per sort: fewer, bigger tiles shorten the look-back chain
// (it costs ~25 us per pass at 1831 tiles) until two workgroups per CU no longer fit (64 KB of LDS at 32 items);
// key + payload pairs (12 bytes per entry in LDS) stop at 24 items for the same reason (75 M entries: 3.03 vs 3.18 ms).
#ifndef TSAMD_SORT_ITEMS
#define TSAMD_SORT_ITEMS 32
#endif
#ifndef TSAMD_SORT_ITEMS_PAIRS
#define TSAMD_SORT_ITEMS_PAIRS 24
#endif
// (a 32-bit payload beside the words -- the position of pairs mode, or a 4-byte VALUE riding along with packed words,
// round 5 -- makes an entry 12 bytes in LDS)
template <bool PACKED, bool VAL = false>
constexpr int kItemsOf = (PACKED && !VAL) ? TSAMD_SORT_ITEMS : TSAMD_SORT_ITEMS_PAIRS;
constexpr int kMinTile = kSortThreads * (TSAMD_SORT_ITEMS < TSAMD_SORT_ITEMS_PAIRS ? TSAMD_SORT_ITEMS : TSAMD_SORT_ITEMS_PAIRS);
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kMaxPasses = 8;
#ifndef TSAMD_SORT_TICKET
#define TSAMD_SORT_TICKET 0
#endif
#ifndef TSAMD_SORT_LOOK
#define TSAMD_SORT_LOOK 8
#endif

// workspace header (zeroed by the one memset of a sort): 64 words
constexpr int kHdrDescents = 0, kHdrDups = 1, kHdrError = 2, kHdrMaxRow = 3, kHdrMaxCol = 4;
constexpr int kHdrFast = 5;  // 1 = the bucket path sorts this input (decided by the build kernel's last workgroup)
[[maybe_unused]] constexpr int kHdrTicket = 8;  // ticket[kMaxPasses], -DTSAMD_SORT_TICKET=1 only
constexpr int kHdrWords = 64;

// status word of (tile, digit): [63:62] 1 = the tile's own count, 2 = inclusive prefix over all tiles up to it;
// [61:56] epoch = pass + 1 (the array is zeroed once per sort, not once per pass); [55:0] count
constexpr unsigned long long kFlagLocal = 1ull << 62, kFlagPrefix = 2ull << 62;
constexpr unsigned long long kCountMask = (1ull << 56) - 1ull;
__device__ __forceinline__ unsigned long long st_load(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_store(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr unsigned int kSpinLimit = 1u << 20;  // a look-back that never sees its predecessor gives up (error flag) instead of hanging

struct KeyLayout {
  int col_bits, key_bits, idx_bits;
  bool packed;
  int passes;
};

int bits_for(int64_t n) {  // bits needed for ids in [0, n)
  int b = 0;
  while (b < 63 && ((int64_t)1 << b) < n) ++b;
  return b;
}

// ---------------------------------------------------------------------------
// Bucket path (round 6): ONE most-significant-digit scatter + ONE sort of every bucket inside LDS, instead of
// ceil(bits / 8) global passes.  The packed word (key << idx_bits | position) is unique per entry, so sorting the
// words IS the stable sort by key -- the result does not depend on the order in which the scatter fills a bucket.
//   bucket = word >> shift (the top `bits` bits of the word space that the sizes M, N, E can populate); the build
//   kernel histograms the buckets beside the pass digits, its last workgroup scans the histogram into the bucket
//   offsets and DECIDES: every bucket <= the LDS capacity -> hdr[kHdrFast] = 1 and the scatter + bucket-sort kernels
//   run while the one-sweep passes return at once; otherwise (power-law rows, heavy duplication over few keys, ...)
//   the two bucket kernels return at once and the one-sweep passes sort as before.  No host sync either way.
// ---------------------------------------------------------------------------
constexpr int kBkMaxBits = 11;                   // buckets one scatter level separates (LDS counters of a tile)
constexpr int kBkMaxBuckets = 1 << kBkMaxBits;
#ifndef TSAMD_BK_MAX_TOTAL_BITS
#define TSAMD_BK_MAX_TOTAL_BITS 14
#endif
constexpr int kBkMaxTotalBits = TSAMD_BK_MAX_TOTAL_BITS;  // buckets of a sort (two levels): bins of the build kernel's LDS histogram
constexpr int kBkMaxHist = 1 << kBkMaxTotalBits;
#ifndef TSAMD_BK_HIST_COPIES
#define TSAMD_BK_HIST_COPIES 8
#endif
constexpr int kBkHistCopies = TSAMD_BK_HIST_COPIES;
#ifndef TSAMD_BK_MIN_ENTRIES
#define TSAMD_BK_MIN_ENTRIES (1 << 17)  // below: the one-sweep passes (a handful of tiles; the bucket kernels' fixed costs win nothing)
#endif
// Two word formats in the bucket array:
//   full  (strip = 0): the packed word (key << idx_bits | position) as built; bucket = word >> shift -- the top bits of
//                      the word space the sizes can populate, which for tiny key spaces reaches into the position bits;
//                      one level only;
//   strip (strip = 1): the bucket id is the top `bits` bits of the KEY; scatter level 1 drops its `bits1` bits from the
//                      word and puts the position in ((key & low) << idx_bits | position): fits 64 bits where the full
//                      word does not (75 M entries of a 4 M x 4 M matrix: 44 + 27 bits), and lets a second scatter
//                      level split every level-1 bucket by the next `bits - bits1` bits (more buckets than a tile has
//                      LDS counters, still >= 64-byte runs per tile and bucket).
struct BucketPlan {
  int on;      // 0: the build kernel takes no bucket histogram and the plan kernel never raises kHdrFast
  int levels;  // scatter levels: 1 or 2
  int strip;
  int bits;    // log2(#buckets) = bits1 + bits2
  int bits1;
  int kshift;  // strip: bucket = key >> kshift
  int shift;   // bits of the word below the bucket id (what the bucket sort orders); full: bucket = word >> shift
  int nb;      // 1 << bits
  int cap;     // largest bucket the bucket-sort kernel of this launch holds in LDS
};
// the bits the bucket sort runs its 8-bit LSD passes over: up to two ranges of the word
struct SortBits {
  int n;      // passes that cover every bit below the bucket id, low digit first
  int n_top;  // the last n_top of them cover the bits from `lo` up: sorted first, the rest settled by the finish step
  int lo;
  unsigned char shift[12], width[12];
};

// ---------------------------------------------------------------------------
// build: words (or keys) + every pass's digit histogram (+ the order probe) in one read of (row, col)
// ---------------------------------------------------------------------------
constexpr int kBuildThreads = 1024;
constexpr int kBuildMaxWgs = 512;
__global__ __launch_bounds__(kBuildThreads) void sort_build_kernel(
    const int64_t *__restrict__ row, const int64_t *__restrict__ col, int64_t n, KeyLayout L,
    unsigned long long *__restrict__ words, unsigned long long *__restrict__ hist /* [passes][256] */,
    unsigned long long *__restrict__ hdr, const int64_t *__restrict__ todo, int probe, BucketPlan B,
    unsigned int *__restrict__ bhist, unsigned long long *__restrict__ wgstat /* [gridDim.x][4], with a bucket plan */) {
  if (todo != nullptr && *todo == 0) return;
  __shared__ unsigned int cnt[kMaxPasses][kRadix];
  __shared__ unsigned int bcnt[kBkMaxHist];
  // (the pass digits are histogrammed whether or not the passes will run: this kernel is bound by its 24 bytes per
  // entry, the counters cost nothing measurable, and a separate histogram kernel in front of the passes costs the
  // one-sweep chain 5 us when it does not run and a read of the words when it does)
#if defined(TSAMD_EXP_BUILD_NO_PASS_HIST)  // timing experiment (scripts/variants.py): the one-sweep passes would be wrong
  const int hist_passes = B.on ? 0 : L.passes;
#else
  const int hist_passes = L.passes;
#endif
  for (int p = 0; p < hist_passes; ++p)
    if (threadIdx.x < kRadix) cnt[p][threadIdx.x] = 0;
  if (B.on)
    for (int b = (int)threadIdx.x; b < B.nb; b += kBuildThreads) bcnt[b] = 0;
  __syncthreads();
  unsigned int desc = 0, dup = 0;
  unsigned long long mr = 0, mc = 0;  // probe == 2: the range check's maxima (unsigned: a negative id reads as huge)
  const int lane = (int)(threadIdx.x & 63);
  constexpr int kB = 4;  // entries per thread and step: the loads of a step are all in flight together
  for (int64_t base = (int64_t)blockIdx.x * (kBuildThreads * kB); base < n; base += (int64_t)gridDim.x * (kBuildThreads * kB)) {
    int64_t r[kB], c[kB];
    int64_t pr0[kB], pc0[kB];  // lane 0: the entry before its own (the last lane of the wave before), requested WITH the
    bool ok[kB];               // step's other loads -- as a load inside the probe below it was a round trip per step
#pragma unroll
    for (int u = 0; u < kB; ++u) {
      const int64_t i = base + u * kBuildThreads + threadIdx.x;
      ok[u] = i < n;
      r[u] = ok[u] ? row[i] : 0;
      c[u] = ok[u] ? col[i] : 0;
      pr0[u] = 0;
      pc0[u] = 0;
      if (probe && lane == 0 && ok[u] && i > 0) {
        pr0[u] = row[i - 1];
        pc0[u] = col[i - 1];
      }
    }
#pragma unroll
    for (int u = 0; u < kB; ++u) {
      const int64_t i = base + u * kBuildThreads + threadIdx.x;
      unsigned long long key = ((unsigned long long)r[u] << L.col_bits) | (unsigned long long)c[u];
      // the entry before: the neighbour lane's registers (exchanged with every lane active: a disabled source lane
      // reads as 0), except for lane 0 (one cached load per wave)
      int64_t pr = 0, pc = 0;
      if (probe) {  // wave_shr:1 on the DPP network (VALU) -- as ds_bpermute these were four LDS-pipe round trips per entry
        pr = lane_below(r[u]);
        pc = lane_below(c[u]);
      }
      if (ok[u]) {
        words[i] = L.packed ? ((key << L.idx_bits) | (unsigned long long)i) : key;
        if (probe == 2) {
          mr = (unsigned long long)r[u] > mr ? (unsigned long long)r[u] : mr;
          mc = (unsigned long long)c[u] > mc ? (unsigned long long)c[u] : mc;
        }
        if (probe && i > 0) {
          if (lane == 0) {
            pr = pr0[u];
            pc = pc0[u];
          }
          desc += (r[u] < pr) || (r[u] == pr && c[u] < pc);
          dup += (r[u] == pr) && (c[u] == pc);
        }
      }
      for (int p = 0; p < hist_passes; ++p) {
        const unsigned int d = (unsigned int)(key >> (p * kRadixBits)) & (kRadix - 1);
        // the high digits of a power-law matrix take a handful of values: one add per wave when all lanes agree
        const unsigned int d0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)d);
        const unsigned long long act = __ballot(ok[u]);
        if (__ballot(ok[u] && d == d0) == act) {
          if (lane == 0 && act) atomicAdd(&cnt[p][d0], (unsigned int)__popcll(act));
        } else if (ok[u]) {
          atomicAdd(&cnt[p][d], 1u);
        }
      }
      if (B.on) {
        const unsigned int b = B.strip ? (unsigned int)(key >> B.kshift)
                                       : (unsigned int)(((key << L.idx_bits) | (unsigned long long)i) >> B.shift);
        const unsigned int b0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b);
        const unsigned long long act = __ballot(ok[u]);
        if (__ballot(ok[u] && b == b0) == act) {
          if (lane == 0 && act) atomicAdd(&bcnt[b0], (unsigned int)__popcll(act));
        } else if (ok[u]) {
          atomicAdd(&bcnt[b], 1u);
        }
      }
    }
  }
  __syncthreads();
  // (few, big workgroups: every histogram word is a hot address -- ~12 ns per atomic, serialised; 4096 workgroups
  // spent 50 us queueing on them, 256 spend 3)
  for (int p = (int)(threadIdx.x >> 8); p < hist_passes; p += kBuildThreads / kRadix) {
    const unsigned int c = cnt[p][threadIdx.x & (kRadix - 1)];
    if (c) atomicAdd(&hist[p * kRadix + (threadIdx.x & (kRadix - 1))], (unsigned long long)c);
  }
  if (probe) {
    for (int off = 32; off > 0; off >>= 1) {
      desc += lane_xor(desc, off);
      dup += lane_xor(dup, off);
    }
    __shared__ unsigned int s_cnt[2][kBuildThreads / 64];
    if (lane == 0) {
      s_cnt[0][threadIdx.x >> 6] = desc;
      s_cnt[1][threadIdx.x >> 6] = dup;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      unsigned int t = 0;
      for (int ww = 0; ww < kBuildThreads / 64; ++ww) t += s_cnt[threadIdx.x][ww];
      // with a bucket plan: one slot per workgroup, added up by the plan kernel (the four result words are hot
      // addresses: 512 workgroups finishing together queued ~15 us on them); else one pair of atomics per workgroup
      if (B.on) wgstat[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
      else if (t) atomicAdd(&hdr[threadIdx.x == 0 ? kHdrDescents : kHdrDups], (unsigned long long)t);
    }
    if (probe == 2) {
      for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long orr = (unsigned long long)lane_xor((int64_t)mr, off), oc = (unsigned long long)lane_xor((int64_t)mc, off);
        mr = orr > mr ? orr : mr;
        mc = oc > mc ? oc : mc;
      }
      __shared__ unsigned long long s_max[2][kBuildThreads / 64];
      if (lane == 0) {
        s_max[0][threadIdx.x >> 6] = mr;
        s_max[1][threadIdx.x >> 6] = mc;
      }
      __syncthreads();
      if (threadIdx.x < 2) {
        unsigned long long t = 0;
        for (int ww = 0; ww < kBuildThreads / 64; ++ww) t = s_max[threadIdx.x][ww] > t ? s_max[threadIdx.x][ww] : t;
        if (B.on) wgstat[(size_t)blockIdx.x * 4 + 2 + threadIdx.x] = t;
        else if (t > hdr[kHdrMaxRow + threadIdx.x]) atomicMax(&hdr[kHdrMaxRow + threadIdx.x], t);
      }
    }
  }
  if (!B.on) return;
  // bucket histogram -> global (bucket_plan_kernel turns it into offsets and decides the route)
  // (kBkHistCopies copies of the histogram, one per residue of the workgroup id: 512 workgroups adding to the same
  // 128 cache lines queued for ~150 us on the lines' atomic units; 16 adds per line and copy do not)
  {
    unsigned int *mine = bhist + (size_t)(blockIdx.x % kBkHistCopies) * B.nb;
    for (int b = (int)threadIdx.x; b < B.nb; b += kBuildThreads) {
      const unsigned int c = bcnt[b];
      if (c) atomicAdd(&mine[b], c);
    }
  }
}

// The bucket histogram -> bucket offsets, the scatter cursors and the decision (one workgroup; a kernel of its own
// because the alternative -- the build kernel's last workgroup -- needs a device-scope release fence in EVERY build
// workgroup, which on this part writes the workgroup's XCD L2 back: the build kernel went from 46 to 200 us).
__global__ __launch_bounds__(kBuildThreads) void bucket_plan_kernel(
    const unsigned int *__restrict__ bhist, unsigned int *__restrict__ boff, unsigned int *__restrict__ cursor,
    unsigned int *__restrict__ cursor1, unsigned long long *__restrict__ hdr, const int64_t *__restrict__ todo,
    int probe, BucketPlan B, int64_t n, const unsigned long long *__restrict__ wgstat, int build_wgs) {
  if (todo != nullptr && *todo == 0) return;  // (hdr[kHdrFast] stays 0: the passes write the copy + identity)
  const int lane = (int)(threadIdx.x & 63);
  if (probe) {  // the probe's results: one slot per build workgroup -> hdr (sums of descents / duplicates, maxima of ids)
    unsigned long long v[4] = {0, 0, 0, 0};
    for (int g = (int)threadIdx.x; g < build_wgs; g += kBuildThreads) {
      v[0] += wgstat[(size_t)g * 4];
      v[1] += wgstat[(size_t)g * 4 + 1];
      if (probe == 2) {
        v[2] = wgstat[(size_t)g * 4 + 2] > v[2] ? wgstat[(size_t)g * 4 + 2] : v[2];
        v[3] = wgstat[(size_t)g * 4 + 3] > v[3] ? wgstat[(size_t)g * 4 + 3] : v[3];
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      v[0] += (unsigned long long)lane_xor((int64_t)v[0], off);
      v[1] += (unsigned long long)lane_xor((int64_t)v[1], off);
      const unsigned long long a = (unsigned long long)lane_xor((int64_t)v[2], off), b = (unsigned long long)lane_xor((int64_t)v[3], off);
      v[2] = a > v[2] ? a : v[2];
      v[3] = b > v[3] ? b : v[3];
    }
    __shared__ unsigned long long s_st[4][kBuildThreads / 64];
    if (lane == 0)
      for (int q = 0; q < 4; ++q) s_st[q][threadIdx.x >> 6] = v[q];
    __syncthreads();
    if (threadIdx.x < 4) {
      unsigned long long t = 0;
      for (int ww = 0; ww < kBuildThreads / 64; ++ww) {
        const unsigned long long x = s_st[threadIdx.x][ww];
        t = threadIdx.x < 2 ? t + x : (x > t ? x : t);
      }
      hdr[threadIdx.x == 0 ? kHdrDescents : (threadIdx.x == 1 ? kHdrDups : kHdrMaxRow + (int)threadIdx.x - 2)] = t;
    }
    __syncthreads();
  }
  __shared__ unsigned int sc[kBkMaxHist];
  for (int b = (int)threadIdx.x; b < B.nb; b += kBuildThreads) {  // (coalesced; the copies' loads in flight together)
    unsigned int part[kBkHistCopies], t = 0;
#pragma unroll
    for (int r = 0; r < kBkHistCopies; ++r) part[r] = bhist[(size_t)r * B.nb + b];
#pragma unroll
    for (int r = 0; r < kBkHistCopies; ++r) t += part[r];
    sc[b] = t;
  }
  __syncthreads();
  const int per = (B.nb + kBuildThreads - 1) / kBuildThreads;  // thread t owns the buckets [t * per, (t + 1) * per)
  unsigned int sum = 0, mx = 0;
  for (int j = 0; j < per; ++j) {
    const int b = (int)threadIdx.x * per + j;
    const unsigned int c = b < B.nb ? sc[b] : 0u;
    sum += c;
    mx = c > mx ? c : mx;
  }
  // exclusive scan of `sum` over the 1024 threads + the block maximum
  unsigned int inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int o = lane_read(inc, lane >= off ? lane - off : lane);
    if (lane >= off) inc += o;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned int o = lane_xor(mx, off);
    mx = o > mx ? o : mx;
  }
  __shared__ unsigned int s_wsum[kBuildThreads / 64], s_wmax[kBuildThreads / 64];
  if (lane == 63) s_wsum[threadIdx.x >> 6] = inc;
  if (lane == 0) s_wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  unsigned int base = 0, gmx = 0;
  for (int ww = 0; ww < kBuildThreads / 64; ++ww) {
    if (ww < (int)(threadIdx.x >> 6)) base += s_wsum[ww];
    gmx = s_wmax[ww] > gmx ? s_wmax[ww] : gmx;
  }
  unsigned int run = base + inc - sum;
  const int bits2 = B.bits - B.bits1;
  for (int j = 0; j < per; ++j) {
    const int b = (int)threadIdx.x * per + j;
    if (b < B.nb) {
      boff[b] = run;
      cursor[b] = run;  // the last scatter level reserves its runs here
      if (B.levels == 2 && (b & ((1 << bits2) - 1)) == 0) cursor1[b >> bits2] = run;  // level 1: one cursor per group of 2^bits2
      run += sc[b];
    }
  }
  if (threadIdx.x == 0) {
    boff[B.nb] = (unsigned int)n;
    bool fast = gmx <= (unsigned int)B.cap;
    // an input without descents is not sorted at all (the passes' last kernel writes the copy + identity)
    if (probe && hdr[kHdrDescents] == 0ull) fast = false;
    hdr[kHdrFast] = fast ? 1ull : 0ull;
  }
}

// exclusive scan of one value per thread over a 256-thread block (u64); smem: 4 words
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long *smem) {
  const int lane = (int)(threadIdx.x & 63), wid = (int)(threadIdx.x >> 6);
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = (unsigned long long)lane_read((int64_t)inc, lane >= off ? lane - off : lane);
    if (lane >= off) inc += o;
  }
  if (lane == 63) smem[wid] = inc;
  __syncthreads();
  unsigned long long base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < wid) base += smem[w];
  __syncthreads();
  return base + inc - v;
}

// ---------------------------------------------------------------------------
// one pass = one kernel
// ---------------------------------------------------------------------------
// VAL (packed words only, round 5): a 4-byte value rides along with its entry through EVERY pass as a 32-bit payload --
// the first pass reads it in input order (`gather_src[e]`: the entry at position e of the first pass IS entry e), the
// last pass stores it next to the decoded ids -- instead of being gathered through the permutation by the last pass
// (7.5 M random 4-byte reads: +125 us, against +12 bytes per entry and pass of streamed traffic).
template <bool PACKED, bool LAST, bool VAL = false, bool BALLOT = false>
__global__ __launch_bounds__(kSortThreads) void onesweep_pass_kernel(
    const unsigned long long *__restrict__ in, const unsigned int *__restrict__ idx_in,
    unsigned long long *__restrict__ out, unsigned int *__restrict__ idx_out, int64_t *__restrict__ row_out,
    int64_t *__restrict__ col_out, int64_t *__restrict__ perm_out, int64_t n, int shift, KeyLayout L,
    const unsigned long long *__restrict__ hist, unsigned long long *__restrict__ tile_state,
    unsigned long long *__restrict__ hdr, unsigned int epoch, const int64_t *__restrict__ todo,
    const int64_t *__restrict__ row, const int64_t *__restrict__ col, int64_t *__restrict__ counts_out,
    const void *__restrict__ gather_src, void *__restrict__ gather_dst, int gather_bytes, int check4) {
  static_assert(!VAL || PACKED, "a riding value needs the position inside the word");
  constexpr bool kPayload = !PACKED || VAL;  // a 32-bit word travels beside the 64-bit one
  constexpr int kSortItems = kItemsOf<PACKED, VAL>;
  constexpr int kSortTile = kSortThreads * kSortItems;
  if (hdr[kHdrFast] != 0) return;  // the bucket path has sorted this input
  if constexpr (LAST) {  // the probe's counters travel with the last pass (no separate kernel)
    if (counts_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
      counts_out[0] = (int64_t)hdr[kHdrDescents];
      counts_out[1] = (int64_t)hdr[kHdrDups];
      if (check4) {
        counts_out[2] = (int64_t)hdr[kHdrMaxRow];
        counts_out[3] = (int64_t)hdr[kHdrMaxCol];
      }
    }
  }
  if (todo != nullptr && *todo == 0) {
    // nothing had to be sorted: the last pass writes what a sort of a sorted input returns -- a copy and the identity
    if constexpr (LAST) {
      const int64_t t0 = (int64_t)blockIdx.x * kSortTile;
      for (int k = 0; k < kSortItems; ++k) {
        const int64_t e = t0 + k * kSortThreads + threadIdx.x;
        if (e >= n) break;
        if (row_out) row_out[e] = row[e];
        if (col_out) col_out[e] = col[e];
        if (perm_out) perm_out[e] = e;
        if (gather_dst != nullptr) {
          if (gather_bytes == 4) reinterpret_cast<uint32_t *>(gather_dst)[e] = reinterpret_cast<const uint32_t *>(gather_src)[e];
          else reinterpret_cast<uint64_t *>(gather_dst)[e] = reinterpret_cast<const uint64_t *>(gather_src)[e];
        }
      }
    }
    return;
  }
  __shared__ unsigned long long sword[kSortTile];
  __shared__ unsigned int sidx[kPayload ? kSortTile : 1];
  __shared__ unsigned int cnt[4][kRadix];
  __shared__ unsigned int dig_off[kRadix];
  __shared__ long long goff[kRadix];
  __shared__ unsigned long long sscan[4];
  __shared__ unsigned int s_tile;

  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
#if TSAMD_SORT_TICKET
  if (tid == 0) s_tile = (unsigned int)atomicAdd(&hdr[kHdrTicket + epoch - 1], 1ull);
#else
  if (tid == 0) s_tile = blockIdx.x;
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) cnt[i][tid] = 0;
  __syncthreads();
  const int64_t tile = (int64_t)s_tile;
  const int64_t tile0 = tile * kSortTile;
  const int64_t base = tile0 + (int64_t)w * (64 * kSortItems);

  unsigned long long word[kSortItems];
  unsigned int idx[kPayload ? kSortItems : 1];
  unsigned int dig[kSortItems], lrank[kSortItems];
  bool valid[kSortItems];
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int64_t e = base + i * 64 + lane;
    valid[i] = e < n;
    word[i] = valid[i] ? in[e] : 0ull;
    if constexpr (VAL) idx[i] = valid[i] ? (idx_in ? idx_in[e] : reinterpret_cast<const uint32_t *>(gather_src)[e]) : 0u;
    else if constexpr (!PACKED) idx[i] = valid[i] ? (idx_in ? idx_in[e] : (unsigned int)e) : 0u;
    dig[i] = (unsigned int)(word[i] >> shift) & (kRadix - 1);
  }
  // rank of every entry among the equal digits of its wave, in input order
  if constexpr (!BALLOT) {
    // One returning LDS atomic per entry.  When several lanes of ONE ds_add_rtn hit the same counter the LDS unit
    // serves them in ascending lane order on gfx950, so the value returned is the stable rank.  That order is not
    // an ISA guarantee: `sort_selftest_kernel` checks it once per process on the device and the host launches the
    // BALLOT instantiation (ballot matching: independent of the order, ~190 instead of ~40 instructions per entry
    // and pass) when the check fails; tests/test_sort_gpu.py forces both and compares the permutations.
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
      lrank[i] = 0;
      if (valid[i]) lrank[i] = atomicAdd(&cnt[w][dig[i]], 1u);
    }
  } else {
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
      unsigned long long peers = __ballot(valid[i]);
#pragma unroll
      for (int b = 0; b < kRadixBits; ++b) {
        const bool bit = (dig[i] >> b) & 1u;
        const unsigned long long m = __ballot(valid[i] && bit);
        peers &= bit ? m : ~m;
      }
      const unsigned int rank = (unsigned int)__popcll(peers & ((1ull << lane) - 1ull));
      const int leader = valid[i] ? (__ffsll((long long)peers) - 1) : lane;
      unsigned int pre = 0;
      if (valid[i] && lane == leader) {
        pre = cnt[w][dig[i]];
        cnt[w][dig[i]] = pre + (unsigned int)__popcll(peers);
      }
      pre = lane_read(pre, leader);
      lrank[i] = pre + rank;
    }
  }
  __syncthreads();

  // thread t owns digit t
  unsigned int mine = 0;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) {
    const unsigned int c = cnt[ww][tid];
    cnt[ww][tid] = mine;
    mine += c;
  }
  unsigned long long *my_state = tile_state + (size_t)tile * kRadix + tid;
  const unsigned long long ep = (unsigned long long)epoch << 56;
  st_store(my_state, (tile == 0 ? kFlagPrefix : kFlagLocal) | ep | (unsigned long long)mine);
  const unsigned long long ex = block_excl_scan_u64((unsigned long long)mine, sscan);    // position of the digit's run in the tile
  const unsigned long long gbase = block_excl_scan_u64(hist[tid], sscan);                // first output slot of the digit
  dig_off[tid] = (unsigned int)ex;
  __syncthreads();

  // reorder the tile in LDS (needs nothing from the other tiles) ...
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    if (valid[i]) {
      const unsigned int pos = dig_off[dig[i]] + cnt[w][dig[i]] + lrank[i];
      sword[pos] = word[i];
      if constexpr (kPayload) sidx[pos] = idx[i];
    }
  }
  // ... then look back: the entries with my digit in the tiles before this one
  unsigned long long before = 0;
#if defined(TSAMD_SORT_EXP_NOLOOKBACK)
  if (false) {
#else
  if (tile > 0) {
#endif
    int64_t t = tile - 1;
    unsigned int spins = 0;
    bool done = false;
    while (!done) {
      // the status words of kLook tiles at once (independent loads: one round trip), consumed nearest first
      constexpr int kLook = TSAMD_SORT_LOOK;
      unsigned long long sv[kLook];
#pragma unroll
      for (int u = 0; u < kLook; ++u) sv[u] = st_load(tile_state + (size_t)(t - u >= 0 ? t - u : 0) * kRadix + tid);
#pragma unroll
      for (int u = 0; u < kLook; ++u) {
        if (done) break;
        const unsigned long long sw = sv[u];
        if ((sw & (0x3full << 56)) != ep || (sw >> 62) == 0) {  // not published yet (for this pass): poll again from here
          if (++spins > kSpinLimit) {
            // ~1 s without the predecessor's status word: the dispatch-order assumption (design note above) does
            // not hold on this machine.  The result would be a silently wrong permutation -- abort the launch
            // instead (the HIP runtime reports the queue error), after leaving the reason in the header.
            hdr[kHdrError] = 1;
            __builtin_trap();
          }
          __builtin_amdgcn_s_sleep(1);
          break;
        }
        before += sw & kCountMask;
        if ((sw >> 62) == 2 || t == 0) done = true;
        else --t;
      }
    }
    st_store(my_state, kFlagPrefix | ep | (before + (unsigned long long)mine));
  }
  goff[tid] = (long long)(gbase + before) - (long long)ex;
  __syncthreads();

  const int64_t rem = n - tile0;
  const int count = rem < kSortTile ? (int)rem : kSortTile;
  if constexpr (LAST) {
    // batches of 8 entries per thread: positions and source ids first, then (values riding along) the 8 random reads
    // in flight together, then the stores -- one entry at a time the gather's latency was paid 16 times in a row
    constexpr int kBatch = 8;
#pragma unroll
    for (int k0 = 0; k0 < kSortItems; k0 += kBatch) {
      int64_t o[kBatch];
      unsigned long long key[kBatch], e[kBatch];
      bool ok[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int j = (k0 + k) * kSortThreads + tid;
        ok[k] = j < count;
        const unsigned long long wd = sword[ok[k] ? j : 0];
        const unsigned int d = (unsigned int)(wd >> shift) & (kRadix - 1);
        o[k] = goff[d] + j;
        if constexpr (PACKED) {
          key[k] = wd >> L.idx_bits;
          e[k] = wd & ((1ull << L.idx_bits) - 1ull);
        } else {
          key[k] = wd;
          e[k] = sidx[ok[k] ? j : 0];
        }
      }
      if constexpr (VAL) {  // the value came along: a coalesced store, no read
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
          if (ok[k]) reinterpret_cast<uint32_t *>(gather_dst)[o[k]] = sidx[(k0 + k) * kSortThreads + tid];
      } else if (gather_dst != nullptr) {
        if (gather_bytes == 4) {
          uint32_t v[kBatch];
#pragma unroll
          for (int k = 0; k < kBatch; ++k) v[k] = reinterpret_cast<const uint32_t *>(gather_src)[ok[k] ? e[k] : 0];
#pragma unroll
          for (int k = 0; k < kBatch; ++k)
            if (ok[k]) reinterpret_cast<uint32_t *>(gather_dst)[o[k]] = v[k];
        } else {
          uint64_t v[kBatch];
#pragma unroll
          for (int k = 0; k < kBatch; ++k) v[k] = reinterpret_cast<const uint64_t *>(gather_src)[ok[k] ? e[k] : 0];
#pragma unroll
          for (int k = 0; k < kBatch; ++k)
            if (ok[k]) reinterpret_cast<uint64_t *>(gather_dst)[o[k]] = v[k];
        }
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        if (!ok[k]) continue;
        if (row_out) row_out[o[k]] = (int64_t)(key[k] >> L.col_bits);
        if (col_out) col_out[o[k]] = (int64_t)(key[k] & ((1ull << L.col_bits) - 1ull));
        if (perm_out) perm_out[o[k]] = (int64_t)e[k];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
      const int j = k * kSortThreads + tid;
      if (j >= count) break;
      const unsigned long long wd = sword[j];
      const unsigned int d = (unsigned int)(wd >> shift) & (kRadix - 1);
      const int64_t o = goff[d] + j;
      out[o] = wd;
      if constexpr (kPayload) idx_out[o] = sidx[j];
    }
  }
}

// keys of zero bits (a 1 x 1 matrix), or a single entry: the input order is the sorted order
__global__ void sort_identity_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col, int64_t n,
                                     int64_t *__restrict__ row_out, int64_t *__restrict__ col_out,
                                     int64_t *__restrict__ perm_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (row_out) row_out[i] = row[i];
  if (col_out) col_out[i] = col[i];
  if (perm_out) perm_out[i] = i;
}

// ---------------------------------------------------------------------------
// Is the returning LDS atomic a stable rank on this device?  One wave per workgroup, 24 collision patterns (all lanes
// on one counter ... all lanes on different counters, strided, hashed): the value a lane gets back from ONE
// ds_add_rtn must be the number of LOWER lanes that hit the same counter.  flag[0] |= 1 when it is not.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sort_selftest_kernel(unsigned int *__restrict__ flag) {
  __shared__ unsigned int c[64];
  const int lane = (int)threadIdx.x;
  unsigned int bad = 0;
  for (int t = 0; t < 24; ++t) {
    c[lane] = 0;
    unsigned int d;
    if (t == 0) d = 0;
    else if (t < 8) d = (unsigned int)lane % (unsigned int)(t + 1);            // 2 .. 8 counters, interleaved
    else if (t < 14) d = (unsigned int)lane >> (t - 8);                         // runs of 1 .. 32 lanes
    else if (t < 23) d = (((unsigned int)lane + blockIdx.x) * 2654435761u >> (t + 3)) & (t & 1 ? 7u : 63u);  // hashed
    else d = (unsigned int)lane;
    d &= 63u;
    const unsigned int got = atomicAdd(&c[d], 1u);
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const unsigned int want = (unsigned int)__popcll(peers & ((1ull << lane) - 1ull));
    bad |= got != want;
    // the packed form of the bucket sort: two 16-bit counters per word, neighbours collide on one address
    c[lane] = 0;
    const unsigned int sh = (d & 1u) * 16u;
    const unsigned int got2 = (atomicAdd(&c[d >> 1], 1u << sh) >> sh) & 0xffffu;
    bad |= got2 != want;
  }
  if (__ballot(bad != 0) != 0ull && lane == 0) atomicOr(flag, 1u);
}

// ---------------------------------------------------------------------------
// bucket path, kernel 1 of 2: scatter the words into their buckets.  A tile counts its entries per bucket in LDS
// (any distinct rank inside the tile will do: the order inside a bucket is irrelevant, see above), reserves a run in
// every bucket with ONE global atomic per (tile, bucket) -- 64 neighbouring cursors per wave instruction -- and
// writes the tile bucket by bucket (reordered in LDS first, so that a run is one contiguous store).
// ---------------------------------------------------------------------------
constexpr int kBkScatterThreads = 256;
#ifndef TSAMD_BK_SCATTER_ITEMS
#define TSAMD_BK_SCATTER_ITEMS 30      // 8-byte entries: 60 KB + 16 KB of counters -> two workgroups per CU
#endif
#ifndef TSAMD_BK_SCATTER_ITEMS_VAL
#define TSAMD_BK_SCATTER_ITEMS_VAL 20  // 12-byte entries
#endif
template <bool VAL>
constexpr int kBkScatterItems = VAL ? TSAMD_BK_SCATTER_ITEMS_VAL : TSAMD_BK_SCATTER_ITEMS;

// a word and the value riding with it, as they lie in the bucket array of a sort with values: one 12-byte record, so
// that a tile's run in a bucket is ONE contiguous piece (two arrays: two pieces of 2-3 entries each per run)
struct __attribute__((packed, aligned(4))) BkRec {
  unsigned int lo, hi, v;
};

// (LDS index of bucket b: one pad word per 32 buckets, so that "thread t scans buckets 8t .. 8t + 7" is not a
// 16-way bank conflict)
__device__ __forceinline__ unsigned int bk_slot(unsigned int b) { return b + (b >> 5); }

// LEVEL 1 reads the built words / keys in input order (entry e IS position e; a riding value comes from `val_in[e]`)
// and separates them by the top bits1 bits of the bucket id; LEVEL 2 (plans with two levels) reads level 1's output
// and separates every level-1 bucket by the remaining bits: the tile's entries lie in a few neighbouring level-1
// buckets (found from the tile's position), its LDS counters cover the final buckets of the first kBkWin of them, and
// an entry beyond that window (a tile over many tiny level-1 buckets) reserves its slot by itself.
// In LDS a strip-mode LEVEL 1 tile holds (key << kBkTileBits | place in the tile): bucket id and final word both follow
// from it (the final word no longer has the bucket's bits, and key + full position may not fit 64 bits).
constexpr int kBkTileBits = 13;
constexpr int kBkWinMax = 32;

template <bool VAL, int LEVEL, bool STRIP>
__global__ __launch_bounds__(kBkScatterThreads) void bucket_scatter_kernel(
    const unsigned long long *__restrict__ in, const unsigned int *__restrict__ val_in, int64_t n, BucketPlan B,
    KeyLayout L, const unsigned int *__restrict__ boff, unsigned int *__restrict__ cursor,
    unsigned long long *__restrict__ out, const unsigned long long *__restrict__ hdr) {
  if (hdr[kHdrFast] == 0) return;
  constexpr int kItems = kBkScatterItems<VAL>;
  constexpr int kTile = kBkScatterThreads * kItems;
  static_assert(kTile <= (1 << kBkTileBits), "place in the tile: kBkTileBits bits");
  constexpr int kPer = kBkMaxBuckets / kBkScatterThreads;
  constexpr int kSlots = kBkMaxBuckets + kBkMaxBuckets / 32;
  __shared__ unsigned long long sword[kTile];
  __shared__ unsigned int sval[VAL ? kTile : 1];
  __shared__ unsigned int cnt[kSlots];   // entries of the tile per bucket, later: first output slot - first LDS slot
  __shared__ unsigned int loff[kSlots];  // first LDS slot of the bucket's run
  __shared__ unsigned int s_wsum[kBkScatterThreads / 64];
  __shared__ unsigned int sbnd[kBkWinMax];  // LEVEL 2: first position of the level-1 buckets behind the tile's first
  __shared__ unsigned int s_direct;         // LEVEL 2: entries outside the counter window
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int b = tid; b < kSlots; b += kBkScatterThreads) cnt[b] = 0;
  if (tid == 0) s_direct = 0;
  const int64_t tile0 = (int64_t)blockIdx.x * kTile;
  const int count = n - tile0 < kTile ? (int)(n - tile0) : kTile;
  const int bits2 = B.bits - B.bits1;
  const unsigned int nb1 = 1u << B.bits1;
  // the buckets this tile counts in LDS: [gid0, gid0 + nloc)
  unsigned int gid0 = 0, b1_first = 0;
  int nloc = 1 << B.bits1, nwin = 1;
  if constexpr (LEVEL == 2) {
    // level-1 bucket of the tile's first entry: the last b with boff[b << bits2] <= tile0
    unsigned int lo = 0, hi = nb1 - 1u;
    while (lo < hi) {
      const unsigned int mid = (lo + hi + 1) >> 1;
      if ((int64_t)boff[mid << bits2] <= tile0) lo = mid; else hi = mid - 1;
    }
    b1_first = lo;
    gid0 = lo << bits2;
    nwin = kBkMaxBuckets >> bits2;
    nwin = nwin > kBkWinMax ? kBkWinMax : nwin;
    nwin = (int)(nb1 - lo) < nwin ? (int)(nb1 - lo) : nwin;
    nloc = nwin << bits2;
    if (tid < kBkWinMax) sbnd[tid] = b1_first + 1u + (unsigned int)tid < nb1 ? boff[(b1_first + 1u + (unsigned int)tid) << bits2] : 0xffffffffu;
  }
  __syncthreads();
  unsigned long long word[kItems];
  unsigned int val[VAL ? kItems : 1], rank[kItems], bid2[LEVEL == 2 ? kItems : 1];
  // the bucket (relative to gid0) of a staged word; LEVEL 2 keeps it in bid2 (it depends on where the entry came from)
  const int kb1 = L.key_bits - B.bits1;
  auto bucket_of = [&](unsigned long long wd) -> unsigned int {
    if constexpr (STRIP) return (unsigned int)((wd >> kBkTileBits) >> kb1);
    else return (unsigned int)(wd >> B.shift);
  };
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int j = i * kBkScatterThreads + tid;
    word[i] = 0ull;
    if constexpr (VAL) val[i] = 0u;
    if (j < count) {
      if constexpr (LEVEL == 1) {
        word[i] = in[tile0 + j];
        if constexpr (VAL) val[i] = val_in[tile0 + j];
      } else if constexpr (VAL) {
        const BkRec rec = reinterpret_cast<const BkRec *>(in)[tile0 + j];
        word[i] = ((unsigned long long)rec.hi << 32) | rec.lo;
        val[i] = rec.v;
      } else {
        word[i] = in[tile0 + j];
      }
    }
  }
  int dl = 0;  // LEVEL 2: this thread's entries come in rising positions, their level-1 bucket only moves forward
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int j = i * kBkScatterThreads + tid;
    rank[i] = 0;
    if constexpr (LEVEL == 2) bid2[i] = 0;
    if (j < count) {
      unsigned int bid;
      if constexpr (LEVEL == 1) {
        if constexpr (STRIP) {
          const unsigned long long key = L.packed ? word[i] >> L.idx_bits : word[i];
          word[i] = (key << kBkTileBits) | (unsigned long long)j;
        }
        bid = bucket_of(word[i]);
      } else {
        const unsigned int p = (unsigned int)(tile0 + j);
        while (dl < kBkWinMax && sbnd[dl] <= p) ++dl;
        // (dl >= nwin: outside the window -- the true level-1 bucket is looked up when the entry is stored)
        bid = dl < nwin ? ((unsigned int)dl << bits2) + (unsigned int)(word[i] >> B.shift) : 0xffffffffu;
        bid2[i] = bid;
      }
      if (LEVEL == 1 || bid != 0xffffffffu) {
#if defined(TSAMD_EXP_SCATTER_NO_RANK)  // timing experiment
        rank[i] = cnt[bk_slot(bid)];
#else
        rank[i] = atomicAdd(&cnt[bk_slot(bid)], 1u);
#endif
      }
    }
  }
  __syncthreads();
  // exclusive scan of the counts in bucket order: thread t takes the kPer buckets from t * kPer on
  unsigned int c[kPer], sum = 0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const unsigned int b = (unsigned int)(tid * kPer + j);
    c[j] = (int)b < nloc ? cnt[bk_slot(b)] : 0u;
    sum += c[j];
  }
  unsigned int inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int o = lane_read(inc, lane >= off ? lane - off : lane);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_wsum[w] = inc;
  __syncthreads();
  unsigned int run = inc - sum;
#pragma unroll
  for (int ww = 0; ww < kBkScatterThreads / 64; ++ww)
    if (ww < w) run += s_wsum[ww];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const unsigned int b = (unsigned int)(tid * kPer + j);
    if ((int)b < nloc) loff[bk_slot(b)] = run;
    run += c[j];
  }
  __syncthreads();
  // reserve the runs: bucket k * 256 + t (neighbouring cursors per wave instruction); the results are needed last
  unsigned int gbase[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const unsigned int b = (unsigned int)(k * kBkScatterThreads + tid);
    const unsigned int cb = (int)b < nloc ? cnt[bk_slot(b)] : 0u;
#if defined(TSAMD_EXP_SCATTER_NO_CURSOR)  // timing experiment (scripts/variants.py): wrong result
    gbase[k] = cursor[gid0 + b];
#else
    gbase[k] = cb ? atomicAdd(&cursor[gid0 + b], cb) : 0u;
#endif
  }
  auto store_rec = [&](unsigned int o, unsigned long long wd, unsigned int v) {
    if constexpr (VAL) {
      BkRec rec;
      rec.lo = (unsigned int)wd;
      rec.hi = (unsigned int)(wd >> 32);
      rec.v = v;
      reinterpret_cast<BkRec *>(out)[o] = rec;
    } else {
      out[o] = wd;
    }
  };
  // reorder the tile by bucket in LDS
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int j = i * kBkScatterThreads + tid;
    if (j < count) {
      unsigned int bid;
      if constexpr (LEVEL == 1) bid = bucket_of(word[i]);
      else bid = bid2[i];
      if (LEVEL == 1 || bid != 0xffffffffu) {
        const unsigned int pos = loff[bk_slot(bid)] + rank[i];
        sword[pos] = word[i];
        if constexpr (VAL) sval[pos] = val[i];
      } else {  // outside the counter window: a slot of its own, stored at once
        const unsigned int p = (unsigned int)(tile0 + j);
        unsigned int lo = b1_first, hi = nb1 - 1u;
        while (lo < hi) {
          const unsigned int mid = (lo + hi + 1) >> 1;
          if (boff[mid << bits2] <= p) lo = mid; else hi = mid - 1;
        }
        store_rec(atomicAdd(&cursor[(lo << bits2) + (unsigned int)(word[i] >> B.shift)], 1u), word[i], VAL ? val[i] : 0u);
        atomicAdd(&s_direct, 1u);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const unsigned int b = (unsigned int)(k * kBkScatterThreads + tid);
    if ((int)b < nloc) cnt[bk_slot(b)] = gbase[k] - loff[bk_slot(b)];  // (modulo 2^32: E < 2^32)
  }
  __syncthreads();
  const int staged = LEVEL == 1 ? count : count - (int)s_direct;
  int dlo = 0;  // LEVEL 2: level-1 bucket (relative to the tile's first) of LDS slot j; rises with j
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int j = i * kBkScatterThreads + tid;
    if (j >= staged) break;
    unsigned long long wd = sword[j];
    unsigned int b;
    if constexpr (LEVEL == 1) {
      b = bucket_of(wd);
      if constexpr (STRIP) {  // the word of the bucket array: the key without its bits1 top bits, then the position
        const unsigned long long key = wd >> kBkTileBits;
        wd = ((key & ((1ull << kb1) - 1ull)) << L.idx_bits) |
             (unsigned long long)(tile0 + (int64_t)(wd & ((1u << kBkTileBits) - 1u)));
      }
    } else {
      while (dlo + 1 < nwin && loff[bk_slot((unsigned int)(dlo + 1) << bits2)] <= (unsigned int)j) ++dlo;
      b = ((unsigned int)dlo << bits2) + (unsigned int)(wd >> B.shift);
    }
    const unsigned int o = cnt[bk_slot(b)] + (unsigned int)j;
#if defined(TSAMD_EXP_SCATTER_LINEAR_STORE)  // timing experiment: the tile goes out as one contiguous piece
    store_rec((unsigned int)(tile0 + j), wd ^ (unsigned long long)(o & 1u), VAL ? sval[j] : 0u);
#else
    store_rec(o, wd, VAL ? sval[j] : 0u);
#endif
  }
}

// ---------------------------------------------------------------------------
// bucket path, kernel 2 of 2: one workgroup sorts one bucket inside LDS and writes the decoded rows / columns /
// permutation (and the value that rode along) as one contiguous, coalesced piece of the output.
//   1. 8-bit LSD passes over the TOP `SB.n_top` digits below the bucket id (16 bits: a bucket of a few thousand
//      entries falls into ~65 k groups), entries in registers between the passes: wave w owns a contiguous segment of
//      the bucket and ranks by one returning LDS atomic per entry on a per-wave counter (two 16-bit counters per
//      word); the BALLOT instantiation ranks by ballot matching (see the pass kernel);
//   2. FINISH: what is left to order are the entries that agree in all of those bits -- neighbours in LDS now.  While
//      the output is written every entry counts the larger words of its group to its left and the smaller ones to its
//      right (full 64-bit compare: words are unique) and goes to place `j - greater + smaller` of the bucket's piece
//      of the output.  Typically one read per side and no match;
//   3. a group longer than kBkGroupMax (many duplicates of one key, a dense block) makes the workgroup sort the whole
//      bucket by LSD passes over ALL the bits below the bucket id first (SB.n passes): fixed cost, any input.
// ---------------------------------------------------------------------------
constexpr int kBkGroupMax = 24;

// COAL (functional coalesce / transpose): the bucket's piece of the output is COMPACTED as it is written -- the
// first entry of every run of equal keys goes to row_out / col_out (here: the distinct pairs), its position in the
// sorted order to Co.seg_ptr, the sorted values (all of them) to gather_dst; the number of distinct pairs in the
// buckets before this one comes from a decoupled look-back over one status word per bucket (as in
// coalesce_compact_kernel; buckets = workgroups in dispatch order).  Saves writing the sorted ids and the permutation
// (24 bytes per entry) and reading them back in a compaction kernel.
struct CoalesceOut {
  int64_t *seg_ptr;            // [E + 1]
  int64_t *nnz_out;            // [1]
  unsigned long long *status;  // [#buckets], zeroed: [63:62] 1 = own count, 2 = inclusive prefix
  int64_t n_total;
  // fused reduction of the riding values (VAL launches, reduce >= 0): value_u[p] = REDUCE over the p-th run, in sorted
  // order, in the accumulator type of segment_reduce_kernel (coalesce.hip) -- neither seg_ptr nor the sorted values
  // are written then
  void *value_u;
  int64_t *fused_out;
  int reduce;    // -1 none, 0 sum, 1 mean, 2 min, 3 max
  int is_float;
};

// REDUCE over sval[j .. j + r) (r >= 1), the bits of segment_reduce_kernel<float / int32_t>
template <typename A>
__device__ __forceinline__ unsigned int bk_reduce_run(const unsigned int *sval, int j, int r, int reduce) {
  A acc;
  unsigned int bits = sval[j];
  __builtin_memcpy(&acc, &bits, 4);
  for (int q = 1; q < r; ++q) {
    A v;
    bits = sval[j + q];
    __builtin_memcpy(&v, &bits, 4);
    if (reduce == 2) acc = v < acc ? v : acc;
    else if (reduce == 3) acc = v > acc ? v : acc;
    else acc += v;
  }
  if (reduce == 1) {
    if constexpr (std::is_integral<A>::value) {  // floor division, as torch_scatter does
      A q = acc / (A)r;
      if ((acc % (A)r != 0) && (acc < 0)) --q;
      acc = q;
    } else {
      acc = acc / (A)r;
    }
  }
  __builtin_memcpy(&bits, &acc, 4);
  return bits;
}

template <int THREADS, int ITEMS, bool VAL, bool BALLOT, bool COAL = false>
__global__ __launch_bounds__(THREADS, (2 * THREADS / 256)) void bucket_sort_kernel(
    const unsigned long long *__restrict__ in, const unsigned int *__restrict__ val_in,
    const unsigned int *__restrict__ boff, SortBits SB, KeyLayout L, BucketPlan B, int64_t *__restrict__ row_out,
    int64_t *__restrict__ col_out, int64_t *__restrict__ perm_out, const unsigned long long *__restrict__ hdr,
    int64_t *__restrict__ counts_out, int check4, const void *__restrict__ gather_src, void *__restrict__ gather_dst,
    int gather_bytes, CoalesceOut Co) {
  if (hdr[kHdrFast] == 0) return;
  if (counts_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {  // the probe's counters (see the last pass kernel)
    counts_out[0] = (int64_t)hdr[kHdrDescents];
    counts_out[1] = (int64_t)hdr[kHdrDups];
    if (check4) {
      counts_out[2] = (int64_t)hdr[kHdrMaxRow];
      counts_out[3] = (int64_t)hdr[kHdrMaxCol];
    }
  }
  constexpr int kW = THREADS / 64, kCap = THREADS * ITEMS;
  static_assert(THREADS >= 128 && kCap < 65536, "two waves scan the digits; 16-bit counters");
  __shared__ unsigned long long sword[kCap];
  __shared__ unsigned int sval[VAL ? kCap : 1];
  __shared__ unsigned int cnt[kW][kRadix / 2];
  __shared__ unsigned int dig_off[kRadix];
  __shared__ unsigned int s_w0;
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned int start = boff[blockIdx.x];
  const int n = (int)(boff[blockIdx.x + 1] - start);
  if (n == 0 && !COAL) return;  // (a compacting launch: an empty bucket still passes the count of its predecessors on)
  // wave w owns the entries [w * seg, (w + 1) * seg), seg a multiple of 64: the waves share the bucket evenly
  const int seg = (((n + kW - 1) / kW) + 63) & ~63;
  const int items = seg >> 6;  // <= ITEMS because n <= kCap
  const int wbase = w * seg;
  unsigned long long word[ITEMS];
  unsigned int val[VAL ? ITEMS : 1];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    word[i] = 0ull;
    if constexpr (VAL) val[i] = 0u;
    const int e = wbase + i * 64 + lane;
    if (i < items && e < n) {
      if constexpr (VAL) {
        const BkRec rec = reinterpret_cast<const BkRec *>(in)[start + e];
        word[i] = ((unsigned long long)rec.hi << 32) | rec.lo;
        val[i] = rec.v;
      } else {
        word[i] = in[start + e];
      }
    }
  }
  // LSD passes p0 .. p1 - 1 of SB over the entries in registers; leaves the result in sword / sval (LDS)
  auto lsd_passes = [&](int p0, int p1) {
    for (int p = p0; p < p1; ++p) {
      const int shift = SB.shift[p];
      const unsigned int mask = (1u << SB.width[p]) - 1u;
      unsigned int lrank[ITEMS];
      cnt[w][lane] = 0;  // this wave's counters (wave-local: LDS operations of one wave stay in order)
      cnt[w][lane + 64] = 0;
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        lrank[i] = 0;
        if (i < items) {
          const bool valid = wbase + i * 64 + lane < n;
          const unsigned int d = (unsigned int)(word[i] >> shift) & mask;
          const unsigned int sh = (d & 1u) * 16u;
          if constexpr (!BALLOT) {
            if (valid) lrank[i] = (atomicAdd(&cnt[w][d >> 1], 1u << sh) >> sh) & 0xffffu;
          } else {
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < kRadixBits; ++b) {
              const bool bit = (d >> b) & 1u;
              const unsigned long long m = __ballot(valid && bit);
              peers &= bit ? m : ~m;
            }
            const unsigned int rank = (unsigned int)__popcll(peers & ((1ull << lane) - 1ull));
            const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
            unsigned int pre = 0;
            if (valid && lane == leader) pre = (atomicAdd(&cnt[w][d >> 1], (unsigned int)__popcll(peers) << sh) >> sh) & 0xffffu;
            pre = lane_read(pre, leader);
            lrank[i] = pre + rank;
          }
        }
      }
      __syncthreads();
      // thread t < 128 owns the digits 2t and 2t + 1: exclusive prefix over the waves, then over the digits
      if (tid < kRadix / 2) {
        unsigned int run0 = 0, run1 = 0;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww) {
          const unsigned int v = cnt[ww][tid];
          cnt[ww][tid] = run0 | (run1 << 16);
          run0 += v & 0xffffu;
          run1 += v >> 16;
        }
        const unsigned int both = run0 + run1;
        unsigned int inc = both;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned int o = lane_read(inc, lane >= off ? lane - off : lane);
          if (lane >= off) inc += o;
        }
        if (tid == 63) s_w0 = inc;
        dig_off[2 * tid] = inc - both;
        dig_off[2 * tid + 1] = inc - both + run0;
      }
      __syncthreads();
      if (tid >= 64 && tid < kRadix / 2) {  // the second wave's digits start behind the first wave's
        const unsigned int b0 = s_w0;
        dig_off[2 * tid] += b0;
        dig_off[2 * tid + 1] += b0;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        if (i < items && wbase + i * 64 + lane < n) {
          const unsigned int d = (unsigned int)(word[i] >> shift) & mask;
          const unsigned int pos = dig_off[d] + ((cnt[w][d >> 1] >> ((d & 1u) * 16u)) & 0xffffu) + lrank[i];
          sword[pos] = word[i];
          if constexpr (VAL) sval[pos] = val[i];
        }
      }
      __syncthreads();
      if (p + 1 < p1) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          const int e = wbase + i * 64 + lane;
          if (i < items && e < n) {
            word[i] = sword[e];
            if constexpr (VAL) val[i] = sval[e];
          }
        }
        // (the next pass's scatter into sword comes two barriers later)
      }
    }
  };
  const int p_top = SB.n - SB.n_top;  // the top digits are the last n_top passes of the list
#if defined(TSAMD_EXP_BSORT_ONE_PASS)  // timing experiments: wrong result
  lsd_passes(SB.n - 1, SB.n);
#else
  lsd_passes(p_top, SB.n);
#endif
  // FINISH.  Groups = runs of entries that agree in every bit from SB.lo up; an entry's final place is its place in
  // the run, taken while the output is written (no second trip through LDS).  First: is any group too long for that?
  // (sorted by prefix: a group longer than G exists exactly when some entry and the entry G places on agree)
  const int lo = SB.lo;
  bool exact = p_top == 0;
#if defined(TSAMD_EXP_BSORT_NO_FINISH)
  exact = true;
#endif
  if (!exact) {
    int over = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int j = k * THREADS + tid;
      if (j + kBkGroupMax < n) over |= (sword[j] >> lo) == (sword[j + kBkGroupMax] >> lo);
    }
    if (__syncthreads_or(over)) {
      // a long group: the whole bucket by LSD passes over every bit below the bucket id (stable, any input)
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        const int e = wbase + i * 64 + lane;
        if (i < items && e < n) {
          word[i] = sword[e];
          if constexpr (VAL) val[i] = sval[e];
        }
      }
      lsd_passes(0, SB.n);
      exact = true;
    }
  }
  const unsigned long long imask = (1ull << L.idx_bits) - 1ull, cmask = (1ull << L.col_bits) - 1ull;
  // strip mode: the words lost the level-1 bits of the key in the scatter; the bucket id has them
  const unsigned long long keybase =
      B.strip ? (unsigned long long)(blockIdx.x >> (B.bits - B.bits1)) << (L.key_bits - B.bits1) : 0ull;
  if constexpr (COAL) {
    __shared__ unsigned int s_heads[ITEMS * kW];              // heads of (step k, wave w) at [k * kW + w], then their exclusive prefix
    __shared__ unsigned long long s_hbits[ITEMS * kW + 1];    // head flags, bit j = entry j (word j >> 6 = k * kW + w)
    __shared__ unsigned long long s_base;
    __shared__ unsigned int s_total;
    static_assert(ITEMS * kW <= 128, "one wave scans the per-(step, wave) head counts, two per lane");
    const bool fuse = VAL && Co.reduce >= 0;
    // 1. the exact order in LDS (the finish step moves what it otherwise only re-addresses)
#if defined(TSAMD_EXP_COAL_NO_EXACT)  // timing experiments (scripts/variants.py): wrong result
    exact = true;
#endif
    if (!exact) {
      unsigned long long fw[ITEMS];
      unsigned int fv[VAL ? ITEMS : 1];
      int fpos[ITEMS];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        const int j = k * THREADS + tid;
        fpos[k] = -1;
        fw[k] = 0ull;
        if (k * THREADS < n && j < n) {
          const unsigned long long x = sword[j];
          const unsigned long long pre = x >> lo;
          int gt = 0, lt = 0;
          for (int q = j - 1; q >= 0; --q) {
            const unsigned long long y = sword[q];
            if ((y >> lo) != pre) break;
            gt += y > x;
          }
          for (int q = j + 1; q < n; ++q) {
            const unsigned long long y = sword[q];
            if ((y >> lo) != pre) break;
            lt += y < x;
          }
          fw[k] = x;
          if constexpr (VAL) fv[k] = sval[j];
          fpos[k] = j - gt + lt;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        if (fpos[k] >= 0) {
          sword[fpos[k]] = fw[k];
          if constexpr (VAL) sval[fpos[k]] = fv[k];
        }
      }
      __syncthreads();
    }
    // 2. head flags (an entry whose key differs from its predecessor's; the bucket's first entry always: other
    //    buckets hold other keys) and their count per (step, wave)
    // (the flags and counts live in LDS, not in registers: as 12-16 ballot words + 12-16 prefixes per thread they pushed
    // this instantiation past its 128 VGPRs -- 16-61 spilled -- and every thread summed the 96-128 counts by itself)
    unsigned int heads = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int j = k * THREADS + tid;
      bool h = false;
      if (k * THREADS < n && j < n) h = j == 0 || (sword[j] >> L.idx_bits) != (sword[j - 1] >> L.idx_bits);
      const unsigned long long hm = __ballot(h);
      heads |= (h ? 1u : 0u) << k;
      if (lane == 0) {
        s_heads[k * kW + w] = (unsigned int)__popcll(hm);
        s_hbits[k * kW + w] = hm;
      }
    }
    if (tid == 0) s_hbits[ITEMS * kW] = 0ull;
    __syncthreads();
    if (w == 0) {  // exclusive prefix of the ITEMS * kW counts in (step, wave) order: two per lane
      constexpr int kCnt = ITEMS * kW;
      const unsigned int v0 = lane < kCnt ? s_heads[lane] : 0u, v1 = lane + 64 < kCnt ? s_heads[lane + 64] : 0u;
      unsigned int i0 = v0, i1 = v1;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o0 = lane_read(i0, lane >= off ? lane - off : lane), o1 = lane_read(i1, lane >= off ? lane - off : lane);
        if (lane >= off) {
          i0 += o0;
          i1 += o1;
        }
      }
      const unsigned int t0 = lane_read(i0, 63);
      if (lane < kCnt) s_heads[lane] = i0 - v0;
      if (lane + 64 < kCnt) s_heads[lane + 64] = t0 + i1 - v1;
      if (lane == 63) s_total = t0 + i1;
    }
    __syncthreads();
    const unsigned int total = s_total;
    // 3. distinct pairs in the buckets before this one: publish, look back
    constexpr unsigned long long kLocal = 1ull << 62, kPrefix = 2ull << 62, kMask = (1ull << 62) - 1ull;
    const int64_t bucket = (int64_t)blockIdx.x;
    if (w == 0 && lane == 0)
      __hip_atomic_store(Co.status + bucket, (bucket == 0 ? kPrefix : kLocal) | (unsigned long long)total, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    // the sorted values do not depend on the look-back: their stores are in flight while wave 0 waits for the
    // buckets before this one (and the own count is out before them)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int j = k * THREADS + tid;
      if (k * THREADS >= n || j >= n) continue;
      if constexpr (VAL) {
        if (!fuse) reinterpret_cast<uint32_t *>(gather_dst)[(size_t)start + j] = sval[j];
      } else if (gather_dst != nullptr) {
        const unsigned long long e = sword[j] & imask;
        if (gather_bytes == 4) reinterpret_cast<uint32_t *>(gather_dst)[(size_t)start + j] = reinterpret_cast<const uint32_t *>(gather_src)[e];
        else reinterpret_cast<uint64_t *>(gather_dst)[(size_t)start + j] = reinterpret_cast<const uint64_t *>(gather_src)[e];
      }
    }
    // look-back over lbw x 64 status words per round trip (wave w reads the 64 buckets behind the 64 * w nearer
    // ones; shipped: wave 0 alone).  The ~512 workgroups in flight finish their sorts together, so none of them finds
    // an inclusive prefix nearby: bucket b of a batch spends ~b / 64 dependent round trips here.
    {
      // (measured and rejected, profiles/r06_ab_coalesce_fused.md: every wave of the workgroup reading its own 64 words per
      // round trip -- 146 against 133 us for this kernel at 7.5 M entries: the words are device-scope loads, i.e. fabric
      // round trips, and eight times as many of them queue behind the stores of the other workgroups)
      // (also rejected: 64 words in the first round trip and kW x 64 in every later one of the same bucket -- the ripple
      // through the first ~512 buckets in 2 hops instead of 8 -- kernel 113.8 vs 112.3 us, coalesce 0.370 / 0.376 vs
      // 0.369 / 0.361 ms: the chain is not what this kernel waits for either; -DTSAMD_EXP_COAL_WIDEN_LOOKBACK)
#if defined(TSAMD_EXP_COAL_WIDE_LOOKBACK)
      int lbw = kW;
#else
      int lbw = 1;
#endif
#if defined(TSAMD_EXP_COAL_WIDEN_LOOKBACK)
      constexpr bool kWiden = true;
#else
      constexpr bool kWiden = false;
#endif
      __shared__ unsigned long long s_lb_sum[kW];
      __shared__ int s_lb_state[kW];  // 0: the wave's 64 buckets are all "own count"; 1: an inclusive prefix ends the sum here; 2: a bucket is not ready
      unsigned long long before = 0;
      int64_t t = bucket - 1;
#if defined(TSAMD_EXP_COAL_NO_LOOKBACK)
      t = -1;
      before = start;
#endif
      unsigned int spins = 0;
      while (t >= 0) {
        const int64_t mt = t - tid;
        unsigned long long sv = kPrefix;  // threads past bucket 0 read as "prefix 0"
        if (mt >= 0 && w < lbw) sv = __hip_atomic_load(Co.status + mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ready = __ballot((sv >> 62) != 0);
        const unsigned long long pref = __ballot((sv >> 62) == 2);
        const int first_gap = ~ready ? __builtin_ctzll(~ready) : 64;
        const int first_pref = pref ? __builtin_ctzll(pref) : 64;
        const int take = first_pref < first_gap ? first_pref + 1 : first_gap;
        unsigned long long v = lane < take ? (sv & kMask) : 0ull;
        for (int off = 32; off > 0; off >>= 1) v += (unsigned long long)lane_xor((int64_t)v, off);
        if (lane == 0) {
          s_lb_sum[w] = v;
          s_lb_state[w] = first_pref < first_gap ? 1 : (first_gap < 64 ? 2 : 0);
        }
        __syncthreads();
        // every thread folds the waves' pieces in order (nearest buckets first) and reaches the same verdict
        unsigned long long acc = 0;
        int verdict = 0;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww) {
          if (ww < lbw && verdict == 0) {
            const int stt = s_lb_state[ww];
            if (stt != 2) acc += s_lb_sum[ww];
            verdict = stt;
          }
        }
        __syncthreads();  // (the pieces are rewritten by the next round)
        if (verdict == 1) {
          before += acc;
          break;
        }
        if (verdict == 0) {  // THREADS buckets of own counts: further back
          before += acc;
          t -= 64 * lbw;
          if (kWiden) lbw = kW;
        } else {  // a bucket in the window has not published yet: read the window again
          if (++spins > kSpinLimit) {  // (see the pass kernel: the dispatch-order assumption does not hold here)
            __builtin_trap();
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (tid == 0) {
        unsigned long long *mine = Co.status + bucket;
        if (bucket > 0)
          __hip_atomic_store(mine, kPrefix | (before + (unsigned long long)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base = before;
        if (bucket == (int64_t)gridDim.x - 1) {
          Co.nnz_out[0] = (int64_t)(before + total);
          if (Co.fused_out != nullptr) Co.fused_out[0] = fuse ? 1 : 0;
          if (!fuse && Co.seg_ptr != nullptr) Co.seg_ptr[before + total] = Co.n_total;
        }
      }
    }
    __syncthreads();
    // 4. the distinct pairs and where their runs start in the sorted order
    const int64_t base = (int64_t)s_base;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int j = k * THREADS + tid;
      if (k * THREADS >= n || j >= n) continue;
#if defined(TSAMD_EXP_COAL_NO_HEAD_STORES)
      if (((heads >> k) & 1u) && sword[j] == 0x123456789ull) {
#else
      if ((heads >> k) & 1u) {
#endif
        const unsigned long long wd = sword[j];
        const int64_t p = base + s_heads[k * kW + w] + (unsigned int)__popcll(s_hbits[k * kW + w] & lt_mask);
        const unsigned long long key = keybase | (wd >> L.idx_bits);
        row_out[p] = (int64_t)(key >> L.col_bits);
        col_out[p] = (int64_t)(key & cmask);
        if constexpr (VAL) {
          if (fuse) {
            // the run ends at the next head (or at the end of the bucket: other buckets hold other keys)
            int wi = (j + 1) >> 6;
            unsigned long long bits = s_hbits[wi] & (~0ull << ((j + 1) & 63));
            while (bits == 0ull && (wi + 1) * 64 < n) bits = s_hbits[++wi];
            int next = bits ? wi * 64 + __builtin_ctzll(bits) : n;
            next = next < n ? next : n;
            const int r = next - j;
            reinterpret_cast<uint32_t *>(Co.value_u)[p] =
                Co.is_float ? bk_reduce_run<float>(sval, j, r, Co.reduce) : bk_reduce_run<int32_t>(sval, j, r, Co.reduce);
            continue;
          }
        }
        if (Co.seg_ptr != nullptr) Co.seg_ptr[p] = (int64_t)start + j;
      }
    }
    return;
  }
  // decoded output, coalesced
  constexpr int kBatch = 4;
  for (int j0 = tid; j0 < n; j0 += THREADS * kBatch) {
    unsigned long long key[kBatch], e[kBatch];
    size_t o[kBatch];
    bool ok[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int j = j0 + k * THREADS;
      ok[k] = j < n;
      const unsigned long long wd = sword[ok[k] ? j : 0];
      int pos = j;
      if (!exact && ok[k]) {
        const unsigned long long pre = wd >> lo;
        int gt = 0, lt = 0;
        for (int q = j - 1; q >= 0; --q) {  // (typically: one read, no match)
          const unsigned long long y = sword[q];
          if ((y >> lo) != pre) break;
          gt += y > wd;
        }
        for (int q = j + 1; q < n; ++q) {
          const unsigned long long y = sword[q];
          if ((y >> lo) != pre) break;
          lt += y < wd;
        }
        pos = j - gt + lt;
      }
      o[k] = (size_t)start + (size_t)pos;
      key[k] = keybase | (wd >> L.idx_bits);
      e[k] = wd & imask;
    }
    if constexpr (VAL) {
#pragma unroll
      for (int k = 0; k < kBatch; ++k)
        if (ok[k]) reinterpret_cast<uint32_t *>(gather_dst)[o[k]] = sval[j0 + k * THREADS];
    } else if (gather_dst != nullptr) {
      if (gather_bytes == 4) {
        uint32_t v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) v[k] = reinterpret_cast<const uint32_t *>(gather_src)[ok[k] ? e[k] : 0];
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
          if (ok[k]) reinterpret_cast<uint32_t *>(gather_dst)[o[k]] = v[k];
      } else {
        uint64_t v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) v[k] = reinterpret_cast<const uint64_t *>(gather_src)[ok[k] ? e[k] : 0];
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
          if (ok[k]) reinterpret_cast<uint64_t *>(gather_dst)[o[k]] = v[k];
      }
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      if (!ok[k]) continue;
#if defined(TSAMD_EXP_BSORT_NO_STORE)
      if (key[k] == 0x123456789ull) perm_out[o[k]] = (int64_t)e[k];
#else
      if (row_out) row_out[o[k]] = (int64_t)(key[k] >> L.col_bits);
      if (col_out) col_out[o[k]] = (int64_t)(key[k] & cmask);
      if (perm_out) perm_out[o[k]] = (int64_t)e[k];
#endif
    }
  }
}

KeyLayout layout_for(int64_t E, int64_t M, int64_t N) {
  KeyLayout L;
  L.col_bits = bits_for(N > 0 ? N : 1);
  L.key_bits = bits_for(M > 0 ? M : 1) + L.col_bits;
  L.idx_bits = bits_for(E > 0 ? E : 1);
  L.packed = L.key_bits + L.idx_bits <= 64;
  L.passes = (L.key_bits + kRadixBits - 1) / kRadixBits;
  return L;
}

// bucket-sort launch shapes: (threads, entries per thread); the capacity is their product
#ifndef TSAMD_BK_SORT_THREADS
#define TSAMD_BK_SORT_THREADS 512
#endif
#ifndef TSAMD_BK_SORT_ITEMS
#define TSAMD_BK_SORT_ITEMS 16      // 8192 entries: 64 KB of words + counters -> two workgroups per CU
#endif
#ifndef TSAMD_BK_SORT_ITEMS_VAL
#define TSAMD_BK_SORT_ITEMS_VAL 12  // 6144 entries of 12 bytes
#endif
#ifndef TSAMD_BK_FILL_PCT
#define TSAMD_BK_FILL_PCT 80        // planned mean fill of a bucket, per cent of the capacity
#endif

BucketPlan plan_buckets(int64_t E, int64_t M, int64_t N, const KeyLayout &L, bool val) {
  BucketPlan B{0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (E < TSAMD_BK_MIN_ENTRIES) return B;
  B.cap = TSAMD_BK_SORT_THREADS * (val ? TSAMD_BK_SORT_ITEMS_VAL : TSAMD_BK_SORT_ITEMS);
  const unsigned long long maxkey = (((unsigned long long)(M > 0 ? M - 1 : 0)) << L.col_bits) | (unsigned long long)(N > 0 ? N - 1 : 0);
  auto fill_ok = [&](long double frac, int bits) {
    const long double populated = frac * (long double)(1 << bits);
    const long double fill = (long double)E / (populated < 1.0L ? 1.0L : populated);
    return fill * 100.0L <= (long double)B.cap * TSAMD_BK_FILL_PCT;
  };
  // full words, one level: the bucket id is the top of the word space [0, 2^total) that the sizes can populate
  const int total = L.key_bits + L.idx_bits;
  if (L.packed && total >= 2) {
    const long double maxword = (long double)((maxkey << L.idx_bits) | (unsigned long long)(E - 1)) + 1.0L;
    const long double frac = maxword / ((long double)(1ull << (total - 1)) * 2.0L);
    for (int bits = 1; bits <= kBkMaxBits && bits <= total; ++bits) {
      if (fill_ok(frac, bits)) {
        B.on = 1;
        B.levels = 1;
        B.strip = 0;
        B.bits = B.bits1 = bits;
        B.nb = 1 << bits;
        B.shift = total - bits;
        return B;
      }
    }
  }
  // strip mode: the bucket id is the top of the KEY space; one or two levels
  if (L.key_bits < 1 || L.key_bits + kBkTileBits > 64) return B;
  const long double kfrac = ((long double)maxkey + 1.0L) / ((long double)(1ull << (L.key_bits - 1)) * 2.0L);
  for (int bits = 1; bits <= kBkMaxTotalBits && bits <= L.key_bits; ++bits) {
    if (!fill_ok(kfrac, bits)) continue;
    const int levels = bits <= kBkMaxBits ? 1 : 2;
    const int bits1 = levels == 1 ? bits : (bits + 1) / 2;
    if (L.key_bits - bits1 + L.idx_bits > 64) return B;  // (more bits would only help with a third level)
    B.on = 1;
    B.levels = levels;
    B.strip = 1;
    B.bits = bits;
    B.bits1 = bits1;
    B.nb = 1 << bits;
    B.kshift = L.key_bits - bits;
    B.shift = L.key_bits - bits + L.idx_bits;
    return B;
  }
  return B;
}

// LSD passes over the bits [0, hi) of the word: the top kBkTopBits in 8-bit digits (n_top passes, from `lo` up), the
// bits below in 8-bit digits from 0 (only run when the finish step meets a long group)
#ifndef TSAMD_BK_TOP_BITS
#define TSAMD_BK_TOP_BITS 16
#endif
SortBits sort_bits_for(int hi) {
  SortBits sb;
  sb.n = 0;
  sb.lo = hi > TSAMD_BK_TOP_BITS ? hi - TSAMD_BK_TOP_BITS : 0;
  for (int lo = 0; lo < sb.lo; lo += kRadixBits) {
    sb.shift[sb.n] = (unsigned char)lo;
    sb.width[sb.n] = (unsigned char)(sb.lo - lo < kRadixBits ? sb.lo - lo : kRadixBits);
    ++sb.n;
  }
  sb.n_top = 0;
  for (int lo = sb.lo; lo < hi; lo += kRadixBits) {
    sb.shift[sb.n] = (unsigned char)lo;
    sb.width[sb.n] = (unsigned char)(hi - lo < kRadixBits ? hi - lo : kRadixBits);
    ++sb.n;
    ++sb.n_top;
  }
  return sb;
}

struct SortWs {
  unsigned long long *hdr, *hist, *tile_state, *a, *b;
  unsigned int *ia, *ib;
  unsigned int *bhist, *boff, *cursor, *cursor1;
  unsigned long long *wgstat;  // [kBuildMaxWgs][4]: the probe's per-workgroup results (with a bucket plan)
  size_t zero_bytes;  // hdr + hist + tile_state (+ the used part of bhist) are contiguous: one memset
};

size_t carve_sort(void *base, int64_t E, SortWs *ws) {
  const size_t n = (size_t)(E > 0 ? E : 1);
  const size_t ntiles = (n + kMinTile - 1) / kMinTile;
  char *p = reinterpret_cast<char *>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) -> void * {
    void *r = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return r;
  };
  SortWs w;
  w.hdr = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * kHdrWords));
  w.hist = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * kMaxPasses * kRadix));
  w.tile_state = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * ntiles * kRadix));
  w.zero_bytes = off;  // (+ the part of bhist a plan uses: sort_coo_onesweep)
  w.bhist = reinterpret_cast<unsigned int *>(take(sizeof(unsigned int) * kBkMaxHist * kBkHistCopies));
  w.boff = reinterpret_cast<unsigned int *>(take(sizeof(unsigned int) * (kBkMaxHist + 1)));
  w.cursor = reinterpret_cast<unsigned int *>(take(sizeof(unsigned int) * kBkMaxHist));
  w.cursor1 = reinterpret_cast<unsigned int *>(take(sizeof(unsigned int) * kBkMaxBuckets));
  w.wgstat = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * 4 * kBuildMaxWgs));
  w.a = reinterpret_cast<unsigned long long *>(take(12 * n));  // built words / keys; level 2's output (12-byte records with values)
  w.b = reinterpret_cast<unsigned long long *>(take(12 * n));  // words, or the 12-byte records of the bucket path
  w.ia = reinterpret_cast<unsigned int *>(take(sizeof(unsigned int) * n));
  w.ib = reinterpret_cast<unsigned int *>(take(sizeof(unsigned int) * n));
  if (ws) *ws = w;
  return off;
}

// 0 = returning LDS atomics are a stable rank (checked on the device), 1 = ballot matching; -1 = not decided yet
std::atomic<int> g_rank_mode{-1};
std::mutex g_rank_mutex;

}  // namespace

// The ranking the radix kernels use.  The first call runs `sort_selftest_kernel` (one synchronising round trip of
// a few words; callers that must not synchronise -- stream capture -- call tsamd_sort_selftest() beforehand, the
// Python package does at import).
int sort_rank_mode(hipStream_t) {
  int m = g_rank_mode.load(std::memory_order_acquire);
  if (m >= 0) return m;
  std::lock_guard<std::mutex> lock(g_rank_mutex);
  m = g_rank_mode.load(std::memory_order_acquire);
  if (m >= 0) return m;
  unsigned int *flag = nullptr, host = 1;
  m = 1;  // anything going wrong below leaves the order-independent ranking
  if (hipMalloc(reinterpret_cast<void **>(&flag), sizeof(unsigned int)) == hipSuccess) {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
      if (hipMemsetAsync(flag, 0, sizeof(unsigned int), st) == hipSuccess) {
        hipLaunchKernelGGL(sort_selftest_kernel, dim3(512), dim3(64), 0, st, flag);
        if (hipGetLastError() == hipSuccess &&
            hipMemcpyAsync(&host, flag, sizeof(unsigned int), hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess)
          m = host == 0 ? 0 : 1;
      }
      (void)hipStreamDestroy(st);
    }
    (void)hipFree(flag);
  }
  g_rank_mode.store(m, std::memory_order_release);
  return m;
}

void sort_set_rank_mode(int mode) { g_rank_mode.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_release); }

size_t sort_coo_workspace_bytes(int64_t E) { return carve_sort(nullptr, E, nullptr); }

const unsigned long long *sort_fast_flag(void *workspace, int64_t E) {
  SortWs ws;
  carve_sort(workspace, E, &ws);
  return ws.hdr + kHdrFast;
}

bool sort_coo_supported(int64_t E, int64_t M, int64_t N) {
  return E < ((int64_t)1 << 32) && bits_for(M > 0 ? M : 1) + bits_for(N > 0 ? N : 1) <= 64 &&
         (bits_for(M > 0 ? M : 1) + bits_for(N > 0 ? N : 1) + kRadixBits - 1) / kRadixBits <= kMaxPasses;
}

int sort_coo_onesweep(const int64_t *row, const int64_t *col, int64_t E, int64_t M, int64_t N, int64_t *row_out,
                      int64_t *col_out, int64_t *perm_out, const int64_t *todo, bool probe, int64_t *counts_out,
                      void *workspace, hipStream_t stream, const void *gather_src, void *gather_dst,
                      int gather_bytes, bool check4, const SortCoalesce *co) {
  if (E <= 0) return TSAMD_OK;
  if (!sort_coo_supported(E, M, N)) return TSAMD_ERR_UNSUPPORTED;
  const KeyLayout L = layout_for(E, M, N);
  const unsigned int eblocks = (unsigned int)ceil_div(E, 256);
  if (gather_dst != nullptr && (gather_src == nullptr || (gather_bytes != 4 && gather_bytes != 8))) return TSAMD_ERR_INVALID;
  SortWs ws;
  carve_sort(workspace, E, &ws);
  const size_t pre_zero = co != nullptr ? co->pre_zero_bytes : 0;
  char *const zero_from = reinterpret_cast<char *>(ws.hdr) - pre_zero;
  if (L.passes == 0 || E == 1) {  // nothing to order (a 1 x 1 matrix: every key is equal)
    TSAMD_HIP_TRY(hipMemsetAsync(zero_from, 0, pre_zero + sizeof(unsigned long long) * kHdrWords, stream));  // (kHdrFast = 0 for a caller that asks)
    if (probe && counts_out != nullptr) {
      const int64_t c[2] = {0, E - 1};  // no descent, every adjacent pair a duplicate
      TSAMD_HIP_TRY(hipMemcpyAsync(counts_out, c, sizeof(c), hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(sort_identity_kernel, dim3(eblocks), dim3(256), 0, stream, row, col, E, row_out, col_out,
                       perm_out);
    TSAMD_LAUNCH_CHECK();
    if (gather_dst != nullptr)
      TSAMD_HIP_TRY(hipMemcpyAsync(gather_dst, gather_src, (size_t)E * gather_bytes, hipMemcpyDeviceToDevice, stream));
    return TSAMD_OK;
  }
  // a 4-byte value array rides through the passes of a packed sort / through the bucket kernels
  const bool want4 = gather_dst != nullptr && gather_bytes == 4;
  const bool ride = L.packed && want4 && L.passes >= 2;
  const int64_t ntiles = ceil_div(E, kSortThreads * (L.packed ? (ride ? kItemsOf<true, true> : kItemsOf<true>) : kItemsOf<false>));
  BucketPlan B = plan_buckets(E, M, N, L, want4);
  // a compacting sort needs buckets that end where keys end (the bucket id inside the key bits)
  if (co != nullptr && B.on && !B.strip && B.shift < L.idx_bits) B.on = 0;
  const bool ballot = sort_rank_mode(stream) == 1;
  TSAMD_HIP_TRY(hipMemsetAsync(zero_from, 0, pre_zero + ws.zero_bytes + (B.on ? sizeof(unsigned int) * (size_t)B.nb * kBkHistCopies : 0), stream));
  int build_wgs = 1;
  {
    const int64_t nb = ceil_div(E, kBuildThreads * 4);
    build_wgs = (int)(nb < kBuildMaxWgs ? nb : kBuildMaxWgs);
    hipLaunchKernelGGL(sort_build_kernel, dim3((unsigned int)build_wgs), dim3(kBuildThreads), 0, stream,
                       row, col, E, L, ws.a, ws.hist, ws.hdr, todo, probe ? (check4 ? 2 : 1) : 0, B, ws.bhist, ws.wgstat);
    TSAMD_LAUNCH_CHECK();
  }
  if (B.on) {
    hipLaunchKernelGGL(bucket_plan_kernel, dim3(1), dim3(kBuildThreads), 0, stream, ws.bhist, ws.boff, ws.cursor,
                       ws.cursor1, ws.hdr, todo, probe ? (check4 ? 2 : 1) : 0, B, E, ws.wgstat, build_wgs);
    TSAMD_LAUNCH_CHECK();
    // bucket path: scatter into ws.b (two levels: on into ws.a), sort every bucket in LDS, write the outputs.  These
    // kernels return at once unless the plan kernel raised hdr[kHdrFast]; the passes below return at once when it did (~5 us
    // per kernel that returns at once.  Measured and rejected, profiles/r06_sort_fallback_chain.md: that chain on a forked
    // side stream, normal or high priority -- SLOWER, the idle kernels queue behind the busy ones and the join waits for
    // them; the whole chain as ONE launch of persistent workgroups with grid barriers -- one idle kernel instead of
    // five (sort 0.195 -> 0.179 ms at 7.5 M entries) but 1.5 instead of 1.0 ms on the 21 M-entry R-MAT input that needs it).
    const SortBits SB = sort_bits_for(B.shift);
    const unsigned int *vin = reinterpret_cast<const unsigned int *>(gather_src);
    unsigned int *cur1 = B.levels == 2 ? ws.cursor1 : ws.cursor;
#define TSAMD_BK_SCATTER(V, LV, ST, SRC, DST, CUR)                                                                            \
  hipLaunchKernelGGL((bucket_scatter_kernel<V, LV, ST>), dim3((unsigned int)ceil_div(E, kBkScatterThreads * kBkScatterItems<V>)), \
                     dim3(kBkScatterThreads), 0, stream, SRC, vin, E, B, L, ws.boff, CUR, DST, ws.hdr)
    if (B.strip) {
      if (want4) TSAMD_BK_SCATTER(true, 1, true, ws.a, ws.b, cur1);
      else TSAMD_BK_SCATTER(false, 1, true, ws.a, ws.b, cur1);
    } else {
      if (want4) TSAMD_BK_SCATTER(true, 1, false, ws.a, ws.b, cur1);
      else TSAMD_BK_SCATTER(false, 1, false, ws.a, ws.b, cur1);
    }
    TSAMD_LAUNCH_CHECK();
    const unsigned long long *sorted_in = ws.b;
    if (B.levels == 2) {
      if (want4) TSAMD_BK_SCATTER(true, 2, true, ws.b, ws.a, ws.cursor);
      else TSAMD_BK_SCATTER(false, 2, true, ws.b, ws.a, ws.cursor);
      TSAMD_LAUNCH_CHECK();
      sorted_in = ws.a;
    }
#undef TSAMD_BK_SCATTER
    CoalesceOut Co{nullptr, nullptr, nullptr, E, nullptr, nullptr, -1, 1};
    if (co != nullptr) {
      if (pre_zero == 0) TSAMD_HIP_TRY(hipMemsetAsync(co->status, 0, sizeof(unsigned long long) * (size_t)B.nb, stream));
      Co.seg_ptr = co->no_seg ? nullptr : co->seg_ptr;  // (null: nobody will reduce values by these run starts)
      Co.nnz_out = co->nnz_out;
      Co.status = co->status;
      Co.fused_out = co->fused_out;  // (written by the bucket path in any case: 1 = value_u holds the reduced values)
      if (want4 && co->reduce >= 0 && co->value_u != nullptr && co->fused_out != nullptr) {
        Co.value_u = co->value_u;
        Co.reduce = co->reduce;
        Co.is_float = co->is_float;
      }
    }
#define TSAMD_BK_SORT(ITEMS, V, BAL)                                                                                     \
  do {                                                                                                                   \
    if (co != nullptr)                                                                                                   \
      hipLaunchKernelGGL((bucket_sort_kernel<TSAMD_BK_SORT_THREADS, ITEMS, V, BAL, true>), dim3((unsigned int)B.nb),     \
                         dim3(TSAMD_BK_SORT_THREADS), 0, stream, sorted_in, (const unsigned int *)nullptr, ws.boff, SB, L, \
                         B, co->row_u, co->col_u, (int64_t *)nullptr, ws.hdr, probe ? counts_out : (int64_t *)nullptr,   \
                         check4 ? 1 : 0, gather_src, gather_dst, gather_bytes, Co);                                      \
    else                                                                                                                 \
      hipLaunchKernelGGL((bucket_sort_kernel<TSAMD_BK_SORT_THREADS, ITEMS, V, BAL>), dim3((unsigned int)B.nb),           \
                         dim3(TSAMD_BK_SORT_THREADS), 0, stream, sorted_in, (const unsigned int *)nullptr, ws.boff, SB, L, \
                         B, row_out, col_out, perm_out, ws.hdr, probe ? counts_out : (int64_t *)nullptr, check4 ? 1 : 0, \
                         gather_src, gather_dst, gather_bytes, Co);                                                      \
  } while (0)
    if (want4) {
      if (ballot) TSAMD_BK_SORT(TSAMD_BK_SORT_ITEMS_VAL, true, true);
      else TSAMD_BK_SORT(TSAMD_BK_SORT_ITEMS_VAL, true, false);
    } else {
      if (ballot) TSAMD_BK_SORT(TSAMD_BK_SORT_ITEMS, false, true);
      else TSAMD_BK_SORT(TSAMD_BK_SORT_ITEMS, false, false);
    }
#undef TSAMD_BK_SORT
    TSAMD_LAUNCH_CHECK();
  }
  // the passes of a probing sort are decided by the probe's own counter
  const int64_t *pass_todo = probe ? reinterpret_cast<const int64_t *>(ws.hdr + kHdrDescents) : todo;
  const unsigned long long *src = ws.a;
  const unsigned int *isrc = nullptr;  // first pass: payload = position
  for (int pass = 0; pass < L.passes; ++pass) {
    const bool last = pass == L.passes - 1;
    unsigned long long *dst = (src == ws.a) ? ws.b : ws.a;
    unsigned int *idst = (isrc == ws.ia) ? ws.ib : ws.ia;
    const int shift = pass * kRadixBits + (L.packed ? L.idx_bits : 0);
#define TSAMD_SORT_PASS(P, LST)                                                                                     \
  if (ballot) TSAMD_SORT_PASS_(P, LST, true); else TSAMD_SORT_PASS_(P, LST, false)
#define TSAMD_SORT_PASS_(P, LST, BAL)                                                                                     \
  hipLaunchKernelGGL((onesweep_pass_kernel<P, LST, false, BAL>), dim3((unsigned int)ntiles), dim3(kSortThreads), 0, stream, src, \
                     isrc, dst, idst, row_out, col_out, perm_out, E, shift, L, ws.hist + pass * kRadix,              \
                     ws.tile_state, ws.hdr, (unsigned int)(pass + 1), pass_todo, row, col,                          \
                     (last && probe) ? counts_out : (int64_t *)nullptr, gather_src, gather_dst, gather_bytes,          \
                     check4 ? 1 : 0)
    if (ride) {
#define TSAMD_SORT_PASS_VAL(LST)                                                                                      \
  if (ballot) TSAMD_SORT_PASS_VAL_(LST, true); else TSAMD_SORT_PASS_VAL_(LST, false)
#define TSAMD_SORT_PASS_VAL_(LST, BAL)                                                                                      \
  hipLaunchKernelGGL((onesweep_pass_kernel<true, LST, true, BAL>), dim3((unsigned int)ntiles), dim3(kSortThreads), 0,      \
                     stream, src, isrc, dst, idst, row_out, col_out, perm_out, E, shift, L, ws.hist + pass * kRadix,  \
                     ws.tile_state, ws.hdr, (unsigned int)(pass + 1), pass_todo, row, col,                           \
                     (last && probe) ? counts_out : (int64_t *)nullptr, gather_src, gather_dst, gather_bytes,           \
                     check4 ? 1 : 0)
      if (last) { TSAMD_SORT_PASS_VAL(true); }
      else { TSAMD_SORT_PASS_VAL(false); }
#undef TSAMD_SORT_PASS_VAL
#undef TSAMD_SORT_PASS_VAL_
    } else if (L.packed) {
      if (last) { TSAMD_SORT_PASS(true, true); }
      else { TSAMD_SORT_PASS(true, false); }
    } else {
      if (last) { TSAMD_SORT_PASS(false, true); }
      else { TSAMD_SORT_PASS(false, false); }
    }
#undef TSAMD_SORT_PASS
#undef TSAMD_SORT_PASS_
    TSAMD_LAUNCH_CHECK();
    src = dst;
    isrc = idst;
  }
  return TSAMD_OK;
}

}  // namespace tsamd
