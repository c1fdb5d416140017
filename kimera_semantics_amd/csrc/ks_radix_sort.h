// ks_radix_sort.h — stable LSD radix sort for gfx950, one kernel per digit pass ("onesweep").
//
// Why not rocPRIM: for the sizes of this path (3e5 points, 5e5..3e6 pairs per frame) rocPRIM's
// radix_sort dispatches to block-sort + merge passes (~30 launches of ~6 us each per frame,
// measured early in round 1); the frame is launch-latency bound there.  This sort is
//   k_rs_hist : one pass over the keys -> digit histograms of ALL passes (LDS atomics, one add
//               per wave when the digit is wave-uniform, as the upper digits of these keys are)
//   k_rs_pass : per pass ONE kernel: 2048..16384-key tiles, wave-ballot multi-split ranking
//               (stable), cross-tile digit prefixes by decoupled look-back on agent-scope
//               atomics (8 predecessor loads in flight), in-kernel scan of the pass histogram,
//               direct scatter.
// Digit width RB is a template parameter; 8 bits is what is used (see sort()).
// Keys are u32 or u64, optional u32 payload.  64-wide wavefronts throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>

namespace ksrs {

constexpr int kMaxPasses = 8;
constexpr int kMaxBins = 2048;
constexpr size_t kHistBlocks = 2048;   // workgroups of the histogram kernel at most
constexpr uint32_t kFlagLocal = 1u << 30, kFlagPrefix = 2u << 30, kCountMask = (1u << 30) - 1u;

template <int RB, typename K>
__device__ __forceinline__ uint32_t digit_of(K key, int shift) {
  return (uint32_t)(key >> shift) & (uint32_t)((1 << RB) - 1);
}

// lanes of this wave (among `active`) that hold the same RB-bit digit: multi-split by ballots
template <int RB>
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, unsigned long long active) {
  unsigned long long peers = active;
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    peers &= bit ? bal : ~bal;
  }
  return peers;
}

// hist[pass][bin] += digit counts of a 2048-key slice, for every pass at once.
// It also clears the workspace half the PREVIOUS sort used (zero_ptr, zero_words: a multiple of 4),
// which becomes the next sort's workspace: no memset launch per sort.
template <typename K, int RB>
__device__ __forceinline__ void rs_hist_body(const K* __restrict__ keys, uint32_t n, int passes, int begin_bit,
                                             uint32_t* __restrict__ hist, uint32_t* __restrict__ zero_ptr,
                                             uint32_t zero_words, const unsigned long long* __restrict__ n_dev, uint32_t* s_hist) {
  constexpr int kBins = 1 << RB;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (n_dev) n = (uint32_t)(*n_dev < (unsigned long long)n ? *n_dev : (unsigned long long)n);   // sort_dev: the key count lives on the device (n = capacity)
  for (uint32_t i = blockIdx.x * 256u + tid; i < zero_words / 4u; i += gridDim.x * 256u)
    ((uint4*)zero_ptr)[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < passes * kBins; i += 256) s_hist[i] = 0;
  __syncthreads();
  // a workgroup walks several 2048-key slices (grid <= kHistBlocks): the flush below is up to passes x 256
  // same-address global atomics per workgroup, which at 2e4 workgroups cost more than the pass itself
  const uint32_t n_slices = (n + 2047u) / 2048u;
  for (uint32_t slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
    const uint32_t base = slice * 2048u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t idx = base + i * 256 + tid;
      const bool valid = idx < n;
      const K key = valid ? keys[idx] : (K)0;
      const unsigned long long active = __ballot(valid);
      for (int p = 0; p < passes; ++p) {
        const uint32_t d = digit_of<RB>(key, begin_bit + p * RB);
        // the upper digits of these keys are nearly constant: one add for a wave-uniform digit,
        // per-lane LDS atomics otherwise
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        if (__ballot(valid && d != d0) == 0ull) {
          if (active && lane == (uint32_t)(__ffsll((long long)active) - 1))
            atomicAdd(&s_hist[p * kBins + d0], (uint32_t)__popcll(active));
        } else {
          // neighbouring keys mostly share their upper digits (a ray's consecutive voxels, a voxel's consecutive updates):
          // ONE add per run of equal digits in lane order — per-lane adds to a handful of counters serialise in LDS
          // (SQ counters, round 6: 0.92 bank-conflict cycles per LDS cycle, the kernel 2x its streaming time)
          const uint32_t dp = __shfl_up(d, 1);
          const unsigned long long lead = __ballot(valid && (lane == 0u || d != dp));
          if (valid && ((lead >> lane) & 1ull)) {
            const unsigned long long rest = (lane < 63u) ? (lead >> (lane + 1u)) : 0ull;
            const uint32_t stop = rest ? lane + (uint32_t)__ffsll((long long)rest) : 64u;      // first lane of the next run
            const unsigned long long in_run = (stop < 64u ? ((1ull << stop) - 1ull) : ~0ull) & ~((1ull << lane) - 1ull);
            atomicAdd(&s_hist[p * kBins + d], (uint32_t)__popcll(in_run & active));
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < passes * kBins; i += 256) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&hist[(i / kBins) * kMaxBins + (i % kBins)], c);
  }
}
template <typename K, int RB>
__global__ void __launch_bounds__(256) k_rs_hist(const K* __restrict__ keys, uint32_t n, int passes, int begin_bit,
                                                 uint32_t* __restrict__ hist, uint32_t* __restrict__ zero_ptr,
                                                 uint32_t zero_words, const unsigned long long* __restrict__ n_dev = nullptr) {
  extern __shared__ uint32_t s_hist[];  // [passes][kBins]
  rs_hist_body<K, RB>(keys, n, passes, begin_bit, hist, zero_ptr, zero_words, n_dev, s_hist);
}
// Up to kRsBatch independent sorts of the same capacity per launch (blockIdx.y = sort): their launches are latency bound,
// and a launch costs the same for one sort or four.
constexpr int kRsBatch = 8;
template <typename K>
struct DevBatch {
  K* keys_a[kRsBatch];
  K* keys_b[kRsBatch];
  uint32_t* vals_a[kRsBatch];
  uint32_t* vals_b[kRsBatch];
  const unsigned long long* n_dev[kRsBatch];
  uint32_t* ws[kRsBatch];
};
template <typename K, int RB>
__global__ void __launch_bounds__(256) k_rs_hist_b(DevBatch<K> B, uint32_t cap, int passes, int begin_bit) {
  extern __shared__ uint32_t s_hist[];
  rs_hist_body<K, RB>(B.keys_a[blockIdx.y], cap, passes, begin_bit, B.ws[blockIdx.y], nullptr, 0u, B.n_dev[blockIdx.y], s_hist);
}

// One pass.  THREADS x ITEMS keys per tile.  REORDER (keys only, large inputs): the tile's keys are first put in
// digit order in LDS and leave from there, so that neighbouring lanes write neighbouring addresses (runs of
// ~tile/256 keys per digit) instead of 8-byte scatters — the direct scatter wrote 2.6x its bytes at 3e7 keys.
template <typename K, bool HAS_VALUES, int THREADS, int ITEMS, int RB, bool REORDER = false>
__device__ __forceinline__ void rs_pass_body(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                             const uint32_t* __restrict__ vals_in,
                                             uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                             const uint32_t* __restrict__ bin_hist,
                                             uint32_t* __restrict__ status, uint32_t* __restrict__ ticket,
                                             const unsigned long long* __restrict__ n_dev) {
  constexpr int kBins = 1 << RB;
  constexpr int kChunks = kBins / 64;
  constexpr int kWaves = THREADS / 64;
  constexpr int kTile = THREADS * ITEMS;
  if (n_dev) n = (uint32_t)(*n_dev < (unsigned long long)n ? *n_dev : (unsigned long long)n);   // sort_dev (the grid covers the capacity)
  __shared__ uint32_t s_cnt[kWaves][kBins];  // per-wave digit counters, later exclusive wave bases
  __shared__ uint32_t s_off[kBins];          // global offset of this tile's first key of each digit
  __shared__ uint32_t s_chunk[kChunks];      // sums of 64-bin chunks of the pass histogram
  __shared__ uint32_t s_tile;
  __shared__ K s_keys[REORDER ? kTile : 1];
  __shared__ uint32_t s_loc[REORDER ? kBins : 1];   // tile-local exclusive prefix of the digit totals
  static_assert(!REORDER || (!HAS_VALUES && kBins <= THREADS), "reorder variant: keys only");
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // Tickets order the tiles by arrival so a predecessor is always resident (forward progress of the
  // look-back does not depend on the dispatch order or on what else shares the CUs); the same-address
  // atomic costs ~1-2 us per pass at these tile counts.
  if (tid == 0) s_tile = ticket ? atomicAdd(ticket, 1u) : blockIdx.x;
  for (int i = tid; i < kWaves * kBins; i += THREADS) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  // (sort_dev: tickets are handed out in arrival order, so the tiles past the last key are exactly those that arrive
  // after every tile with keys has its ticket: nothing ever looks back at them)
  if (n_dev && (unsigned long long)tile * kTile >= (unsigned long long)n) return;
  const uint32_t wbase = tile * kTile + wave * (ITEMS * 64);

  K key[ITEMS];
  uint32_t rd[ITEMS];  // rank within the tile's wave (low 20 bits) | digit (high 12 bits)
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = wbase + i * 64 + lane;
    key[i] = (idx < n) ? keys_in[idx] : (K)0;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = wbase + i * 64 + lane;
    const bool valid = idx < n;
    const uint32_t d = digit_of<RB>(key[i], shift);
    const unsigned long long peers = match_digit<RB>(d, __ballot(valid));
    const uint32_t prev = s_cnt[wave][d];
    rd[i] = (prev + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull))) | (d << 20);
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull) s_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // each thread owns the digits tid, tid + THREADS, ...
  for (int d = tid; d < kBins; d += THREADS) {
    // exclusive bases of digit d across the waves of this tile, tile total
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t c = s_cnt[w][d];
      s_cnt[w][d] = total;
      total += c;
    }
    if (REORDER) s_loc[d] = total;
    // decoupled look-back over the tiles with smaller index; kWindow predecessor loads are
    // kept in flight.  Tile "-1" reads as an inclusive prefix of zero.
    uint32_t* st = status + d;
    uint32_t prefix = 0;
    __hip_atomic_store(st + (size_t)tile * kBins, total | kFlagLocal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int kWindow = 8;
    int t = (int)tile - 1;
    bool done = false;
    while (!done) {
      uint32_t s[kWindow];
#pragma unroll
      for (int j = 0; j < kWindow; ++j)
        s[j] = (t - j >= 0) ? __hip_atomic_load(st + (size_t)(t - j) * kBins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : kFlagPrefix;
      int consumed = 0;
#pragma unroll
      for (int j = 0; j < kWindow; ++j) {
        if (done || consumed != j) continue;
        const uint32_t flag = s[j] >> 30;
        if (flag == 0u) continue;  // not published yet: retry from here
        prefix += s[j] & kCountMask;
        ++consumed;
        if (flag == 2u) done = true;
      }
      t -= consumed;
    }
    __hip_atomic_store(st + (size_t)tile * kBins, (prefix + total) | kFlagPrefix, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    s_off[d] = prefix;
  }
  __syncthreads();
  // exclusive scan of the pass histogram = global base of each digit.  Phase 1: every 64-bin
  // chunk is scanned by one wave (the loop bound is uniform per wave, no barrier inside);
  // phase 2: chunk sums are added.
  for (int c = wave; c < kChunks; c += kWaves) {
    const uint32_t hv = bin_hist[c * 64 + lane];
    uint32_t x = hv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) s_chunk[c] = x;
    s_off[c * 64 + lane] += x - hv;
  }
  __syncthreads();
  for (int d = tid; d < kBins; d += THREADS) {
    uint32_t add = 0;
    for (int cc = 0; cc < d / 64; ++cc) add += s_chunk[cc];
    s_off[d] += add;
  }
  __syncthreads();

  if (REORDER) {
    // exclusive scan of the tile's digit totals (s_loc holds the totals): 64-bin chunks by one wave each
    __shared__ uint32_t s_lchunk[kChunks];
    for (int c = wave; c < kChunks; c += kWaves) {
      const uint32_t tv = s_loc[c * 64 + lane];
      uint32_t x = tv;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if (lane >= (uint32_t)o) x += y;
      }
      if (lane == 63) s_lchunk[c] = x;
      s_loc[c * 64 + lane] = x - tv;
    }
    __syncthreads();
    for (int d = tid; d < kBins; d += THREADS) {
      uint32_t add = 0;
      for (int cc = 0; cc < d / 64; ++cc) add += s_lchunk[cc];
      s_loc[d] += add;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const uint32_t idx = wbase + i * 64 + lane;
      if (idx < n) {
        const uint32_t d = rd[i] >> 20;
        s_keys[s_loc[d] + s_cnt[wave][d] + (rd[i] & 0xfffffu)] = key[i];
      }
    }
    __syncthreads();
    const uint32_t t0 = tile * (uint32_t)kTile;
    const uint32_t count = (n - t0 < (uint32_t)kTile) ? n - t0 : (uint32_t)kTile;
    for (uint32_t j = tid; j < count; j += THREADS) {
      const K k = s_keys[j];
      const uint32_t d = digit_of<RB>(k, shift);
      keys_out[s_off[d] + (j - s_loc[d])] = k;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = wbase + i * 64 + lane;
    if (idx < n) {
      const uint32_t d = rd[i] >> 20;
      const uint32_t pos = s_off[d] + s_cnt[wave][d] + (rd[i] & 0xfffffu);
      keys_out[pos] = key[i];
      if (HAS_VALUES) vals_out[pos] = vals_in[idx];
    }
  }
}

template <typename K, bool HAS_VALUES, int THREADS, int ITEMS, int RB, bool REORDER = false>
__global__ void __launch_bounds__(THREADS) k_rs_pass(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                                     const uint32_t* __restrict__ vals_in,
                                                     uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                     const uint32_t* __restrict__ bin_hist,
                                                     uint32_t* __restrict__ status, uint32_t* __restrict__ ticket,
                                                     const unsigned long long* __restrict__ n_dev = nullptr) {
  rs_pass_body<K, HAS_VALUES, THREADS, ITEMS, RB, REORDER>(keys_in, keys_out, vals_in, vals_out, n, shift, bin_hist, status, ticket, n_dev);
}
// pass p of the batched sorts: odd passes read the b buffers
template <typename K, int THREADS, int ITEMS, int RB>
__global__ void __launch_bounds__(THREADS) k_rs_pass_b(DevBatch<K> B, uint32_t cap, int begin_bit, int p, uint32_t tiles) {
  const uint32_t y = blockIdx.y;
  uint32_t* ws = B.ws[y];
  const bool odd = (p & 1) != 0;
  rs_pass_body<K, true, THREADS, ITEMS, RB, false>(odd ? B.keys_b[y] : B.keys_a[y], odd ? B.keys_a[y] : B.keys_b[y], odd ? B.vals_b[y] : B.vals_a[y],
                                                   odd ? B.vals_a[y] : B.vals_b[y], cap, begin_bit + p * RB, ws + (size_t)p * kMaxBins,
                                                   ws + ((size_t)kMaxPasses * kMaxBins + kMaxPasses) + (size_t)p * tiles * (1 << RB),
                                                   ws + (size_t)kMaxPasses * kMaxBins + p, B.n_dev[y]);
}

// Host-side workspace + launcher.  Sorts bits [begin_bit, end_bit) of the keys; the result is
// in (*keys_result, *vals_result), each pointing at one of the two ping-pong buffers.
// Two halves used alternately: while a sort runs in one half its histogram kernel clears what the
// previous sort left in the other, so a sort never needs a memset of its own.
struct Workspace {
  uint32_t* d_ws = nullptr;  // per half: [kMaxPasses][kMaxBins] histograms | kMaxPasses tickets | status[passes][tiles][bins]
  size_t words = 0;          // capacity of ONE half (multiple of 4)
  int cur = 0;               // half used by the last sort
  size_t dirty[2] = {0, 0};  // words that sort left non-zero in each half
};
constexpr size_t kHeadWords = (size_t)kMaxPasses * kMaxBins + kMaxPasses;

inline hipError_t ensure(Workspace& w, size_t words, hipStream_t stream) {
  if (words <= w.words) return hipSuccess;
  if (w.d_ws) (void)hipFree(w.d_ws);  // waits for the device
  w.d_ws = nullptr;
  w.words = 0;
  const size_t cap = ((words + words / 4) + 3) & ~(size_t)3;
  hipError_t e = hipMalloc((void**)&w.d_ws, 2 * cap * sizeof(uint32_t));
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(w.d_ws, 0, 2 * cap * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  w.words = cap;
  w.cur = 0;
  w.dirty[0] = w.dirty[1] = 0;
  return hipSuccess;
}

inline void release(Workspace& w) {
  if (w.d_ws) (void)hipFree(w.d_ws);
  w = Workspace{};
}

template <typename K, bool HAS_VALUES, int THREADS, int ITEMS, int RB, bool REORDER = false>
inline void launch_passes(Workspace& w, K*& kin, K*& kout, uint32_t*& vin, uint32_t*& vout, size_t n, int passes,
                          uint32_t tiles, unsigned begin_bit, hipStream_t stream) {
  constexpr int kBins = 1 << RB;
  uint32_t* base = w.d_ws + (size_t)w.cur * w.words;
  uint32_t* hist = base;
  uint32_t* tickets = base + (size_t)kMaxPasses * kMaxBins;
  uint32_t* status = base + kHeadWords;
  for (int p = 0; p < passes; ++p) {
    hipLaunchKernelGGL((k_rs_pass<K, HAS_VALUES, THREADS, ITEMS, RB, REORDER>), dim3(tiles), dim3(THREADS), 0, stream, kin, kout,
                       vin, vout, (uint32_t)n, (int)begin_bit + p * RB, hist + (size_t)p * kMaxBins,
                       status + (size_t)p * tiles * kBins, tickets + p);
    K* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
}

template <typename K, bool HAS_VALUES, int RB>
inline hipError_t sort_rb(Workspace& w, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n,
                          unsigned begin_bit, unsigned end_bit, hipStream_t stream, K** keys_result,
                          uint32_t** vals_result) {
  constexpr int kBins = 1 << RB;
  const int passes = (int)((end_bit - begin_bit + RB - 1) / RB);
  // tile size: 2048 keys keeps per-tile latency low for per-frame sizes; larger tiles shorten
  // the look-back chain for million-scale inputs
  // keys-only sorts of more than 2^20 keys (the pair sort of large frames) take the LDS-reordering variant
  const bool reorder = !HAS_VALUES && n > (1u << 20);
  const int tile = (n <= (1u << 20)) ? 2048 : (reorder || n <= (1u << 24)) ? 8192 : 16384;
  const uint32_t tiles = (uint32_t)((n + tile - 1) / tile);
  const size_t words = kHeadWords + (size_t)passes * tiles * kBins;
  hipError_t e = ensure(w, words, stream);
  if (e != hipSuccess) return e;
  const int h = w.cur ^ 1;  // this sort's half is clean; the histogram kernel clears the other one
  hipLaunchKernelGGL((k_rs_hist<K, RB>), dim3((uint32_t)std::min<size_t>((n + 2047) / 2048, kHistBlocks)), dim3(256),
                     (size_t)passes * kBins * sizeof(uint32_t), stream, keys_a, (uint32_t)n, passes, (int)begin_bit,
                     w.d_ws + (size_t)h * w.words, w.d_ws + (size_t)(h ^ 1) * w.words,
                     (uint32_t)((w.dirty[h ^ 1] + 3) & ~(size_t)3));
  w.dirty[h ^ 1] = 0;
  w.dirty[h] = words;
  w.cur = h;
  K* kin = keys_a;
  K* kout = keys_b;
  uint32_t* vin = vals_a;
  uint32_t* vout = vals_b;
  if (tile == 2048) launch_passes<K, HAS_VALUES, 256, 8, RB>(w, kin, kout, vin, vout, n, passes, tiles, begin_bit, stream);
  else if (tile == 8192 && reorder) {
    if constexpr (!HAS_VALUES) launch_passes<K, false, 512, 16, RB, true>(w, kin, kout, vin, vout, n, passes, tiles, begin_bit, stream);
  } else if (tile == 8192) launch_passes<K, HAS_VALUES, 512, 16, RB>(w, kin, kout, vin, vout, n, passes, tiles, begin_bit, stream);
  else launch_passes<K, HAS_VALUES, 512, 32, RB>(w, kin, kout, vin, vout, n, passes, tiles, begin_bit, stream);
  *keys_result = kin;
  if (vals_result) *vals_result = vin;
  return hipGetLastError();
}

// The same sort with the key count in DEVICE memory and nothing on the host that changes from call to call: the launch
// sequence depends only on `cap` (capacity of the buffers) and can be captured into a graph and replayed.  `ws` holds
// ws_words_dev(cap, passes) words and is cleared by a memset node at the head of the sequence.  Keys + values, tiles of
// 2048 (cap <= 2^20) / 8192 / 16384 keys.  The result is in (keys_b, vals_b) after an odd number of passes, else in (a).
inline size_t dev_tile(size_t cap) { return cap <= (1u << 20) ? 2048 : cap <= (1u << 24) ? 8192 : 16384; }
inline size_t ws_words_dev(size_t cap, int passes) { return kHeadWords + (size_t)passes * ((cap + dev_tile(cap) - 1) / dev_tile(cap)) * 256; }
template <typename K>
inline hipError_t sort_dev(uint32_t* ws, size_t ws_words, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                           const unsigned long long* n_dev, size_t cap, unsigned begin_bit, unsigned end_bit, hipStream_t stream,
                           K** keys_result, uint32_t** vals_result) {
  constexpr int RB = 8;
  constexpr int kBins = 1 << RB;
  const int passes = (int)((end_bit - begin_bit + RB - 1) / RB);
  const size_t tile = dev_tile(cap);
  const uint32_t tiles = (uint32_t)((cap + tile - 1) / tile);
  if (kHeadWords + (size_t)passes * tiles * kBins > ws_words) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(ws, 0, (kHeadWords + (size_t)passes * tiles * kBins) * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  uint32_t* hist = ws;
  uint32_t* tickets = ws + (size_t)kMaxPasses * kMaxBins;
  uint32_t* status = ws + kHeadWords;
  hipLaunchKernelGGL((k_rs_hist<K, RB>), dim3((uint32_t)std::min<size_t>((cap + 2047) / 2048, kHistBlocks)), dim3(256),
                     (size_t)passes * kBins * sizeof(uint32_t), stream, (const K*)keys_a, (uint32_t)cap, passes, (int)begin_bit, hist,
                     (uint32_t*)nullptr, 0u, n_dev);
  K* kin = keys_a;
  K* kout = keys_b;
  uint32_t* vin = vals_a;
  uint32_t* vout = vals_b;
  for (int p = 0; p < passes; ++p) {
#define KS_RS_DEV_PASS(THREADS, ITEMS)                                                                                                 \
  hipLaunchKernelGGL((k_rs_pass<K, true, THREADS, ITEMS, RB>), dim3(tiles), dim3(THREADS), 0, stream, (const K*)kin, kout,             \
                     (const uint32_t*)vin, vout, (uint32_t)cap, (int)begin_bit + p * RB, (const uint32_t*)(hist + (size_t)p * kMaxBins), \
                     status + (size_t)p * tiles * kBins, tickets + p, n_dev)
    if (tile == 2048) KS_RS_DEV_PASS(256, 8);
    else if (tile == 8192) KS_RS_DEV_PASS(512, 16);
    else KS_RS_DEV_PASS(512, 32);
#undef KS_RS_DEV_PASS
    K* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  *keys_result = kin;
  *vals_result = vin;
  return hipGetLastError();
}

// nb sorts of the same capacity in one launch sequence (memset per workspace, then hist + passes with blockIdx.y = sort).
// Results as sort_dev: in the b buffers after an odd number of passes.
template <typename K>
inline hipError_t sort_dev_batch(const DevBatch<K>& B, int nb, size_t ws_words, size_t cap, unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
  constexpr int RB = 8;
  constexpr int kBins = 1 << RB;
  const int passes = (int)((end_bit - begin_bit + RB - 1) / RB);
  const size_t tile = dev_tile(cap);
  const uint32_t tiles = (uint32_t)((cap + tile - 1) / tile);
  if (kHeadWords + (size_t)passes * tiles * kBins > ws_words || nb < 1 || nb > kRsBatch) return hipErrorInvalidValue;
  for (int y = 0; y < nb; ++y) {
    hipError_t e = hipMemsetAsync(B.ws[y], 0, (kHeadWords + (size_t)passes * tiles * kBins) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_rs_hist_b<K, RB>), dim3((uint32_t)std::min<size_t>((cap + 2047) / 2048, kHistBlocks), (uint32_t)nb), dim3(256),
                     (size_t)passes * kBins * sizeof(uint32_t), stream, B, (uint32_t)cap, passes, (int)begin_bit);
  for (int p = 0; p < passes; ++p) {
    if (tile == 2048) hipLaunchKernelGGL((k_rs_pass_b<K, 256, 8, RB>), dim3(tiles, (uint32_t)nb), dim3(256), 0, stream, B, (uint32_t)cap, (int)begin_bit, p, tiles);
    else if (tile == 8192) hipLaunchKernelGGL((k_rs_pass_b<K, 512, 16, RB>), dim3(tiles, (uint32_t)nb), dim3(512), 0, stream, B, (uint32_t)cap, (int)begin_bit, p, tiles);
    else hipLaunchKernelGGL((k_rs_pass_b<K, 512, 32, RB>), dim3(tiles, (uint32_t)nb), dim3(512), 0, stream, B, (uint32_t)cap, (int)begin_bit, p, tiles);
  }
  return hipGetLastError();
}

template <typename K, bool HAS_VALUES>
inline hipError_t sort(Workspace& w, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n,
                       unsigned end_bit, hipStream_t stream, K** keys_result, uint32_t** vals_result,
                       unsigned begin_bit = 0) {
  *keys_result = keys_a;
  if (vals_result) *vals_result = vals_a;
  if (end_bit > sizeof(K) * 8) end_bit = (unsigned)(sizeof(K) * 8);
  if (n == 0 || end_bit <= begin_bit) return hipSuccess;
  // 8-bit digits.  (11-bit digits were tried to cut 21-bit sorts from 3 passes to 2: each
  // thread then walks 8 look-back chains one after the other and a pass got ~4x slower —
  // measured 0.103 vs 0.056 ms for the dedup sort — so the wider digit is not used.)
  return sort_rb<K, HAS_VALUES, 8>(w, keys_a, keys_b, vals_a, vals_b, n, begin_bit, end_bit, stream, keys_result, vals_result);
}

}  // namespace ksrs
