// ks_radix_sort.h — stable LSD radix sort for gfx950, one kernel per 8-bit pass ("onesweep").
//
// Why not rocPRIM: for the sizes of this path (3e5 points, 5e5..3e6 pairs per frame) rocPRIM's
// radix_sort dispatches to block-sort + merge passes (~30 launches of ~6 us each per frame,
// profiles/r01_fast_kernel_stats.txt); the frame is launch-latency bound there.  This sort is
//   k_rs_hist : one pass over the keys -> digit histograms of ALL passes (wave-aggregated LDS atomics)
//   k_rs_pass : per pass ONE kernel: 2048..16384-key tiles, wave-ballot multi-split ranking
//               (stable), cross-tile digit prefixes by decoupled look-back on agent-scope
//               atomics (tiles take tickets so a predecessor is always resident; 8 loads in
//               flight), in-kernel scan of the pass histogram, direct scatter.
// Keys are u32 or u64, optional u32 payload.  64-wide wavefronts throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ksrs {

constexpr int kRadixBits = 8;
constexpr int kBins = 1 << kRadixBits;
constexpr int kMaxPasses = 8;
constexpr uint32_t kFlagLocal = 1u << 30, kFlagPrefix = 2u << 30, kCountMask = (1u << 30) - 1u;

template <typename K>
__device__ __forceinline__ uint32_t digit_of(K key, int shift) {
  return (uint32_t)(key >> shift) & (uint32_t)(kBins - 1);
}

// lanes of this wave (among `active`) that hold the same 8-bit digit: multi-split by ballots
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, unsigned long long active) {
  unsigned long long peers = active;
#pragma unroll
  for (int b = 0; b < kRadixBits; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    peers &= bit ? bal : ~bal;
  }
  return peers;
}

// hist[pass][bin] += digit counts of a 2048-key slice, for every pass at once.  Counting is
// wave-aggregated (one LDS atomic per distinct digit per wave instruction): the upper digits
// of these keys are almost constant, a per-lane atomic would serialise 2048-fold.
template <typename K>
__global__ void __launch_bounds__(256) k_rs_hist(const K* __restrict__ keys, uint32_t n, int passes, int begin_bit,
                                                 uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_hist[kMaxPasses][kBins];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  for (int p = 0; p < passes; ++p) s_hist[p][tid] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * 2048u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t idx = base + i * 256 + tid;
    const bool valid = idx < n;
    const K key = valid ? keys[idx] : (K)0;
    const unsigned long long active = __ballot(valid);
    for (int p = 0; p < passes; ++p) {
      const uint32_t d = digit_of(key, begin_bit + p * kRadixBits);
      // the upper digits of these keys are nearly constant: one add for a wave-uniform digit,
      // per-lane LDS atomics otherwise
      const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
      if (__ballot(valid && d != d0) == 0ull) {
        if (lane == (uint32_t)(__ffsll((long long)active) - 1) && active) atomicAdd(&s_hist[p][d0], (uint32_t)__popcll(active));
      } else if (valid) {
        atomicAdd(&s_hist[p][d], 1u);
      }
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    const uint32_t c = s_hist[p][tid];
    if (c) atomicAdd(&hist[p * kBins + tid], c);
  }
}

// One pass.  THREADS x ITEMS keys per tile; large tiles keep the number of co-resident tiles
// (and with it the depth of the look-back chain, ~1.5 us per dependent L2 round trip) small.
template <typename K, bool HAS_VALUES, int THREADS, int ITEMS>
__global__ void __launch_bounds__(THREADS) k_rs_pass(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                                     const uint32_t* __restrict__ vals_in,
                                                     uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                     const uint32_t* __restrict__ bin_hist,
                                                     uint32_t* __restrict__ status, uint32_t* __restrict__ ticket) {
  constexpr int kWaves = THREADS / 64;
  constexpr int kTile = THREADS * ITEMS;
  constexpr int kDigitsPerThread = kBins / THREADS > 0 ? kBins / THREADS : 1;
  static_assert(THREADS >= kBins, "one thread per digit for the look-back");
  __shared__ uint32_t s_cnt[kWaves][kBins];   // per-wave digit counters, later exclusive wave bases
  __shared__ uint32_t s_off[kBins];           // global offset of this tile's first key of each digit
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_wsum[kBins / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // Tickets order the tiles by arrival so a predecessor is always resident.  When the whole
  // grid is co-resident anyway (ticket == nullptr) the block index is used directly and the
  // same-address atomic (serialised at ~88/us) is avoided.
  if (tid == 0) s_tile = ticket ? atomicAdd(ticket, 1u) : blockIdx.x;
  for (int w = 0; w < kWaves; ++w)
    if (tid < kBins) s_cnt[w][tid] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t wbase = tile * kTile + wave * (ITEMS * 64);

  K key[ITEMS];
  uint32_t rd[ITEMS];  // rank within the tile's wave (low 24 bits) | digit (high 8 bits)
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = wbase + i * 64 + lane;
    key[i] = (idx < n) ? keys_in[idx] : (K)0;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = wbase + i * 64 + lane;
    const bool valid = idx < n;
    const uint32_t d = digit_of(key[i], shift);
    const unsigned long long peers = match_digit(d, __ballot(valid));
    const uint32_t prev = s_cnt[wave][d];
    rd[i] = (prev + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull))) | (d << 24);
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull) s_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  uint32_t total = 0, prefix = 0;
  if (tid < kBins) {
    // thread d: exclusive bases of digit d across the waves of this tile, tile total
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t c = s_cnt[w][tid];
      s_cnt[w][tid] = total;
      total += c;
    }
    // decoupled look-back over the tiles that took earlier tickets; kWindow predecessor loads
    // are kept in flight.  Tile "-1" reads as an inclusive prefix of zero.
    uint32_t* st = status + tid;
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
  }
  // exclusive scan of the pass histogram (256 bins) gives the global base of each digit
  uint32_t hv = 0, x = 0;
  if (tid < kBins) {
    hv = bin_hist[tid];
    x = hv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) s_wsum[wave] = x;
  }
  __syncthreads();
  if (tid < kBins) {
    uint32_t add = 0;
    for (uint32_t w = 0; w < wave; ++w) add += s_wsum[w];
    s_off[tid] = add + x - hv + prefix;
  }
  __syncthreads();
  (void)kDigitsPerThread;

#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = wbase + i * 64 + lane;
    if (idx < n) {
      const uint32_t d = rd[i] >> 24;
      const uint32_t pos = s_off[d] + s_cnt[wave][d] + (rd[i] & 0xffffffu);
      keys_out[pos] = key[i];
      if (HAS_VALUES) vals_out[pos] = vals_in[idx];
    }
  }
}

// Host-side workspace + launcher.  Sorts bits [0, end_bit) of the keys; the result is in
// (*keys_result, *vals_result), each pointing at one of the two ping-pong buffers.
struct Workspace {
  uint32_t* d_ws = nullptr;  // [kMaxPasses][256] histograms | kMaxPasses tickets | status[passes][tiles][256]
  size_t words = 0;
};
constexpr size_t kHeadWords = kMaxPasses * kBins + kMaxPasses;

inline hipError_t ensure(Workspace& w, size_t words) {
  if (words <= w.words) return hipSuccess;
  if (w.d_ws) (void)hipFree(w.d_ws);
  w.d_ws = nullptr;
  const size_t cap = words + words / 4;
  hipError_t e = hipMalloc((void**)&w.d_ws, cap * sizeof(uint32_t));
  if (e != hipSuccess) return e;
  w.words = cap;
  return hipSuccess;
}

inline void release(Workspace& w) {
  if (w.d_ws) (void)hipFree(w.d_ws);
  w = Workspace{};
}

template <typename K, bool HAS_VALUES, int THREADS, int ITEMS>
inline void launch_passes(Workspace& w, K*& kin, K*& kout, uint32_t*& vin, uint32_t*& vout, size_t n, int passes,
                          uint32_t tiles, unsigned begin_bit, hipStream_t stream) {
  uint32_t* hist = w.d_ws;
  uint32_t* tickets = w.d_ws + kMaxPasses * kBins;
  uint32_t* status = w.d_ws + kHeadWords;
  for (int p = 0; p < passes; ++p) {
    hipLaunchKernelGGL((k_rs_pass<K, HAS_VALUES, THREADS, ITEMS>), dim3(tiles), dim3(THREADS), 0, stream, kin, kout, vin,
                       vout, (uint32_t)n, (int)begin_bit + p * kRadixBits, hist + p * kBins, status + (size_t)p * tiles * kBins,
                       tiles <= 1024u ? (uint32_t*)nullptr : tickets + p);
    K* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
}

template <typename K, bool HAS_VALUES>
inline hipError_t sort(Workspace& w, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n,
                       unsigned end_bit, hipStream_t stream, K** keys_result, uint32_t** vals_result,
                       unsigned begin_bit = 0) {
  *keys_result = keys_a;
  if (vals_result) *vals_result = vals_a;
  if (n == 0 || end_bit <= begin_bit) return hipSuccess;
  if (end_bit > sizeof(K) * 8) end_bit = (unsigned)(sizeof(K) * 8);
  const int passes = (int)((end_bit - begin_bit + kRadixBits - 1) / kRadixBits);
  // tile size: keep the number of tiles (look-back chain depth) small for per-frame sizes
  const int tile = (n <= (1u << 20)) ? 2048 : (n <= (1u << 24)) ? 8192 : 16384;
  const uint32_t tiles = (uint32_t)((n + tile - 1) / tile);
  const size_t words = kHeadWords + (size_t)passes * tiles * kBins;
  hipError_t e = ensure(w, words);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(w.d_ws, 0, words * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_rs_hist<K>), dim3((uint32_t)((n + 2047) / 2048)), dim3(256), 0, stream, keys_a, (uint32_t)n,
                     passes, (int)begin_bit, w.d_ws);
  K* kin = keys_a;
  K* kout = keys_b;
  uint32_t* vin = vals_a;
  uint32_t* vout = vals_b;
  if (tile == 2048) launch_passes<K, HAS_VALUES, 256, 8>(w, kin, kout, vin, vout, n, passes, tiles, begin_bit, stream);
  else if (tile == 8192) launch_passes<K, HAS_VALUES, 512, 16>(w, kin, kout, vin, vout, n, passes, tiles, begin_bit, stream);
  else launch_passes<K, HAS_VALUES, 512, 32>(w, kin, kout, vin, vout, n, passes, tiles, begin_bit, stream);
  *keys_result = kin;
  if (vals_result) *vals_result = vin;
  return hipGetLastError();
}

}  // namespace ksrs
