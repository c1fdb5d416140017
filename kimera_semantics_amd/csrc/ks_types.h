// ks_types.h — device-side types and helpers shared by every kernel of the semantic TSDF
// integrator: per-frame counters, compaction helpers, ray descriptors, the tile hash table and
// voxel pool, per-frame parameters, the integration-order arithmetic, tile get-or-insert.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ks_hip.h"
#include "ks_device_math.h"

namespace ksk {
using namespace ksd;

constexpr uint64_t kEmpty64 = ~0ull;
constexpr uint32_t kEmpty32 = ~0u;
constexpr int kSetBits = 20;                                   // [K:semantic_tsdf_integrator_fast.h:102]
constexpr uint64_t kSetMask = (1ull << kSetBits) - 1;
constexpr uint64_t kFullResetThreshold = 10000;                // [K:semantic_tsdf_integrator_fast.h:107]
constexpr float kPriorInit = -0.60205999132f;                  // [K:include/kimera_semantics/semantic_voxel.h:23]
constexpr int kCoordBias = 1 << 20;                            // voxel coordinates packed as 21-bit fields
constexpr int kTileBias = 1 << 17;                             // tile coordinates packed as 18-bit fields

// error bits raised by kernels
enum : uint32_t { kErrLabel = 1u, kErrPool = 2u, kErrIndex = 4u, kErrTable = 8u, kErrPairs = 16u /* pair buffer too small: the host grows it and repeats the emission */,
                 kErrExact = 32u /* exact early-out: the device-driven fix point gave up, the host-driven loop repeats the frame's stage B */ };

struct Counters {
  unsigned long long n_pairs;
  uint32_t n_valid;
  uint32_t n_rays;
  uint32_t n_xlong;   // of them, runs of more than kXLongRun updates (listed apart: their kernel looks further ahead)
  uint32_t err;
  uint32_t n_long;    // voxel runs handed to the wave-per-run apply kernel
  uint32_t n_long_bundles;
};

constexpr uint32_t kLongRun = 32;        // runs of >= kLongRun updates get a whole wavefront
constexpr uint32_t kLongRunLanes = 16;   // ... where the long runs are taken a lane per run, bucketed by length over the frame (k_apply_long_lanes: frames of
                                         // 2^24 pairs and more), already from 17 updates on: what bounds a tile of k_apply_runs is its longest run
constexpr uint32_t kXLongRun = 1024;     // runs of more than this many updates: the voxels next to the sensor (one chain of 1e4..1e5 updates)
constexpr uint32_t kInvalidSlot = 1u << kSetBits;  // sort key of dropped points (sorts last)

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Two facts about a wavefront that kernels rely on and a host-side functional model (tools/emu) has to be told:
//  * its LDS operations run in program order, so a lane may read what ANOTHER lane of the same wavefront wrote earlier in
//    the program without any barrier in between — marked KS_WAVE_LDS_ORDER() (nothing on the GPU, a wave barrier in the model);
//  * KS_WAIT_VMEM(): the wavefront's outstanding global-memory operations (atomics without return included) have been performed.
#ifndef KS_WAVE_LDS_ORDER
#define KS_WAVE_LDS_ORDER()
#endif
#ifndef KS_WAIT_VMEM
#define KS_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
// KS_WAIT_LOADS(): the same wait as an instruction the COMPILER knows (its own bookkeeping of outstanding loads restarts
// here; after a batch of loads consumed under divergent branches it otherwise re-waits, conservatively, before every later
// memory operation — and a wait completes everything issued before it, atomics included).
#ifndef KS_WAIT_LOADS
#define KS_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)   // vmcnt(0), lgkmcnt / expcnt untouched (gfx9 encoding)
#endif
// KS_VALUE_BARRIER(x): x is a value in a vector register from here on, whatever it was computed from.  (Selecting one of
// several fields of a struct by lane number is otherwise folded into ONE load from a lane-dependent address, which
// pins the whole struct to scratch memory: a round trip through the vector memory path where a v_cndmask would do.)
#ifndef KS_VALUE_BARRIER
#define KS_VALUE_BARRIER(x) asm volatile("" : "+v"(x))
#endif

// Compaction slot for lanes with pred == true: one atomic per wavefront (a same-address
// returning atomic per lane saturates at ~88/us on MI355X).  Must be called converged.
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter) {
  const unsigned long long m = __ballot(pred);
  const uint32_t lane = lane_id();
  uint32_t base = 0;
  if (lane == 0 && m) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, 0);
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
// Block-level variants: ONE atomic per workgroup (every thread of the block must call).
__device__ __forceinline__ uint32_t block_append(bool pred, uint32_t* counter) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_base;
  const unsigned long long m = __ballot(pred);
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < nwaves; ++w) {
      const uint32_t t = s_wave[w];
      s_wave[w] = total;
      total += t;
    }
    s_base = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  const uint32_t pos = s_base + s_wave[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  return pos;
}
__device__ __forceinline__ void block_count(bool pred, uint32_t* counter) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const unsigned long long m = __ballot(pred);
  if (lane_id() == 0 && m) atomicAdd(&s_cnt, (uint32_t)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(counter, s_cnt);
}

struct RayDesc {  // 32 B, indexed by point position p (fast) / bundle first-point position (merged)
  float px, py, pz;   // point_G
  float weight;
  uint32_t color;
  float d_match, d_non;  // pure-label log-likelihood increments
  uint32_t info;         // [7:0] label, [9:8] kind (0 none, 1 pure, 2 mixed), [10] clearing
};

struct TileEntry {
  uint64_t key;
  uint32_t val;   // pool slot (kSlotPending until published)
  uint32_t pad;
};

struct TileTable {
  TileEntry* ent;      // open addressing; key == kEmpty64 = free.  Key and slot share one 16-B
                       // entry so a lookup is ONE memory round trip (it sits on the ray-march chain)
  uint64_t* slot_keys; // slot -> packed tile key
  uint32_t* n_tiles;   // persistent: tiles allocated so far (never reset between frames)
  uint32_t mask;       // capacity - 1
  uint32_t max_tiles;
};

struct Pool {
  // One 128-byte record per voxel (array of structures, 8 x uint4):
  //   dword 0 distance | 1 weight | 2 colour (rgba) | 3 label (255 = never updated)
  //   dwords 4..24 the 21 class priors | 25..31 spare
  // A record is exactly one 128-B line: the 8 lanes that cooperate on a voxel move it with one
  // coalesced 16-B access each, and a whole tile (512 voxels) is one contiguous 64 KiB range.
  uint4* vox;          // [tile][512][8]
  uint8_t* updated;    // per tile: touched since the host last fetched the updated-block list
  uint8_t* dirty;      // per tile: touched since the last multi-GPU reduce (ks_reduce)
};

struct FrameParams {
  Pose T;
  float voxel_size_inv;
  float min_ray, max_ray, trunc;
  float start_inv;           // start_voxel_subsampling_factor * voxel_size_inv
  float log_match, log_non_match;
  TsdfParams tsdf;
  uint64_t start_offset, observed_offset;
  uint32_t obs_tag, obs_tag_lo;  // frame tag of this frame's marks; first tag of the current offset generation
  uint64_t* observed;            // the early-out table this frame uses (one per march stream)
  int32_t max_collisions;
  uint32_t n;                // points this frame
  // "mixed" order (vxb::MixedThreadSafeIndex): position s < order_groups * order_per is point
  // (s % order_groups) * order_per + s / order_groups, later positions are their own point.
  //   KS_ORDER_MIXED (upstream as published): order_groups = n / 1024, order_per = 1024
  //   KS_ORDER_MIXED_1024_GROUPS:             order_groups = 1024,     order_per = n / 1024
  // (n < 1024: order_groups = 1, order_per = 0 — the identity).  The groups are the chains of the ordered-phase
  // schedule (ks_k_march.h): position s = chain s % chains, generation s / chains.
  uint32_t order_groups, order_per;
  uint32_t chains;
  int carving, allow_clear, freespace, use_const_weight;
  int method, color_mode, early_out, sorted_order;
  int n_dynamic;
  const uint64_t* grazing_keys;  // merged + anti-grazing: sorted end-voxel keys of this frame (else nullptr)
  const uint64_t* ray_keys;      // merged + anti-grazing: end-voxel key of each bundle, by first position
  const uint32_t* order;     // sorted mode: position -> index (nullptr in mixed mode)
  const uint32_t* inv_order; // sorted mode: index -> position
  uint32_t seq_bits;         // low bits of a pair key hold the ray sequence
  uint32_t point_mask;       // (1 << bits_for(n)) - 1
  uint32_t clear_bit;        // merged: sequence bit that orders clearing bundles last
  uint32_t eo_frame;         // fast, exact early-out: number of the frame among those of the exact path (ks_k_exact.h)
  // merged stage A: the points are grouped by a 32-bit key — the end voxel relative to key_base, key_bits bits per axis, bit
  // 3 * key_bits = clearing; a voxel outside that window gets bit 31 | its slot in a small hash table (ks_k_rays.h).  0: the 64-bit
  // end-voxel keys themselves are sorted (anti-grazing searches them; windows wider than 10 bits per axis)
  int32_t key_base[3];
  uint32_t key_bits;
  uint8_t dynamic_labels[32];
};

__device__ __forceinline__ uint32_t point_order(const FrameParams& F, const uint32_t* order, uint32_t p) {
  // vxb::MixedThreadSafeIndex — [K:src/semantic_tsdf_integrator_fast.cpp:172-174]
  if (F.sorted_order) return order[p];
  if (F.order_groups * F.order_per <= p) return p;
  return (p % F.order_groups) * F.order_per + p / F.order_groups;
}

// Anti-grazing (vxb Config::enable_anti_grazing, off by default): a bundle's ray skips voxels that
// are the END voxel of another non-clearing bundle of this frame
// [K:src/semantic_tsdf_integrator_merged.cpp:306-313].  Membership = binary search in the sorted
// point keys (non-clearing keys have bit 63 clear and sort first).
__device__ __forceinline__ bool grazing_skip(const FrameParams& F, int cx, int cy, int cz, bool clearing, uint64_t own_key) {
  if (!F.grazing_keys) return false;
  const int lim = kCoordBias - 1;
  if (abs(cx) >= lim || abs(cy) >= lim || abs(cz) >= lim) return false;
  const uint64_t k = ((uint64_t)(uint32_t)(cx + kCoordBias) << 42) | ((uint64_t)(uint32_t)(cy + kCoordBias) << 21) |
                     (uint64_t)(uint32_t)(cz + kCoordBias);
  if (!clearing && k == own_key) return false;
  uint32_t lo = 0, hi = F.n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (F.grazing_keys[mid] < k) lo = mid + 1;
    else hi = mid;
  }
  return lo < F.n && F.grazing_keys[lo] == k;
}

// Ray descriptors: fast = one per point, stored at the point's memory index; merged = one
// per bundle, stored at the bundle's first position.
__device__ __forceinline__ uint32_t ray_index(const FrameParams& F, uint32_t p) {
  return (F.method == KS_METHOD_FAST) ? point_order(F, F.order, p) : p;
}

// inverse of point_order: integration position of the point stored at index idx
__device__ __forceinline__ uint32_t point_position(const FrameParams& F, const uint32_t* inv_order, uint32_t idx) {
  if (F.sorted_order) return inv_order[idx];
  if (F.order_groups * F.order_per <= idx) return idx;
  return (idx % F.order_per) * F.order_groups + idx / F.order_per;
}

__host__ __device__ __forceinline__ uint64_t pack_tile(int tx, int ty, int tz) {
  return ((uint64_t)(uint32_t)(tx + kTileBias) << 36) | ((uint64_t)(uint32_t)(ty + kTileBias) << 18) |
         (uint64_t)(uint32_t)(tz + kTileBias);
}
__device__ __forceinline__ void unpack_tile(uint64_t k, int& tx, int& ty, int& tz) {
  tx = (int)((k >> 36) & 0x3ffffu) - kTileBias;
  ty = (int)((k >> 18) & 0x3ffffu) - kTileBias;
  tz = (int)(k & 0x3ffffu) - kTileBias;
}
__host__ __device__ __forceinline__ uint32_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

constexpr uint32_t kSlotPending = 0xffffffffu;  // table value before the allocating lane has published the slot
constexpr uint32_t kSlotBad = 0xfffffffeu;      // pool exhausted

// Allocation of a voxel tile on first touch: CAS on the key claims the table entry, an
// atomic bump of the pool counter assigns the slot.  Replaces the reference's temp-block map
// under a global mutex, [K:src/semantic_integrator_base.cpp:205-265].
__device__ __forceinline__ void tile_insert(const TileTable& T, Counters* C, uint64_t key) {
  uint32_t h = mix64(key) & T.mask;
  for (uint32_t probes = 0; probes <= T.mask; ++probes) {
    const uint64_t k = __hip_atomic_load(&T.ent[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return;
    if (k == kEmpty64) {
      const uint64_t old = atomicCAS((unsigned long long*)&T.ent[h].key, (unsigned long long)kEmpty64, (unsigned long long)key);
      if (old == kEmpty64) {
        uint32_t slot = atomicAdd(T.n_tiles, 1u);
        if (slot < T.max_tiles) {
          T.slot_keys[slot] = key;
        } else {
          atomicOr(&C->err, kErrPool);
          slot = kSlotBad;
        }
        __hip_atomic_store(&T.ent[h].val, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      if (old == key) return;
    }
    h = (h + 1) & T.mask;
  }
  atomicOr(&C->err, kErrTable);
}

// get-or-insert WITHOUT waiting: returns the slot, or kSlotPending (with the table position in
// *hpos) when another lane has claimed the key but not yet published its slot.  Waiting is done
// by the caller after the wave has reconverged, so a waiting lane can never sit in front of the
// publishing lane of its own wavefront.
__device__ __forceinline__ uint32_t tile_slot_nowait(const TileTable& T, Counters* C, uint64_t key, uint32_t* hpos) {
  uint32_t h = mix64(key) & T.mask;
  for (uint32_t probes = 0; probes <= T.mask; ++probes) {
    // one 16-B plain load first: tiles of earlier frames hit here with key and slot together
    const uint4 e = *(const uint4*)&T.ent[h];
    uint64_t k = (uint64_t)e.x | ((uint64_t)e.y << 32);
    if (k == key && e.z != kSlotPending) {
      *hpos = h;
      return e.z;
    }
    if (k != key) k = __hip_atomic_load(&T.ent[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmpty64) {
      const uint64_t old = atomicCAS((unsigned long long*)&T.ent[h].key, (unsigned long long)kEmpty64, (unsigned long long)key);
      if (old == kEmpty64) {
        uint32_t slot = atomicAdd(T.n_tiles, 1u);
        if (slot < T.max_tiles) {
          T.slot_keys[slot] = key;
        } else {
          atomicOr(&C->err, kErrPool);
          slot = kSlotBad;
        }
        __hip_atomic_store(&T.ent[h].val, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return slot;
      }
      k = old;
    }
    if (k == key) {
      *hpos = h;
      return __hip_atomic_load(&T.ent[h].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    h = (h + 1) & T.mask;
  }
  atomicOr(&C->err, kErrTable);
  return kSlotBad;
}

__device__ __forceinline__ uint32_t tile_lookup(const TileTable& T, uint64_t key) {
  uint32_t h = mix64(key) & T.mask;
  for (uint32_t probes = 0; probes <= T.mask; ++probes) {
    const uint4 e = *(const uint4*)&T.ent[h];
    const uint64_t k = (uint64_t)e.x | ((uint64_t)e.y << 32);
    if (k == key) return e.z;
    if (k == kEmpty64) return 0xffffffffu;
    h = (h + 1) & T.mask;
  }
  return 0xffffffffu;
}

__device__ __forceinline__ float bcast_f(float x, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k));
}
__device__ __forceinline__ uint32_t bcast_u(uint32_t x, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)x, k); }
// x of the lane below; lane 0 reads `first`.  One DPP move (wave_shr:1): no LDS, no scalar round trip — the step of a
// recurrence that runs THROUGH the lanes (lane k's value from lane k-1's).
#ifndef KS_LANE_BELOW
#define KS_LANE_BELOW(first_i, x_i) __builtin_amdgcn_update_dpp((first_i), (x_i), 0x138, 0xf, 0xf, false)
#endif
__device__ __forceinline__ float lane_below_f(float x, float first) {
  return __int_as_float(KS_LANE_BELOW(__float_as_int(first), __float_as_int(x)));
}

// Correctly rounded a / b given r = RN(1/b) (Markstein): q0 = RN(a r); rem = a - q0 b (exact
// with FMA); q = RN(q0 + rem r).  Outside a comfortable exponent window fall back to the
// hardware IEEE division so subnormal remainders cannot perturb the result.
__device__ __forceinline__ float div_by_recip(float a, float b, float r) {
  const float aa = fabsf(a);
  if (aa >= 1e-20f && aa <= 1e20f) {
    const float q0 = a * r;
    const float rem = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(rem, r, q0);
  }
  return a / b;
}

}  // namespace ksk
