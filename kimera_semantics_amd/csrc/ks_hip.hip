// ks_hip.hip — MI355X (gfx950) semantic TSDF integrator: HIP kernels + the C ABI of include/ks_hip.h.
//
// Replaces, behind the reference's plugin surface, the CPU hot path
//   kimera::FastSemanticTsdfIntegrator::integratePointCloud    [K:src/semantic_tsdf_integrator_fast.cpp:57-199]
//   kimera::MergedSemanticTsdfIntegrator::integratePointCloud  [K:src/semantic_tsdf_integrator_merged.cpp:65-329]
//   kimera::SemanticIntegratorBase::updateSemanticVoxel         [K:src/semantic_integrator_base.cpp:136-194, 283-380]
// ([K:...] = path under /root/reference/kimera_semantics/).
//
// Pipeline per frame (all on one HIP stream; no CPU fallback exists):
//   points  : one lane per point — validity, T_G_C * p, start-voxel / end-voxel key          (k_points_*)
//   sort    : radix sort of point keys (start-voxel dedup slots | end-voxel bundles)
//   rays    : exact sequential-equivalent dedup (fast) or per-bundle merge (merged)           (k_dedup / k_bundles)
//   march   : one lane per ray — ONE DDA walk: tile allocation in the spatial hash, early-out,
//             (voxel slot, ray seq) pairs staged per wavefront in LDS                          (k_march)
//   sort    : radix sort of pairs => every voxel's updates contiguous, in reference order
//   apply   : one lane per voxel run — sequential TSDF + log-likelihood update, one RMW        (k_apply)
// Ordering contract: per voxel, updates are applied in exactly the order the reference's
// single-threaded integrator would apply them, which makes labels bit-exact.
//
// Data layout in HBM: 8x8x8-voxel tiles, struct-of-arrays per tile
//   dist f32[512] | weight f32[512] | color u32[512] | label u8[512] | priors f32[21][512]
// addressed through an open-addressing hash table keyed by the packed tile index.

#include <chrono>
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/ks_hip.h"
#include "ks_device_math.h"
#include "ks_radix_sort.h"

using namespace ksd;

namespace {

constexpr uint64_t kEmpty64 = ~0ull;
constexpr int kSetBits = 20;                                   // [K:semantic_tsdf_integrator_fast.h:102]
constexpr uint64_t kSetMask = (1ull << kSetBits) - 1;
constexpr uint64_t kFullResetThreshold = 10000;                // [K:semantic_tsdf_integrator_fast.h:107]
constexpr float kPriorInit = -0.60205999132f;                  // [K:include/kimera_semantics/semantic_voxel.h:23]
constexpr int kCoordBias = 1 << 20;                            // voxel coordinates packed as 21-bit fields
constexpr int kTileBias = 1 << 17;                             // tile coordinates packed as 18-bit fields

// error bits raised by kernels
enum : uint32_t { kErrLabel = 1u, kErrPool = 2u, kErrIndex = 4u, kErrTable = 8u };

struct Counters {
  unsigned long long n_pairs;
  uint32_t n_valid;
  uint32_t n_rays;
  uint32_t pad0;
  uint32_t err;
  uint32_t n_long;    // voxel runs handed to the wave-per-run apply kernel
  uint32_t n_long_bundles;
};

constexpr uint32_t kLongRun = 32;        // runs of >= kLongRun updates get a whole wavefront
constexpr uint32_t kInvalidSlot = 1u << kSetBits;  // sort key of dropped points (sorts last)

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

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
  uint8_t* updated;    // per tile
};

struct FrameParams {
  Pose T;
  float voxel_size_inv;
  float min_ray, max_ray, trunc;
  float start_inv;           // start_voxel_subsampling_factor * voxel_size_inv
  float log_match, log_non_match;
  TsdfParams tsdf;
  uint64_t start_offset, observed_offset;
  int32_t max_collisions;
  uint32_t n;                // points this frame
  uint32_t per_group;        // n / 1024 (mixed order)
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
  uint8_t dynamic_labels[32];
};

__device__ __forceinline__ uint32_t point_order(const FrameParams& F, const uint32_t* order, uint32_t p) {
  // vxb::MixedThreadSafeIndex — [K:src/semantic_tsdf_integrator_fast.cpp:172-174]
  if (F.sorted_order) return order[p];
  if (1024u * F.per_group <= p) return p;
  return (p % 1024u) * F.per_group + p / 1024u;
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
  if (1024u * F.per_group <= idx) return idx;
  return (idx % F.per_group) * 1024u + idx / F.per_group;
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

// ------------------------------------------------------------------------------------------
// K1/K2 (fast): per point — label, validity, dynamic-label filter, point_G, start-voxel slot.
// [K:src/semantic_tsdf_integrator_fast.cpp:71-92, 150-158]
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_points_fast(FrameParams F, const float* __restrict__ xyz,
                                                      const uint8_t* __restrict__ rgba,
                                                      const uint8_t* __restrict__ labels,
                                                      const uint8_t* __restrict__ color_lut,
                                                      RayDesc* __restrict__ rays, uint32_t* __restrict__ hash_out,
                                                      uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                      Counters* C) {
  // One lane per point in MEMORY order (coalesced reads, coalesced descriptor writes); the
  // integration position p of the point is arithmetic, only the 4-byte sort key is scattered.
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  bool counted = false;
  if (idx < F.n) {
    uint32_t key = kInvalidSlot;
    const f3 pc = {xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
    uint32_t color = 0;
    if (rgba) color = ((const uint32_t*)rgba)[idx];
    uint32_t label;
    if (labels) label = labels[idx];
    else label = color_lut ? color_lut[color & 0xffffffu] : 0u;
    if (label >= (uint32_t)kNumLabels) {
      atomicOr(&C->err, kErrLabel);
    } else {
      int valid = point_validity(pc, F.min_ray, F.max_ray, F.allow_clear != 0, F.freespace != 0);
      for (int i = 0; i < F.n_dynamic; ++i)
        if (F.dynamic_labels[i] == label) valid = 0;
      if (valid) {
        const f3 pg = transform_point(F.T, pc);
        const float gx = grid_coord(pg.x, F.start_inv), gy = grid_coord(pg.y, F.start_inv),
                    gz = grid_coord(pg.z, F.start_inv);
        const float lim = 2.0f * (float)kCoordBias;  // finer grid; only the hash of the index is used
        if (!(fabsf(gx) < lim && fabsf(gy) < lim && fabsf(gz) < lim)) {
          atomicOr(&C->err, kErrIndex);
        } else {
          const uint32_t h = index_hash((int)gx, (int)gy, (int)gz);
          hash_out[idx] = h;
          key = (uint32_t)(((uint64_t)h + F.start_offset) & kSetMask);
          RayDesc d;
          d.px = pg.x; d.py = pg.y; d.pz = pg.z;
          d.weight = voxel_weight(pc.z, F.use_const_weight != 0);
          d.color = color;
          d.d_match = F.log_match;
          d.d_non = F.log_non_match;
          d.info = label | ((label != 0u ? 1u : 0u) << 8) | ((valid == 2 ? 1u : 0u) << 10);
          rays[idx] = d;
          counted = true;
        }
      }
    }
    keys[point_position(F, F.inv_order, idx)] = key;
    vals[idx] = idx;  // identity: vals[p] = p
  }
  block_count(counted, &C->n_valid);
}

// Start-voxel dedup, exactly as the serial reference.  ApproxHashSet::replaceHash leaves the
// caller's hash in the slot whether or not it was already there, so a point is kept iff the
// previous point that mapped to the same slot (in integration order) had a different hash —
// or, for the first point of a slot this frame, iff the slot's persistent content differs.
// Input is stably sorted by slot (so position order is preserved inside a slot).
// [K:src/semantic_tsdf_integrator_fast.cpp:87-92]
__global__ void __launch_bounds__(1024) k_dedup(FrameParams F, const uint32_t* __restrict__ skeys,
                                                const uint32_t* __restrict__ svals, const uint32_t* __restrict__ hash,
                                                uint64_t* __restrict__ start_set, uint32_t* __restrict__ ray_list,
                                                Counters* C) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = F.n;
  bool kept = false;
  uint32_t p = 0;
  if (i < n && C->err == 0) {
    const uint32_t slot = skeys[i];
    if (slot != kInvalidSlot) {
      p = svals[i];
      const uint64_t h = hash[point_order(F, F.order, p)];
      const bool first = (i == 0) || (skeys[i - 1] != slot);
      uint64_t prev;
      // the slot's persistent content is only READ here; k_dedup_commit writes it afterwards
      if (first) prev = start_set[slot];
      else prev = hash[point_order(F, F.order, svals[i - 1])];
      kept = prev != h;
    }
  }
  const uint32_t pos = block_append(kept, &C->n_rays);
  if (kept) ray_list[pos] = p;
}

// Leaves the last hash of every slot run in the persistent approximate set (what
// replaceHash would have left behind after the frame).
__global__ void __launch_bounds__(1024) k_dedup_commit(FrameParams F, const uint32_t* __restrict__ skeys,
                                                       const uint32_t* __restrict__ svals,
                                                       const uint32_t* __restrict__ hash, uint64_t* __restrict__ start_set,
                                                       const Counters* C) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n || C->err != 0) return;
  const uint32_t slot = skeys[i];
  if (slot == kInvalidSlot) return;
  if (i + 1 < F.n && skeys[i + 1] == slot) return;
  start_set[slot] = (uint64_t)hash[point_order(F, F.order, svals[i])];
}

// ------------------------------------------------------------------------------------------
// K4 (merged): per point — validity, point_G, end-voxel key.  vxb::MergedTsdfIntegrator::bundleRays,
// called at [K:src/semantic_tsdf_integrator_merged.cpp:119-124].
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_points_merged(FrameParams F, const float* __restrict__ xyz,
                                                        const uint8_t* __restrict__ rgba,
                                                        const uint8_t* __restrict__ labels,
                                                        const uint8_t* __restrict__ color_lut,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        Counters* C) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  bool counted = false;
  if (idx < F.n) {
    const f3 pc = {xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
    uint32_t label;
    if (labels) label = labels[idx];
    else label = (color_lut && rgba) ? color_lut[((const uint32_t*)rgba)[idx] & 0xffffffu] : 0u;
    uint64_t key = kEmpty64;
    if (label >= (uint32_t)kNumLabels) {
      atomicOr(&C->err, kErrLabel);
    } else {
      const int valid = point_validity(pc, F.min_ray, F.max_ray, F.allow_clear != 0, F.freespace != 0);
      if (valid) {
        const f3 pg = transform_point(F.T, pc);
        const float gx = grid_coord(pg.x, F.voxel_size_inv), gy = grid_coord(pg.y, F.voxel_size_inv),
                    gz = grid_coord(pg.z, F.voxel_size_inv);
        const float lim = (float)(kCoordBias - 1);
        if (!(fabsf(gx) < lim && fabsf(gy) < lim && fabsf(gz) < lim)) {
          atomicOr(&C->err, kErrIndex);
        } else {
          key = ((uint64_t)(valid == 2 ? 1u : 0u) << 63) | ((uint64_t)(uint32_t)((int)gx + kCoordBias) << 42) |
                ((uint64_t)(uint32_t)((int)gy + kCoordBias) << 21) | (uint64_t)(uint32_t)((int)gz + kCoordBias);
          counted = true;
        }
      }
    }
    keys[point_position(F, F.inv_order, idx)] = key;
    vals[idx] = idx;  // identity: vals[p] = p
  }
  block_count(counted, &C->n_valid);
}

// Gather the per-point operands of the bundle merge into bundle (sorted) order, so that the
// sequential merge below streams contiguous memory: {x, y, z, weight} and {label, colour}.
__global__ void __launch_bounds__(256) k_gather_sorted(FrameParams F, const float* __restrict__ xyz,
                                                       const uint8_t* __restrict__ rgba,
                                                       const uint8_t* __restrict__ labels,
                                                       const uint8_t* __restrict__ color_lut,
                                                       const uint32_t* __restrict__ order,
                                                       const uint64_t* __restrict__ skeys,
                                                       const uint32_t* __restrict__ svals, float4* __restrict__ g_pw,
                                                       uint2* __restrict__ g_lc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  if (skeys[i] == kEmpty64) return;
  const uint32_t idx = point_order(F, order, svals[i]);
  const f3 pc = {xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
  const uint32_t color = rgba ? ((const uint32_t*)rgba)[idx] : 0u;
  uint32_t label;
  if (labels) label = labels[idx];
  else label = (color_lut && rgba) ? color_lut[color & 0xffffffu] : 0u;
  g_pw[i] = make_float4(pc.x, pc.y, pc.z, voxel_weight(pc.z, F.use_const_weight != 0));
  g_lc[i] = make_uint2(label, color);
}

// K5 (merged): bundle merge — running weighted mean of point_C, colour blend, label histogram,
// log-likelihood increment.  [K:src/semantic_tsdf_integrator_merged.cpp:248-287]
//   k_bundles      : one lane per bundle of < kLongRun points
//   k_bundles_long : one wavefront per larger bundle (a surface close to the sensor puts
//                    thousands of pixels into one 5 cm voxel)
__device__ __forceinline__ void finish_bundle(const FrameParams& F, f3 mp, float mw, uint32_t merged_color,
                                              bool clearing, int n_labels, int the_label, float c, RayDesc* out) {
  const f3 pg = transform_point(F.T, mp);
  RayDesc d;
  d.px = pg.x; d.py = pg.y; d.pz = pg.z;
  d.weight = mw;
  d.color = merged_color;
  d.d_match = 0.0f;
  d.d_non = 0.0f;
  uint32_t kind = 0;
  if (n_labels == 1) {
    kind = 1;
    d.d_match = F.log_match * c;
    d.d_non = F.log_non_match * c;
  } else if (n_labels > 1) {
    kind = 2;
  }
  d.info = (uint32_t)the_label | (kind << 8) | ((clearing ? 1u : 0u) << 10);
  *out = d;
}

__global__ void __launch_bounds__(256) k_bundles(FrameParams F, const uint64_t* __restrict__ skeys,
                                                 const uint32_t* __restrict__ svals, const float4* __restrict__ g_pw,
                                                 const uint2* __restrict__ g_lc, RayDesc* __restrict__ rays,
                                                 float* __restrict__ deltas, uint32_t* __restrict__ ray_list,
                                                 uint32_t* __restrict__ long_list, uint64_t* __restrict__ ray_keys,
                                                 Counters* C) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool head = false, is_long = false;
  uint32_t first_p = 0;
  uint64_t key = 0;
  if (i < F.n && C->err == 0) {
    key = skeys[i];
    head = (key != kEmpty64) && (i == 0 || skeys[i - 1] != key);
    if (head) is_long = (i + kLongRun < F.n) && (skeys[i + kLongRun] == key);
  }
  const uint32_t lpos = block_append(head && is_long, &C->n_long_bundles);
  if (head && is_long) long_list[lpos] = i;
  const bool work = head && !is_long;
  if (work) {
    const bool clearing = (key >> 63) != 0;
    uint32_t merged_color = 0;
    f3 mp = {0.f, 0.f, 0.f};
    float mw = 0.0f;
    float freq[kNumLabels];
#pragma unroll
    for (int l = 0; l < kNumLabels; ++l) freq[l] = 0.0f;
    uint32_t j = i;
    do {
      const float4 q = g_pw[j];
      const float pw = q.w;
      if (!(pw < kEps)) {
        const uint2 lc = g_lc[j];
        const float denom = mw + pw;
        mp.x = (mp.x * mw + q.x * pw) / denom;
        mp.y = (mp.y * mw + q.y * pw) / denom;
        mp.z = (mp.z * mw + q.z * pw) / denom;
        if (F.color_mode == KS_COLOR_MODE_COLOR) merged_color = blend_two_colors(merged_color, mw, lc.y, pw);
        mw += pw;
#pragma unroll
        for (int l = 0; l < kNumLabels; ++l) freq[l] += (lc.x == (uint32_t)l) ? 1.0f : 0.0f;
        if (clearing) break;
      }
      ++j;
    } while (j < F.n && skeys[j] == key);

    first_p = svals[i];
    // priors += L * freq with L[i][j] = (j == 0) ? 0 : (i == j ? log p : log(1-p)), j ascending, no FMA
    // [K:src/semantic_integrator_base.cpp:93-128, 306-307]
    int n_labels = 0, the_label = 0;
    float c = 0.0f;
#pragma unroll
    for (int l = 1; l < kNumLabels; ++l)
      if (freq[l] > 0.0f) { ++n_labels; the_label = l; c = freq[l]; }
    if (n_labels > 1) {
#pragma unroll
      for (int r = 0; r < kNumLabels; ++r) {
        float acc = 0.0f;
        acc += 0.0f * freq[0];
#pragma unroll
        for (int l = 1; l < kNumLabels; ++l) acc += ((r == l) ? F.log_match : F.log_non_match) * freq[l];
        deltas[(size_t)first_p * kNumLabels + r] = acc;
      }
    }
    finish_bundle(F, mp, mw, merged_color, clearing, n_labels, the_label, c, &rays[first_p]);
    if (ray_keys) ray_keys[first_p] = key & ~(1ull << 63);
  }
  const uint32_t pos = block_append(work, &C->n_rays);
  if (work) ray_list[pos] = first_p;
}

__global__ void __launch_bounds__(64) k_bundles_long(FrameParams F, const uint64_t* __restrict__ skeys,
                                                     const uint32_t* __restrict__ svals,
                                                     const float4* __restrict__ g_pw, const uint2* __restrict__ g_lc,
                                                     RayDesc* __restrict__ rays, float* __restrict__ deltas,
                                                     uint32_t* __restrict__ ray_list,
                                                     const uint32_t* __restrict__ long_list,
                                                     uint64_t* __restrict__ ray_keys, Counters* C) {
  const uint32_t n_long = C->n_long_bundles;
  const int lane = (int)lane_id();
  for (uint32_t run = blockIdx.x; run < n_long; run += gridDim.x) {
    const uint32_t start = long_list[run];
    const uint64_t key = skeys[start];
    const bool clearing = (key >> 63) != 0;
    float mpc = 0.0f;  // lane 0/1/2: x/y/z of the running weighted mean
    float mw = 0.0f;
    uint32_t merged_color = 0;
    float freq = 0.0f;  // lane l < 21 counts label l
    bool done = false;
    uint32_t base = start;
    // prefetch one batch ahead (contiguous, coalesced)
    uint32_t j = base + (uint32_t)lane;
    bool in = (j < F.n) && (skeys[j] == key);
    float4 q = in ? g_pw[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 lc = in ? g_lc[j] : make_uint2(0u, 0u);
    while (!done) {
      const int cnt = (int)__popcll(__ballot(in));
      if (cnt == 0) break;
      const uint32_t jn = base + 64u + (uint32_t)lane;
      const bool in_n = (jn < F.n) && (skeys[jn] == key);
      const float4 q_n = in_n ? g_pw[jn] : make_float4(0.f, 0.f, 0.f, 0.f);
      const uint2 lc_n = in_n ? g_lc[jn] : make_uint2(0u, 0u);

      const bool valid = in && !(q.w < kEps);
      unsigned long long vmask = __ballot(valid);
      if (clearing && vmask) {  // only the first usable point of a clearing bundle is integrated
        vmask &= (~vmask + 1ull);
        done = true;
      }
      const bool use = valid && ((vmask >> lane) & 1ull);
      // pass 1: weight recurrence; lane k keeps (weight before, denominator)
      float my_mw = 0.0f, my_den = 1.0f;
      for (unsigned long long m = vmask; m; m &= m - 1ull) {
        const int k = __ffsll((long long)m) - 1;
        const float den = mw + bcast_f(q.w, k);
        if (lane == k) { my_mw = mw; my_den = den; }
        mw = den;
      }
      const float my_r = 1.0f / my_den;
      const float ax = q.x * q.w, ay = q.y * q.w, az = q.z * q.w;
      // pass 2: weighted-mean recurrence; lanes 0,1,2 each walk ONE component chain (x, y, z),
      // so a step is one multiply-add + one reciprocal-based division for the whole wave
      for (unsigned long long m = vmask; m; m &= m - 1ull) {
        const int k = __ffsll((long long)m) - 1;
        const float mw_k = bcast_f(my_mw, k), den_k = bcast_f(my_den, k), r_k = bcast_f(my_r, k);
        const float ax_k = bcast_f(ax, k), ay_k = bcast_f(ay, k), az_k = bcast_f(az, k);
        const float a_k = (lane == 0) ? ax_k : (lane == 1) ? ay_k : az_k;
        mpc = div_by_recip(mpc * mw_k + a_k, den_k, r_k);
        if (F.color_mode == KS_COLOR_MODE_COLOR)
          merged_color = blend_two_colors(merged_color, mw_k, bcast_u(lc.y, k), bcast_f(q.w, k));
      }
      // label histogram: counts are order independent and exact in f32
#pragma unroll
      for (int l = 0; l < kNumLabels; ++l) {
        const unsigned long long lm = __ballot(use && lc.x == (uint32_t)l);
        if (lane == l) freq += (float)__popcll(lm);
      }
      if (cnt < 64) break;
      in = in_n;
      q = q_n;
      lc = lc_n;
      base += 64u;
    }
    const f3 mp = {bcast_f(mpc, 0), bcast_f(mpc, 1), bcast_f(mpc, 2)};
    const uint32_t first_p = svals[start];
    const unsigned long long present = __ballot(lane >= 1 && lane < kNumLabels && freq > 0.0f);
    const int n_labels = (int)__popcll(present);
    const int the_label = present ? (63 - __clzll((long long)present)) : 0;
    const float c = bcast_f(freq, the_label);
    if (n_labels > 1 && lane < kNumLabels) {
      float acc = 0.0f;
      acc += 0.0f * bcast_f(freq, 0);
#pragma unroll
      for (int l = 1; l < kNumLabels; ++l) acc += ((lane == l) ? F.log_match : F.log_non_match) * bcast_f(freq, l);
      deltas[(size_t)first_p * kNumLabels + lane] = acc;
    }
    if (lane == 0) {
      finish_bundle(F, mp, mw, merged_color, clearing, n_labels, the_label, c, &rays[first_p]);
      if (ray_keys) ray_keys[first_p] = key & ~(1ull << 63);
      ray_list[atomicAdd(&C->n_rays, 1u)] = first_p;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K3a/K3b: march + emit — ONE DDA walk per ray: tile allocation in the spatial hash, optional
// observed-set early-out, and one (voxel slot id, ray sequence) key per update.  Keys are staged
// in a per-wavefront LDS buffer and flushed with one global atomic per flush (a per-lane or even
// per-step atomic on the pair counter would serialise at ~88/us).  Launched over an upper bound
// of rays; the live count is read from device memory, so the host does not synchronise between
// the ray stage and the march.
// [K:src/semantic_tsdf_integrator_fast.cpp:94-141], [K:src/semantic_tsdf_integrator_merged.cpp:288-328]
// ------------------------------------------------------------------------------------------
constexpr uint32_t kWaveBuf = 512;  // pair keys staged per wavefront (4 KiB)

__global__ void __launch_bounds__(256) k_march(FrameParams F, const uint32_t* __restrict__ ray_list,
                                               const RayDesc* __restrict__ rays, TileTable T, Pool P,
                                               uint64_t* __restrict__ observed_set, uint64_t* __restrict__ pairs,
                                               unsigned long long pairs_cap, Counters* C) {
  __shared__ uint64_t s_buf[4][kWaveBuf];
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_rays = C->n_rays;
  if (blockIdx.x * blockDim.x >= n_rays) return;  // whole block idle (uniform)
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  uint64_t* buf = s_buf[wave];

  bool done = true;
  Dda dda{};
  uint32_t seq = 0;
  bool clearing = false;
  uint64_t own_key = 0;
  if (r < n_rays && (C->err & (kErrLabel | kErrIndex)) == 0) {
    const uint32_t p = ray_list[r];
    const RayDesc d = rays[ray_index(F, p)];
    clearing = ((d.info >> 10) & 1u) != 0;
    dda.setup(F.T.t, {d.px, d.py, d.pz}, clearing, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc,
              /*cast_from_origin=*/F.method == KS_METHOD_MERGED);
    // merged: normal bundles integrate before clearing bundles ([K:src/semantic_tsdf_integrator_merged.cpp:126-144])
    seq = (F.method == KS_METHOD_MERGED && clearing) ? (p | F.clear_bit) : p;
    own_key = F.ray_keys ? F.ray_keys[p] : 0ull;
    if (!dda.in_range) atomicOr(&C->err, kErrIndex);
    else done = false;
  }

  uint32_t wcount = 0;  // keys in this wave's buffer (wave-uniform)
  auto flush = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&C->n_pairs, (unsigned long long)wcount);
    base = __shfl(base, 0);
    if (base + wcount <= pairs_cap) {
      for (uint32_t i = lane; i < wcount; i += 64) pairs[base + i] = buf[i];
    } else if (lane == 0) {
      atomicOr(&C->err, kErrTable);
    }
    wcount = 0;
  };

  int s = 0;
  int consecutive = 0;
  uint64_t last_tile = kEmpty64;
  uint32_t slot = 0;
  constexpr int kBatch = 4;
  while (__ballot(!done) != 0ull) {
    // ---- (A) which of the next steps of this ray are integrated ----
    // Early-out: the ray stops at the first voxel that makes `consecutive` exceed the limit.
    // With the counter at c, the next (limit + 1 - c) voxels are visited whatever their
    // state, so that many approximate-set exchanges can be IN FLIGHT TOGETHER without
    // speculation; the stop can only fall on the last of them.  On the long rays (the first
    // through their corridor, c stays 0) this cuts the dependent L2 round trips 3-4x.
    int vx[kBatch], vy[kBatch], vz[kBatch];
    uint64_t tk[kBatch];   // tile key of each step
    uint4 pre[kBatch];     // its first-probe table entry, loaded TOGETHER with the exchanges below:
                           // the tile lookup leaves the dependent chain of the ray
    bool em[kBatch];       // step emits an update
    int n_adv = 0;         // steps of this iteration the DDA advances over
#pragma unroll
    for (int j = 0; j < kBatch; ++j) em[j] = false;
    if (!done) {
      const int remaining = dda.steps - s + 1;
      if (remaining <= 0) {
        done = true;
      } else if (F.early_out) {
        int k = F.max_collisions + 1 - consecutive;
        k = k < 1 ? 1 : (k > kBatch ? kBatch : k);
        k = k > remaining ? remaining : k;
        uint64_t hh[kBatch], old[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          if (j < k) {
            vx[j] = dda.cx; vy[j] = dda.cy; vz[j] = dda.cz;
            hh[j] = (uint64_t)index_hash(dda.cx, dda.cy, dda.cz);
            // ApproxHashSet::replaceHash on voxel_observed_approx_set_ — racy by design in the
            // multi-threaded reference; here one atomic exchange per visited voxel.
            old[j] = atomicExch((unsigned long long*)&observed_set[(hh[j] + F.observed_offset) & kSetMask],
                                (unsigned long long)hh[j]);
            dda.advance();
          }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          if (j < k) {
            tk[j] = pack_tile(vx[j] >> 3, vy[j] >> 3, vz[j] >> 3);
            pre[j] = *(const uint4*)&T.ent[mix64(tk[j]) & T.mask];
          }
        }
        int n_upd = k;
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          if (j < k && !done) {
            if (old[j] == hh[j]) ++consecutive;
            else consecutive = 0;
            if (consecutive > F.max_collisions) {
              done = true;   // break BEFORE updating this voxel
              n_upd = j;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) em[j] = j < n_upd;
        n_adv = k;
      } else {
        const int k = remaining < kBatch ? remaining : kBatch;
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          if (j < k) {
            vx[j] = dda.cx; vy[j] = dda.cy; vz[j] = dda.cz;
            tk[j] = pack_tile(vx[j] >> 3, vy[j] >> 3, vz[j] >> 3);
            pre[j] = *(const uint4*)&T.ent[mix64(tk[j]) & T.mask];
            em[j] = !grazing_skip(F, dda.cx, dda.cy, dda.cz, clearing, own_key);
            dda.advance();
          }
        }
        n_adv = k;
      }
      s += n_adv;
    }
    // ---- (B) emit the integrated steps (uniform loop over the batch) ----
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const bool emit = em[j];
      bool any_left = emit;
#pragma unroll
      for (int jj = j + 1; jj < kBatch; ++jj) any_left = any_left || em[jj];
      if (__ballot(any_left) == 0ull) break;
      uint32_t hpos = 0, got = 0;
      bool need_tile = false;
      if (emit && tk[j] != last_tile) {
        need_tile = true;
        last_tile = tk[j];
        const uint64_t k64 = (uint64_t)pre[j].x | ((uint64_t)pre[j].y << 32);
        if (k64 == tk[j] && pre[j].z != kSlotPending) got = pre[j].z;   // resident tile: no further memory access
        else got = tile_slot_nowait(T, C, tk[j], &hpos);
      }
      // the wave has reconverged: every allocating lane of THIS wave has published its slot
      if (need_tile) {
        uint32_t spins = 0;
        while (got == kSlotPending) {
          got = __hip_atomic_load(&T.ent[hpos].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > (1u << 22)) {
            atomicOr(&C->err, kErrTable);
            got = kSlotBad;
          }
        }
        slot = got;
        if (slot < T.max_tiles) P.updated[slot] = 1;
      }
      const unsigned long long m = __ballot(emit);
      if (emit) {
        const uint32_t local = (uint32_t)(vx[j] & 7) + 8u * ((uint32_t)(vy[j] & 7) + 8u * (uint32_t)(vz[j] & 7));
        const uint32_t pos = wcount + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        buf[pos] = ((uint64_t)(slot * (uint32_t)kTileVoxels + local) << F.seq_bits) | seq;
      }
      wcount += (uint32_t)__popcll(m);
      if (wcount > kWaveBuf - 64u) flush();
    }
  }
  if (wcount) flush();
}

// End of stage B: the frame's counters and the persistent tile count go to pinned host memory,
// and the counters are cleared for the slot's next frame (the tail only uses n_long, which it
// expects to be zero): no memset launch per frame.
__global__ void __launch_bounds__(64) k_publish(Counters* __restrict__ C, const uint32_t* __restrict__ n_tiles,
                                                uint32_t* __restrict__ host_snap) {
  static_assert(sizeof(Counters) == 32, "snapshot layout");
  if (threadIdx.x < 8) {
    host_snap[threadIdx.x] = ((const uint32_t*)C)[threadIdx.x];
    ((uint32_t*)C)[threadIdx.x] = 0u;
  }
  if (threadIdx.x == 8) host_snap[8] = *n_tiles;
}

__global__ void __launch_bounds__(512) k_init_tiles(Pool P, uint32_t first_slot) {
  const size_t slot = (size_t)first_slot + blockIdx.x;
  uint4* tile = P.vox + slot * (size_t)kTileVoxels * 8;
  const uint32_t pi = __float_as_uint(kPriorInit);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t q = r * 512u + threadIdx.x;  // uint4 index inside the tile, coalesced
    const uint32_t sub = q & 7u;
    uint4 v;
    if (sub == 0) v = make_uint4(0u, 0u, 0u, 255u);
    else if (sub < 6) v = make_uint4(pi, pi, pi, pi);
    else if (sub == 6) v = make_uint4(pi, 0u, 0u, 0u);
    else v = make_uint4(0u, 0u, 0u, 0u);
    tile[q] = v;
  }
  if (threadIdx.x == 0) P.updated[slot] = 1;
}

// ------------------------------------------------------------------------------------------
// K3c: apply — the per-voxel update.  Pairs are sorted by (voxel, ray sequence); one voxel's
// updates form a contiguous run that is replayed in order:
//   updateTsdfVoxel  (Voxblox; called at [K:fast.cpp:128], [K:merged.cpp:317-319])
//   updateSemanticVoxel: priors += L*freq, argmax, colour  ([K:src/semantic_integrator_base.cpp:136-194])
// One read-modify-write of the voxel per frame however many rays crossed it.
//   k_apply      : one lane per short run (< kLongRun updates); long runs are queued
//   k_apply_long : one wavefront per long run (voxels near the sensor collect thousands of
//                  updates): lanes fetch 64 updates at once and pre-compute the voxel-state-
//                  independent part (sdf, updated weight); the state recurrence is then walked
//                  in order with lane broadcasts; lanes 0..20 own one class prior each.
// ------------------------------------------------------------------------------------------
struct VoxelRef {
  uint32_t slot, local;
  int vx, vy, vz;
};
__device__ __forceinline__ VoxelRef voxel_ref(const TileTable& T, uint32_t vox) {
  VoxelRef v;
  v.slot = vox >> 9;
  v.local = vox & 511u;
  int tx, ty, tz;
  unpack_tile(T.slot_keys[v.slot], tx, ty, tz);
  v.vx = tx * 8 + (int)(v.local & 7u);
  v.vy = ty * 8 + (int)((v.local >> 3) & 7u);
  v.vz = tz * 8 + (int)(v.local >> 6);
  return v;
}

// lane permute with every lane of the wave active (ds_bpermute reads 0 from disabled lanes)
__device__ __forceinline__ uint32_t perm_u(uint32_t x, uint32_t src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)x);
}
__device__ __forceinline__ float perm_f(float x, uint32_t src_lane) { return __uint_as_float(perm_u(__float_as_uint(x), src_lane)); }

// Operands of one (voxel, ray) update that do not depend on the voxel state.
struct UpdateOps {
  float sdf, uw, dm, dn;
  uint32_t info, color, rp;
};
__device__ __forceinline__ UpdateOps load_update_ops(const FrameParams& F, const RayDesc* __restrict__ rays, uint64_t key,
                                                     const VoxelRef& v) {
  UpdateOps u;
  u.rp = (uint32_t)key & F.point_mask;
  const uint4* r4 = (const uint4*)rays + (size_t)ray_index(F, u.rp) * 2;
  const uint4 d0 = r4[0], d1 = r4[1];
  tsdf_operands(F.tsdf, F.T.t, {__uint_as_float(d0.x), __uint_as_float(d0.y), __uint_as_float(d0.z)}, v.vx, v.vy, v.vz,
                __uint_as_float(d0.w), u.sdf, u.uw);
  u.color = d1.x;
  u.dm = __uint_as_float(d1.y);
  u.dn = __uint_as_float(d1.z);
  u.info = d1.w;
  return u;
}

template <int COLOR_MODE>
__global__ void __launch_bounds__(256) k_apply(FrameParams F, unsigned long long n_pairs,
                                               const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                               const float* __restrict__ deltas, TileTable T, Pool P,
                                               const uint32_t* __restrict__ label_lut,
                                               unsigned long long* __restrict__ long_list, Counters* C) {
  // Phase A — one lane per pair (64 consecutive pairs per wavefront): ray-descriptor gather and
  //   the voxel-state-independent half of updateTsdfVoxel, all 64 in flight at once.
  // Phase B — EIGHT LANES COOPERATE PER VOXEL: lane `sub` of a group moves 16 bytes of the
  //   128-byte record (one coalesced line per voxel); sub 0 walks the TSDF recurrence, subs
  //   1..6 own four class priors each.  The run's operands come from the phase-A lanes through
  //   ds_bpermute, so the recurrence has no memory access on its critical path.
  const uint32_t lane = lane_id();
  const unsigned long long wbase = ((unsigned long long)blockIdx.x * 4ull + (threadIdx.x >> 6)) * 64ull;
  const unsigned long long i = wbase + lane;
  const bool valid = i < n_pairs;
  uint64_t key = 0;
  uint32_t vox = 0xffffffffu;
  bool head = false, is_long = false;
  UpdateOps u{};
  if (valid) {
    key = pairs[i];
    vox = (uint32_t)(key >> F.seq_bits);
    head = (i == 0) || ((uint32_t)(pairs[i - 1] >> F.seq_bits) != vox);
    if (head) is_long = (i + kLongRun < n_pairs) && ((uint32_t)(pairs[i + kLongRun] >> F.seq_bits) == vox);
    u = load_update_ops(F, rays, key, voxel_ref(T, vox));
  }
  const uint32_t lpos = block_append(head && is_long, &C->n_long);
  if (head && is_long) long_list[lpos] = i;

  // run boundaries inside the window: every head (short or long) and every invalid lane ends a run
  const unsigned long long bounds = __ballot(head || !valid);
  const unsigned long long H = __ballot(head && !is_long);
  const uint32_t grp = lane >> 3, sub = lane & 7u;
  const uint32_t cbase = (sub - 1u) * 4u;  // first class index of this lane (subs 1..6)
  // The heads are served 8 at a time in lane order: head number r of the window goes to group
  // r % 8 of iteration r / 8.  One forward permute turns "lane -> is a head" into "r -> lane of
  // head r" (heads are sent to [0, nh), every other lane to [nh, 64), so it is a permutation).
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t nh = (uint32_t)__popcll(H);
  const bool is_h = (H >> lane) & 1ull;
  const uint32_t dst = is_h ? (uint32_t)__popcll(H & below) : nh + (uint32_t)__popcll(~H & below);
  const uint32_t head_lane = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)lane);
  // software pipeline over the groups of 8 heads: the record of the NEXT head is requested
  // before the recurrence of the current one runs
  uint32_t it = 0;
  auto next_head = [&]() {
    const uint32_t r = 8u * it + grp;
    ++it;
    const uint32_t p = perm_u(head_lane, r & 63u);
    return r < nh ? (int)p : -1;
  };
  bool more = nh != 0u;
  int nxt_pos = more ? next_head() : -1;
  uint32_t nxt_vox = perm_u(vox, nxt_pos >= 0 ? (uint32_t)nxt_pos : lane);
  uint4 nxt_q = make_uint4(0u, 0u, 0u, 0u);
  if (nxt_pos >= 0 && sub < 7u) nxt_q = (P.vox + (size_t)nxt_vox * 8)[sub];
  while (more) {
    const int my_pos = nxt_pos;
    const bool active = my_pos >= 0;
    const uint32_t hp = active ? (uint32_t)my_pos : lane;
    const uint32_t hvox = nxt_vox;
    const uint4 q = nxt_q;
    more = 8u * it < nh;
    if (more) {
      nxt_pos = next_head();
      nxt_vox = perm_u(vox, nxt_pos >= 0 ? (uint32_t)nxt_pos : lane);
      nxt_q = make_uint4(0u, 0u, 0u, 0u);
      if (nxt_pos >= 0 && sub < 7u) nxt_q = (P.vox + (size_t)nxt_vox * 8)[sub];
    }
    // length of the run inside this window
    uint32_t len = 0;
    if (active) {
      const unsigned long long above = (hp < 63u) ? (bounds >> (hp + 1u)) : 0ull;
      len = above ? (uint32_t)__ffsll((long long)above) : (64u - hp);
    }
    uint4* rec = P.vox + (size_t)(active ? hvox : 0u) * 8;
    float dist = __uint_as_float(q.x), weight = __uint_as_float(q.y);  // meaningful for sub 0
    uint32_t color = q.z;
    float p0 = __uint_as_float(q.x), p1 = __uint_as_float(q.y), p2 = __uint_as_float(q.z), p3 = __uint_as_float(q.w);

    for (uint32_t s = 0;; ++s) {
      const bool on = active && s < len;
      if (__ballot(on) == 0ull) break;
      const uint32_t src = on ? hp + s : lane;
      const float sdf_s = perm_f(u.sdf, src), uw_s = perm_f(u.uw, src);
      // fast: every ray carries the same two increments (log p, log(1-p)); merged: per bundle
      const bool per_ray_inc = F.method == KS_METHOD_MERGED;
      const float dm_s = per_ray_inc ? perm_f(u.dm, src) : F.log_match;
      const float dn_s = per_ray_inc ? perm_f(u.dn, src) : F.log_non_match;
      const uint32_t info_s = perm_u(u.info, src);
      uint32_t color_s = 0, rp_s = 0;
      if (COLOR_MODE == KS_COLOR_MODE_COLOR) color_s = perm_u(u.color, src);
      if (F.method == KS_METHOD_MERGED) rp_s = perm_u(u.rp, src);
      // The step is straight-line code with selects: k_apply is bound by instruction issue (one
      // wave per SIMD slot), and the nested divergent branches of the obvious formulation cost
      // more scalar/branch instructions than the arithmetic they skip.  Every lane evaluates the
      // TSDF recurrence (only sub 0 keeps it) and its four class sums (only subs 1..6 of a
      // pure-label update keep them).
      {
        // updateTsdfVoxel's state half (tsdf_combine), [K:src/semantic_tsdf_integrator_fast.cpp:128]
        const float nw = weight + uw_s;
        const bool upd = on && sub == 0u && !(nw < kEps);
        const float ns = (sdf_s * uw_s + dist * weight) / nw;
        const float nd = (ns > 0.0f) ? std_min(F.tsdf.trunc, ns) : std_max(-F.tsdf.trunc, ns);
        if (COLOR_MODE == KS_COLOR_MODE_COLOR) {
          if (upd && fabsf(sdf_s) < F.tsdf.trunc) color = blend_two_colors(color, weight, color_s, uw_s);
        }
        dist = upd ? nd : dist;
        weight = upd ? std_min(F.tsdf.max_weight, nw) : weight;
      }
      const uint32_t kind = (info_s >> 8) & 3u;
      const bool sem_lane = on && (sub - 1u) < 6u;
      {
        const uint32_t lab = info_s & 0xffu;
        const bool pure = sem_lane && kind == 1u;
        const float a0 = p0 + ((cbase == lab) ? dm_s : dn_s);
        const float a1 = p1 + ((cbase + 1u == lab) ? dm_s : dn_s);
        const float a2 = p2 + ((cbase + 2u == lab) ? dm_s : dn_s);
        const float a3 = p3 + ((cbase + 3u == lab) ? dm_s : dn_s);
        p0 = pure ? a0 : p0;
        p1 = pure ? a1 : p1;
        p2 = pure ? a2 : p2;
        p3 = pure ? a3 : p3;
      }
      if (F.method == KS_METHOD_MERGED) {  // mixed-label bundles carry a 21-entry increment vector
        if (sem_lane && kind == 2u) {
          const float* dl = deltas + (size_t)rp_s * kNumLabels + cbase;
          p0 += dl[0];
          if (sub < 6u) { p1 += dl[1]; p2 += dl[2]; p3 += dl[3]; }
        }
      }
    }
    // a run may continue past the 64-pair window: finish it from global memory (rare)
    if (active && hp + len == 64u) {
      const VoxelRef v = voxel_ref(T, hvox);
      for (unsigned long long j = wbase + 64ull; j < n_pairs; ++j) {
        const uint64_t k = pairs[j];
        if ((uint32_t)(k >> F.seq_bits) != hvox) break;
        const UpdateOps t = load_update_ops(F, rays, k, v);
        if (sub == 0u) {
          tsdf_combine<COLOR_MODE == KS_COLOR_MODE_COLOR>(F.tsdf, t.sdf, t.uw, t.color, dist, weight, color);
        } else if (sub < 7u) {
          const uint32_t kind = (t.info >> 8) & 3u;
          if (kind == 1u) {
            const uint32_t lab = t.info & 0xffu;
            p0 += (cbase == lab) ? t.dm : t.dn;
            p1 += (cbase + 1u == lab) ? t.dm : t.dn;
            p2 += (cbase + 2u == lab) ? t.dm : t.dn;
            p3 += (cbase + 3u == lab) ? t.dm : t.dn;
          } else if (kind == 2u) {
            const float* dl = deltas + (size_t)t.rp * kNumLabels + cbase;
            p0 += dl[0];
            if (sub < 6u) { p1 += dl[1]; p2 += dl[2]; p3 += dl[3]; }
          }
        }
      }
    }

    // calculateMaximumLikelihoodLabel: first strict maximum [K:src/semantic_integrator_base.cpp:352-367]
    float bv = -INFINITY;
    uint32_t bi = 1000u;
    if (sub >= 1u && sub < 7u) {
      bv = p0; bi = cbase;
      if (sub < 6u) {
        if (p1 > bv) { bv = p1; bi = cbase + 1u; }
        if (p2 > bv) { bv = p2; bi = cbase + 2u; }
        if (p3 > bv) { bv = p3; bi = cbase + 3u; }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const float ov = perm_f(bv, lane ^ (uint32_t)o);
      const uint32_t oi = perm_u(bi, lane ^ (uint32_t)o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (active) {
      if (sub == 0u) {
        if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = label_lut[bi];
        else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) color = rainbow_color_map((double)(float)exp((double)bv));
        rec[0] = make_uint4(__float_as_uint(dist), __float_as_uint(weight), color, bi);
      } else if (sub < 6u) {
        rec[sub] = make_uint4(__float_as_uint(p0), __float_as_uint(p1), __float_as_uint(p2), __float_as_uint(p3));
      } else if (sub == 6u) {
        rec[6] = make_uint4(__float_as_uint(p0), 0u, 0u, 0u);
      }
    }
  }
}

template <int COLOR_MODE>
__global__ void __launch_bounds__(64) k_apply_long(FrameParams F, unsigned long long n_pairs,
                                                   const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                   const float* __restrict__ deltas, TileTable T, Pool P,
                                                   const uint32_t* __restrict__ label_lut,
                                                   const unsigned long long* __restrict__ long_list, const Counters* C) {
  // One wavefront per block: LDS traffic below is ordered by program order (DS operations of
  // a wave execute in order), no s_barrier needed; wave_barrier() only pins the compiler.
  __shared__ float s_inc[64][kNumLabels];  // class increments of the 64 updates in flight
  const uint32_t n_long = C->n_long;
  const int lane = (int)lane_id();
  const int cls = lane < kNumLabels ? lane : 0;
  const TsdfParams& Pm = F.tsdf;
  for (uint32_t run = blockIdx.x; run < n_long; run += gridDim.x) {
    const unsigned long long start = long_list[run];
    const uint32_t vox = (uint32_t)(pairs[start] >> F.seq_bits);
    const VoxelRef v = voxel_ref(T, vox);
    uint32_t* rec = (uint32_t*)(P.vox + (size_t)vox * 8);
    float dist = __uint_as_float(rec[0]), weight = __uint_as_float(rec[1]);
    uint32_t color = rec[2];
    float pri = (lane < kNumLabels) ? __uint_as_float(rec[4 + lane]) : 0.0f;
    // voxel centre and the origin->centre vector are constant over the run
    const f3 c = {((float)v.vx + 0.5f) * Pm.voxel_size, ((float)v.vy + 0.5f) * Pm.voxel_size,
                  ((float)v.vz + 0.5f) * Pm.voxel_size};
    const f3 v_voxel_origin = sub3(c, F.T.t);

    // software pipeline: rays of batch b+1 and pair keys of batch b+2 are in flight while batch b is applied
    // All loads of the pipeline are UNCONDITIONAL (indices clamped): a load under a divergent
    // branch makes the compiler drain vmcnt at the join, which serialises the prefetch.
    const unsigned long long last = n_pairs - 1ull;
    unsigned long long base = start;
    uint64_t key_cur = pairs[min(base + lane, last)];
    uint64_t key_nxt = pairs[min(base + 64ull + lane, last)];
    bool in = (base + lane < n_pairs) && ((uint32_t)(key_cur >> F.seq_bits) == vox);
    RayDesc d = rays[ray_index(F, (uint32_t)key_cur & F.point_mask)];
    for (;;) {
      const int cnt = (int)__popcll(__ballot(in));  // sorted => the in-lanes form a prefix
      if (cnt == 0) break;
      const bool in_n = (base + 64ull + lane < n_pairs) && ((uint32_t)(key_nxt >> F.seq_bits) == vox);
      const RayDesc d_n = rays[ray_index(F, (uint32_t)key_nxt & F.point_mask)];
      const uint64_t key_nn = pairs[min(base + 128ull + lane, last)];

      // ---- per-lane, voxel-state-independent part: computeDistance + weight drop-off ----
      float sdf = 0.f, uw = 0.f;
      if (in) {
        const f3 v_point_origin = sub3({d.px, d.py, d.pz}, F.T.t);
        const float dist_G = norm3(v_point_origin);
        const float dist_G_V = dot3(v_voxel_origin, v_point_origin) / dist_G;
        sdf = dist_G - dist_G_V;
        uw = d.weight;
        if (Pm.use_dropoff && sdf < -Pm.voxel_size) {
          uw = d.weight * (Pm.trunc + sdf) / Pm.dropoff_denominator;
          uw = std_max(uw, 0.0f);
        }
        if (Pm.use_sparsity) {
          if (fabsf(sdf) < Pm.trunc) uw *= Pm.sparsity_factor;
        }
        const uint32_t kind = (d.info >> 8) & 3u;
        const uint32_t lab = d.info & 0xffu;
        if (kind == 2u) {
          const float* dl = deltas + (size_t)((uint32_t)key_cur & F.point_mask) * kNumLabels;
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) s_inc[lane][l] = dl[l];
        } else {
          const float a = (kind == 1u) ? d.d_match : 0.0f, b = (kind == 1u) ? d.d_non : 0.0f;
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) s_inc[lane][l] = ((uint32_t)l == lab) ? a : b;
        }
      }
      __builtin_amdgcn_wave_barrier();

      // ---- pass 1: the weight recurrence (independent of the distance) ----
      // w' = min(max_weight, w + uw) unless w + uw < 1e-6 (then the TSDF update is a no-op).
      float my_w = 0.0f, my_nw = 1.0f;
      if (weight == Pm.max_weight && __ballot(in && !(uw >= 0.0f)) == 0ull) {
        // Weight already clamped at max_weight and every increment is non-negative: each
        // update sees w = max_weight and leaves min(max_weight, max_weight + uw) = max_weight,
        // so the recurrence degenerates to 64 independent additions (the steady state of the
        // voxels next to the sensor, which are the long runs).
        my_w = weight;
        my_nw = weight + uw;
      } else {
        float w_run = weight;
        for (int k = 0; k < cnt; ++k) {
          const float nw = w_run + bcast_f(uw, k);  // lane broadcast: an LDS read here costs its full latency per step
          if (lane == k) { my_w = w_run; my_nw = nw; }
          if (!(nw < kEps)) w_run = std_min(Pm.max_weight, nw);
        }
        weight = w_run;
      }
      const bool my_skip = my_nw < kEps;
      const float my_r = 1.0f / my_nw;  // correctly rounded reciprocal, off the critical path
      const float my_p = sdf * uw;      // fl(sdf * uw)
      // Saturation: with dist == +trunc on entry, an update whose exact weighted mean exceeds
      // trunc by more than the rounding slack of the f32 operations leaves dist == +trunc (the
      // clamp).  If that holds for every update of the batch the distance recurrence is skipped.
      const bool my_sat = my_skip || ((sdf - Pm.trunc) * uw >= 1e-6f * Pm.trunc * my_nw);
      const bool all_sat = (__ballot(in && !my_sat) == 0ull);
      if (!(all_sat && dist == Pm.trunc && COLOR_MODE != KS_COLOR_MODE_COLOR)) {
        // ---- pass 2: the distance recurrence ----
        for (int k = 0; k < cnt; ++k) {
          if (bcast_u(my_skip ? 1u : 0u, k)) continue;
          const float w_k = bcast_f(my_w, k), nw_k = bcast_f(my_nw, k), r_k = bcast_f(my_r, k);
          const float num = bcast_f(my_p, k) + dist * w_k;
          const float q = div_by_recip(num, nw_k, r_k);
          if (COLOR_MODE == KS_COLOR_MODE_COLOR) {
            if (fabsf(bcast_f(sdf, k)) < Pm.trunc)
              color = blend_two_colors(color, w_k, bcast_u(d.color, k), bcast_f(uw, k));
          }
          dist = (q > 0.0f) ? std_min(Pm.trunc, q) : std_max(-Pm.trunc, q);
        }
      }
      // ---- pass 3: semantic log-likelihood, lane l owns class l; increments stream from LDS ----
      if (cnt == 64) {
        // full batch: all 64 increments are requested from LDS before the first dependent add
        // (this wave is alone on its SIMD: nothing else hides the LDS latency)
        float x[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) x[k] = s_inc[k][cls];
#pragma unroll
        for (int k = 0; k < 64; ++k) pri += x[k];
      } else {
#pragma unroll 8
        for (int k = 0; k < cnt; ++k) pri += s_inc[k][cls];
      }
      __builtin_amdgcn_wave_barrier();
      if (cnt < 64) break;
      d = d_n;
      in = in_n;
      key_cur = key_nxt;
      key_nxt = key_nn;
      base += 64;
    }
    // argmax over lanes 0..20, first strict maximum
    int best = 0;
    float m = bcast_f(pri, 0);
#pragma unroll
    for (int l = 1; l < kNumLabels; ++l) {
      const float x = bcast_f(pri, l);
      if (x > m) { m = x; best = l; }
    }
    if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = label_lut[best];
    else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY)
      color = rainbow_color_map((double)(float)exp((double)m));
    if (lane < kNumLabels) rec[4 + lane] = __float_as_uint(pri);
    if (lane == 0) *(uint4*)rec = make_uint4(__float_as_uint(dist), __float_as_uint(weight), color, (uint32_t)best);
    __builtin_amdgcn_wave_barrier();
  }
}


// ------------------------------------------------------------------------------------------
// Multi-GPU exchange (new functionality): tiles travel as raw 64 KiB records.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_export_tiles(Pool P, const uint32_t* __restrict__ slots, uint4* __restrict__ out) {
  const uint4* src = P.vox + (size_t)slots[blockIdx.x] * kTileVoxels * 8;
  uint4* dst = out + (size_t)blockIdx.x * kTileVoxels * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) dst[r * 512 + threadIdx.x] = src[r * 512 + threadIdx.x];
}

__global__ void __launch_bounds__(256) k_insert_tiles(TileTable T, Counters* C, const uint64_t* __restrict__ keys, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tile_insert(T, C, keys[i]);
}

// Merge one incoming tile per workgroup into the resident map; 8 lanes per voxel.
//   TSDF: Voxblox's layer-merge rule (mergeVoxelAIntoVoxelB): weight-averaged distance and
//         colour, summed weight (clamped to max_weight);
//   semantics: log-likelihoods are additive: priors += (incoming - initial), then argmax/colour
//         exactly as updateSemanticVoxel ends ([K:src/semantic_integrator_base.cpp:164-191]).
template <int COLOR_MODE>
__global__ void __launch_bounds__(512) k_merge_tiles(TileTable T, Pool P, const uint64_t* __restrict__ keys,
                                                     const uint4* __restrict__ in, float max_weight,
                                                     const uint32_t* __restrict__ label_lut) {
  const uint32_t slot = tile_lookup(T, keys[blockIdx.x]);
  if (slot == 0xffffffffu) return;
  const uint4* src = in + (size_t)blockIdx.x * kTileVoxels * 8;
  uint4* dst = P.vox + (size_t)slot * kTileVoxels * 8;
  const uint32_t lane = lane_id(), sub = lane & 7u;
  const uint32_t cbase = (sub - 1u) * 4u;
  for (uint32_t r = 0; r < 8; ++r) {
    const uint32_t q = r * 512u + threadIdx.x;  // uint4 index in the tile; voxel = q >> 3
    uint4 a = src[q];
    uint4 b = dst[q];
    // every lane of the voxel's group needs A's label (dword 3 of sub 0)
    const uint32_t a_label = perm_u(a.w, lane & ~7u);
    const bool touched = a_label != 255u;
    float bv = -INFINITY;
    uint32_t bi = 1000u;
    if (touched) {
      if (sub == 0u) {
        const float ad = __uint_as_float(a.x), aw = __uint_as_float(a.y);
        float bd = __uint_as_float(b.x), bw = __uint_as_float(b.y);
        const float cw = aw + bw;
        if (cw > 0.0f) {
          bd = (ad * aw + bd * bw) / cw;
          if (COLOR_MODE == KS_COLOR_MODE_COLOR) b.z = blend_two_colors(a.z, aw, b.z, bw);
          bw = std_min(max_weight, cw);
        }
        b.x = __float_as_uint(bd);
        b.y = __float_as_uint(bw);
      } else if (sub < 7u) {
        float p0 = __uint_as_float(b.x) + (__uint_as_float(a.x) - kPriorInit);
        float p1 = __uint_as_float(b.y), p2 = __uint_as_float(b.z), p3 = __uint_as_float(b.w);
        bv = p0; bi = cbase;
        if (sub < 6u) {
          p1 += __uint_as_float(a.y) - kPriorInit;
          p2 += __uint_as_float(a.z) - kPriorInit;
          p3 += __uint_as_float(a.w) - kPriorInit;
          if (p1 > bv) { bv = p1; bi = cbase + 1u; }
          if (p2 > bv) { bv = p2; bi = cbase + 2u; }
          if (p3 > bv) { bv = p3; bi = cbase + 3u; }
        }
        b = make_uint4(__float_as_uint(p0), sub < 6u ? __float_as_uint(p1) : 0u, sub < 6u ? __float_as_uint(p2) : 0u,
                       sub < 6u ? __float_as_uint(p3) : 0u);
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const float ov = perm_f(bv, lane ^ (uint32_t)o);
      const uint32_t oi = perm_u(bi, lane ^ (uint32_t)o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (touched && sub < 7u) {
      if (sub == 0u) {
        b.w = bi;
        if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) b.z = label_lut[bi];
        else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) b.z = rainbow_color_map((double)(float)exp((double)bv));
      }
      dst[q] = b;
    }
  }
}


// ------------------------------------------------------------------------------------------
// f-1: depth + label image -> camera-frame points on the GPU (replaces the XYZRGB cloud and the
// colour->label round trip).  Pinhole back-projection exactly as
// [KR:include/kimera_semantics_ros/depth_map_to_pointcloud.h:256-272]:
//   x = (u - cx) * depth * (unit/fx);  y = (v - cy) * depth * (unit/fy);  z = toMeters(depth)
// Invalid pixels (non-finite f32 / zero u16) are DROPPED with a stable compaction, as the
// Voxblox server drops non-finite points before integratePointCloud, so the
// point order — and with it the integration order — is that of the reference pipeline.
// ------------------------------------------------------------------------------------------
struct DepthParams {
  const void* depth;
  const uint8_t* label_img;   // u8 labels (preferred) or nullptr
  const uint8_t* rgba_img;    // rgba8 segmentation colours (used when label_img == nullptr) or nullptr
  int fmt;                    // 0 = f32 metres, 1 = u16 millimetres
  int width, height;
  float cx, cy, constant_x, constant_y;
};
__device__ __forceinline__ bool depth_pixel(const DepthParams& D, uint32_t i, float& x, float& y, float& z) {
  const int u = (int)(i % (uint32_t)D.width), v = (int)(i / (uint32_t)D.width);
  if (D.fmt == 0) {
    const float d = ((const float*)D.depth)[i];
    if (!isfinite(d)) return false;
    x = ((float)u - D.cx) * d * D.constant_x;
    y = ((float)v - D.cy) * d * D.constant_y;
    z = d;
  } else {
    const uint16_t d = ((const uint16_t*)D.depth)[i];
    if (d == 0) return false;
    x = ((float)u - D.cx) * (float)d * D.constant_x;
    y = ((float)v - D.cy) * (float)d * D.constant_y;
    z = (float)d * 0.001f;
  }
  return true;
}
__global__ void __launch_bounds__(1024) k_depth_count(DepthParams D, uint32_t n_px, uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float x, y, z;
  const bool ok = i < n_px && depth_pixel(D, i, x, y, z);
  const unsigned long long m = __ballot(ok);
  if (lane_id() == 0 && m) atomicAdd(&s_cnt, (uint32_t)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt;
}
// single workgroup: exclusive scan of the per-block counts (<= 4096 blocks), total in out[nb]
__global__ void __launch_bounds__(1024) k_depth_scan(uint32_t* __restrict__ counts, uint32_t nb) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? counts[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t add = s_carry;
    for (uint32_t w = 0; w < wave; ++w) add += s_wave[w];
    if (i < nb) counts[i] = add + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = add + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[nb] = s_carry;
}
__global__ void __launch_bounds__(1024) k_depth_compact(DepthParams D, uint32_t n_px, const uint32_t* __restrict__ block_off,
                                                        const uint32_t* __restrict__ label_lut, float* __restrict__ xyz,
                                                        uint8_t* __restrict__ rgba, uint8_t* __restrict__ labels) {
  __shared__ uint32_t s_wave[16];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  float x = 0.f, y = 0.f, z = 0.f;
  const bool ok = i < n_px && depth_pixel(D, i, x, y, z);
  const unsigned long long m = __ballot(ok);
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = block_off[blockIdx.x];
  for (uint32_t w = 0; w < wave; ++w) off += s_wave[w];
  if (ok) {
    const uint32_t o = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    xyz[3 * o] = x;
    xyz[3 * o + 1] = y;
    xyz[3 * o + 2] = z;
    if (D.label_img) {
      const uint32_t lab = D.label_img[i];
      labels[o] = (uint8_t)lab;
      ((uint32_t*)rgba)[o] = (label_lut[lab] & 0x00ffffffu) | 0xff000000u;  // cloud alpha is 255 (:269)
    } else if (D.rgba_img) {
      ((uint32_t*)rgba)[o] = (((const uint32_t*)D.rgba_img)[i] & 0x00ffffffu) | 0xff000000u;
    }
  }
}

// sorted integration order: key = bits of squared norm (non-negative float => monotone as u32)
__global__ void __launch_bounds__(256) k_sqnorm(uint32_t n, const float* __restrict__ xyz, uint32_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 p = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  keys[i] = __float_as_uint(dot3(p, p));
  vals[i] = i;
}

__global__ void __launch_bounds__(256) k_invert(uint32_t n, const uint32_t* __restrict__ order, uint32_t* __restrict__ inv) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) inv[order[p]] = p;
}

// Host-layout export: one lane per voxel of a requested host block (edge vps), AoS records.
__global__ void __launch_bounds__(256) k_download(TileTable T, Pool P, const int32_t* __restrict__ block_idx, int vps,
                                                  const uint32_t* __restrict__ label_lut, uint8_t* __restrict__ tsdf_out,
                                                  uint8_t* __restrict__ sem_out) {
  const uint32_t b = blockIdx.y;
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nv = (uint32_t)(vps * vps * vps);
  if (l >= nv) return;
  const int lx = (int)(l % (uint32_t)vps), ly = (int)((l / (uint32_t)vps) % (uint32_t)vps), lz = (int)(l / (uint32_t)(vps * vps));
  const int vx = block_idx[3 * b] * vps + lx, vy = block_idx[3 * b + 1] * vps + ly, vz = block_idx[3 * b + 2] * vps + lz;
  const uint32_t slot = tile_lookup(T, pack_tile(vx >> 3, vy >> 3, vz >> 3));
  float dist = 0.0f, weight = 0.0f;
  uint32_t color = 0, label = 255;
  float pri[kNumLabels];
#pragma unroll
  for (int k = 0; k < kNumLabels; ++k) pri[k] = kPriorInit;
  if (slot != 0xffffffffu) {
    const uint32_t local = (uint32_t)(vx & 7) + 8u * ((uint32_t)(vy & 7) + 8u * (uint32_t)(vz & 7));
    const uint4* rec = P.vox + ((size_t)slot * kTileVoxels + local) * 8;
    const uint4 q0 = rec[0];
    dist = __uint_as_float(q0.x);
    weight = __uint_as_float(q0.y);
    color = q0.z;
    label = q0.w;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      const uint4 q = rec[1 + g];
      pri[4 * g] = __uint_as_float(q.x);
      if (g < 5) {
        pri[4 * g + 1] = __uint_as_float(q.y);
        pri[4 * g + 2] = __uint_as_float(q.z);
        pri[4 * g + 3] = __uint_as_float(q.w);
      }
    }
  }
  const size_t o = (size_t)b * nv + l;
  if (tsdf_out) {
    uint32_t* t = (uint32_t*)(tsdf_out + o * 12);
    t[0] = __float_as_uint(dist);
    t[1] = __float_as_uint(weight);
    t[2] = color;
  }
  if (sem_out) {
    uint32_t* s = (uint32_t*)(sem_out + o * 92);
    const bool touched = label != 255u;
    s[0] = touched ? label : 0u;
#pragma unroll
    for (int k = 0; k < kNumLabels; ++k) s[1 + k] = __float_as_uint(pri[k]);
    // never-updated voxel: HashableColor::Gray() [K:include/kimera_semantics/semantic_voxel.h:26]
    s[22] = touched ? label_lut[label] : (127u | (127u << 8) | (127u << 16) | (255u << 24));
  }
}


// Host-layout import (the inverse of k_download): one lane per voxel of a host block.  A voxel
// that still looks default-constructed on the semantic side (label 0, Gray, initial priors:
// [K:include/kimera_semantics/semantic_voxel.h:14-27]) keeps the "never updated" marker.
__global__ void __launch_bounds__(256) k_upload(TileTable T, Pool P, const int32_t* __restrict__ block_idx, int vps,
                                                const uint8_t* __restrict__ tsdf_in, const uint8_t* __restrict__ sem_in) {
  const uint32_t b = blockIdx.y;
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nv = (uint32_t)(vps * vps * vps);
  if (l >= nv) return;
  const int lx = (int)(l % (uint32_t)vps), ly = (int)((l / (uint32_t)vps) % (uint32_t)vps), lz = (int)(l / (uint32_t)(vps * vps));
  const int vx = block_idx[3 * b] * vps + lx, vy = block_idx[3 * b + 1] * vps + ly, vz = block_idx[3 * b + 2] * vps + lz;
  const uint32_t slot = tile_lookup(T, pack_tile(vx >> 3, vy >> 3, vz >> 3));
  if (slot == 0xffffffffu) return;
  const uint32_t local = (uint32_t)(vx & 7) + 8u * ((uint32_t)(vy & 7) + 8u * (uint32_t)(vz & 7));
  uint32_t* rec = (uint32_t*)(P.vox + ((size_t)slot * kTileVoxels + local) * 8);
  const size_t o = (size_t)b * nv + l;
  if (tsdf_in) {
    const uint32_t* t = (const uint32_t*)(tsdf_in + o * 12);
    rec[0] = t[0];
    rec[1] = t[1];
    rec[2] = t[2];
  }
  if (sem_in) {
    const uint32_t* s = (const uint32_t*)(sem_in + o * 92);
    const uint32_t label = s[0] & 0xffu;
    bool pristine = label == 0u && s[22] == (127u | (127u << 8) | (127u << 16) | (255u << 24));
    for (int k = 0; k < kNumLabels; ++k) {
      const uint32_t p = s[1 + k];
      pristine = pristine && p == __float_as_uint(kPriorInit);
      rec[4 + k] = p;
    }
    rec[3] = pristine ? 255u : label;
  }
}


std::string g_create_error;

}  // namespace

// ==========================================================================================
// Host side of the C ABI
// ==========================================================================================
// A frame runs in two halves on the one stream:
//   front: points -> sort -> dedup / bundles -> march(+emit) -> 40-byte counter snapshot to the host
//   tail : init new tiles -> sort pairs -> apply           (sized by the snapshot)
// Everything a later stage reads from an earlier one lives in a FrameSlot.  With
// ks_config.pipeline_frames the three stages of a frame run on three streams,
//   A  points -> sort -> dedup / bundles          (stream)
//   B  ray march + snapshot                        (stream_march, after A of the same frame)
//   T  init tiles -> sort pairs -> apply           (stream_tail, enqueued by the NEXT call)
// so that A(i+1), B(i) and T(i-1) execute concurrently: the march is bound by device-scope atomic
// throughput, the sorts by dependent-launch latency, the voxel update by memory latency, and
// neither A nor B touches voxel data.  The host's one wait per frame (for the snapshot that sizes
// T) never idles the GPU.  Three slots rotate; stage A of a frame waits for the tail that last
// used its slot.
constexpr int kSlots = 3;
struct HostSnap {
  Counters c;
  uint32_t n_tiles;
  uint32_t pad[7];
};
struct FrameSlot {
  int index = 0;
  RayDesc* d_rays = nullptr;
  float* d_deltas = nullptr;        // merged: label histograms of mixed bundles
  uint64_t* d_pairs = nullptr;      // unsorted (voxel, ray) pairs written by k_march
  size_t cap_pairs_in = 0;
  Counters* d_counters = nullptr;   // inside ks_ctx::d_state
  uint32_t* d_ray_list = nullptr;   // rays to march (written by stage A, read by B)
  HostSnap* h_snap = nullptr;       // pinned + device-visible: written by k_publish at the end of B
  hipEvent_t a_done = nullptr;      // stage A complete
  hipEvent_t ready = nullptr;       // snapshot has landed
  hipEvent_t tail_done = nullptr;   // the tail has consumed this slot's buffers
  bool tail_recorded = false;
  FrameParams F{};
  size_t n = 0;
  int prof_set = -1;
  bool pending = false;
  const Counters& counters() const { return h_snap->c; }
  uint32_t n_tiles() const { return h_snap->n_tiles; }
};

// HIP-event sets for ks_profile: recorded in stream order, resolved lazily (before reuse or in
// ks_profile_get) so that profiling never adds a host wait to a frame.
constexpr int kProfSets = 4;
constexpr int kStageEvents = KS_STAGE_COUNT + 3;  // 0..3 stage A | 4,5 march begin/end | 6 tail begin, 7..10
struct ProfSet {
  hipEvent_t ev[kStageEvents]{};
  hipEvent_t k0 = nullptr, k1 = nullptr;  // begin/end of the k_apply dispatch itself
  bool used = false, complete = false, stages = false, apply = false, applied = false;
  uint64_t n_pairs = 0, n_points = 0;
};

struct ks_ctx {
  ks_config cfg{};
  std::string err;
  hipStream_t stream = nullptr;        // stage A (and everything else)
  hipStream_t stream_march = nullptr;  // stage B; == stream unless pipelined
  hipStream_t stream_tail = nullptr;   // stage T; == stream unless pipelined
  float voxel_size_inv = 0.f, log_match = 0.f, log_non_match = 0.f;
  int vps_shift = 1;  // log2(vps / 8)

  TileTable table{};
  Pool pool{};
  uint64_t* d_start_set = nullptr;
  uint64_t* d_observed_set = nullptr;
  uint64_t start_offset = 0, observed_offset = 0;
  int64_t reset_counter = 0;
  uint8_t* d_color_lut = nullptr;   // 16 MiB rgb -> label
  uint32_t* d_label_lut = nullptr;  // 256 label -> rgba
  uint32_t tiles_initialised = 0;

  // per-frame buffers
  size_t cap_points = 0;
  float* d_xyz = nullptr;
  uint8_t* d_rgba = nullptr;
  uint8_t* d_labels = nullptr;
  uint32_t* d_hash = nullptr;
  uint32_t *d_skeys32 = nullptr, *d_skeys32b = nullptr;
  float4* d_gpw = nullptr;
  uint64_t* d_ray_keys = nullptr;
  uint2* d_glc = nullptr;
  unsigned long long* d_long_list = nullptr;
  uint32_t* d_blong = nullptr;
  uint64_t *d_pkeys = nullptr, *d_pkeys2 = nullptr;
  uint32_t *d_pvals = nullptr, *d_pvals2 = nullptr;
  uint32_t* d_order = nullptr;
  uint32_t* d_inv_order = nullptr;
  uint32_t *d_okeys = nullptr, *d_okeys2 = nullptr, *d_ovals = nullptr;
  size_t cap_pairs = 0;
  uint64_t* d_pairs2 = nullptr;
  ksrs::Workspace sort_ws, sort_ws_tail;
  // Device words: Counters of slot k at 64 * k, the persistent tile count at 64 * kSlots.
  uint8_t* d_state = nullptr;
  FrameSlot slot[kSlots];
  uint64_t frame_no = 0;
  ks_frame_stats last_stats{};
  bool stats_undelivered = false;  // a query completed a pipelined frame: its statistics go to the next call
  int32_t* d_block_idx = nullptr;
  size_t cap_block_idx = 0;
  uint8_t *d_tsdf_out = nullptr, *d_sem_out = nullptr;
  size_t cap_out_blocks = 0;

  uint32_t* d_depth_blocks = nullptr;
  size_t cap_depth_blocks = 0;
  uint8_t* d_img_depth = nullptr;
  uint8_t* d_img_aux = nullptr;
  size_t cap_img_depth = 0, cap_img_aux = 0;

  int profiling = 0;  // 0 off, 1 all stages + every k_apply, 2 every 4th k_apply only
  ks_profile prof{};
  ProfSet pset[kProfSets];
  bool fatal = false;
};

#define HIPCHK(ctx, expr)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
      return KS_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

namespace {

template <typename T>
int dev_alloc(ks_ctx* c, T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIPCHK(c, hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
  return KS_OK;
}

int ensure_points(ks_ctx* c, size_t n) {
  if (n <= c->cap_points) return KS_OK;
  const size_t cap = std::max<size_t>(n, 1024);
  int rc;
  if ((rc = dev_alloc(c, &c->d_xyz, cap * 3))) return rc;
  if ((rc = dev_alloc(c, &c->d_rgba, cap * 4))) return rc;
  if ((rc = dev_alloc(c, &c->d_labels, cap))) return rc;
  for (int i = 0; i < (c->cfg.pipeline_frames ? kSlots : 1); ++i) {
    if ((rc = dev_alloc(c, &c->slot[i].d_rays, cap))) return rc;
    if ((rc = dev_alloc(c, &c->slot[i].d_ray_list, cap))) return rc;
    if (c->cfg.method == KS_METHOD_MERGED && (rc = dev_alloc(c, &c->slot[i].d_deltas, cap * kNumLabels))) return rc;
  }
  if ((rc = dev_alloc(c, &c->d_hash, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_skeys32, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_skeys32b, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pkeys, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pkeys2, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pvals, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pvals2, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_order, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_inv_order, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_okeys, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_okeys2, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_ovals, cap))) return rc;
  if (c->cfg.method == KS_METHOD_MERGED) {
    if ((rc = dev_alloc(c, &c->d_gpw, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_glc, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_ray_keys, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_blong, cap / kLongRun + 64))) return rc;
  }
  c->cap_points = cap;
  return KS_OK;
}

// d_pairs is written by k_march before the pair count is known: it is sized for the worst
// case (every ray at full length); d_pairs2 / the long-run list are sized by the actual count.
int ensure_pairs_in(ks_ctx* c, FrameSlot& S, size_t bound) {
  if (bound <= S.cap_pairs_in) return KS_OK;
  const size_t cap = std::max<size_t>(bound, 1 << 20);
  int rc;
  if ((rc = dev_alloc(c, &S.d_pairs, cap))) return rc;
  S.cap_pairs_in = cap;
  return KS_OK;
}
int ensure_pairs_out(ks_ctx* c, size_t n) {
  if (n <= c->cap_pairs) return KS_OK;
  const size_t cap = std::max<size_t>(n + n / 4, 1 << 20);
  int rc;
  if ((rc = dev_alloc(c, &c->d_pairs2, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_long_list, cap / kLongRun + 64))) return rc;
  c->cap_pairs = cap;
  return KS_OK;
}

template <typename K>
int sort_keys(ks_ctx* c, K* a, K* b, size_t n, unsigned end_bit, K** result, unsigned begin_bit = 0, bool tail = false) {
  HIPCHK(c, (ksrs::sort<K, false>(tail ? c->sort_ws_tail : c->sort_ws, a, b, nullptr, nullptr, n, end_bit,
                                  tail ? c->stream_tail : c->stream, result, nullptr, begin_bit)));
  return KS_OK;
}
template <typename K>
int sort_pairs(ks_ctx* c, K* ka, K* kb, uint32_t* va, uint32_t* vb, size_t n, unsigned end_bit, K** kres,
               uint32_t** vres) {
  HIPCHK(c, (ksrs::sort<K, true>(c->sort_ws, ka, kb, va, vb, n, end_bit, c->stream, kres, vres)));
  return KS_OK;
}

inline unsigned bits_for(uint64_t n) {  // number of bits needed to represent values < n
  unsigned b = 1;
  while (b < 64 && (1ull << b) < n) ++b;
  return b;
}

// ApproxHashSet::resetApproxSet
int reset_set(ks_ctx* c, uint64_t* d_set, uint64_t* offset) {
  if (++(*offset) >= kFullResetThreshold) {
    if (c->stream_march != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_march));  // the march reads the observed set
    HIPCHK(c, hipMemsetAsync(d_set, 0, sizeof(uint64_t) << kSetBits, c->stream));
    *offset = 0;
    const uint64_t poison = ~0ull;
    HIPCHK(c, hipMemcpyAsync(d_set, &poison, sizeof(poison), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return KS_OK;
}

inline void stage_mark(ks_ctx* c, int set, int ev) {
  if (set >= 0 && c->pset[set].stages)
    (void)hipEventRecord(c->pset[set].ev[ev], ev <= 3 ? c->stream : ev <= 5 ? c->stream_march : c->stream_tail);
}

// fold a finished event set into ks_profile
void resolve_prof(ks_ctx* c, int set) {
  ProfSet& P = c->pset[set];
  if (!P.used || !P.complete) return;
  (void)hipEventSynchronize(P.ev[kStageEvents - 1]);
  if (P.stages) {
    for (int s = 0; s < KS_STAGE_COUNT; ++s) {
      float ms = 0.f;
      const int a = s < 3 ? s : s == 3 ? 4 : s + 2;
      if (hipEventElapsedTime(&ms, P.ev[a], P.ev[a + 1]) == hipSuccess) {
        c->prof.ms[s] += ms;
        c->prof.launches[s] += 1;
      }
    }
  }
  if (P.applied) {
    float kms = 0.f;
    if (hipEventElapsedTime(&kms, P.k0, P.k1) == hipSuccess) {
      c->prof.apply_kernel_ms += kms;
      c->prof.apply_kernel_launches += 1;
      c->prof.apply_kernel_updates += P.n_pairs;
    }
  }
  c->prof.frames += 1;
  c->prof.updates += P.n_pairs;
  c->prof.points += P.n_points;
  P.used = P.complete = P.applied = false;
}

// ---- front half: everything up to the counter snapshot --------------------------------------
int frame_front(ks_ctx* c, FrameSlot& S, const float Tq[7], const float* d_xyz, const uint8_t* d_rgba,
                const uint8_t* d_labels, size_t n, int freespace) {
  const ks_config& cfg = c->cfg;
  int rc;
  FrameParams& F = S.F;
  F = FrameParams{};
  F.T.w = Tq[0];
  F.T.v = {Tq[1], Tq[2], Tq[3]};
  F.T.t = {Tq[4], Tq[5], Tq[6]};
  F.voxel_size_inv = c->voxel_size_inv;
  F.min_ray = cfg.min_ray_length_m;
  F.max_ray = cfg.max_ray_length_m;
  F.trunc = cfg.truncation_distance;
  F.start_inv = cfg.start_voxel_subsampling_factor * c->voxel_size_inv;
  F.log_match = c->log_match;
  F.log_non_match = c->log_non_match;
  F.tsdf.voxel_size = cfg.voxel_size;
  F.tsdf.trunc = cfg.truncation_distance;
  F.tsdf.max_weight = cfg.max_weight;
  F.tsdf.dropoff_denominator = cfg.truncation_distance - cfg.voxel_size;
  F.tsdf.sparsity_factor = cfg.sparsity_compensation_factor;
  F.tsdf.use_dropoff = cfg.use_weight_dropoff;
  F.tsdf.use_sparsity = cfg.use_sparsity_compensation_factor;
  F.start_offset = c->start_offset;
  F.observed_offset = c->observed_offset;
  F.max_collisions = cfg.max_consecutive_ray_collisions;
  F.n = (uint32_t)n;
  F.per_group = (uint32_t)(n / 1024);
  F.carving = cfg.voxel_carving_enabled;
  F.allow_clear = cfg.allow_clear;
  F.freespace = freespace;
  F.use_const_weight = cfg.use_const_weight;
  F.method = cfg.method;
  F.color_mode = cfg.color_mode;
  F.sorted_order = cfg.integration_order_mode == KS_ORDER_SORTED;
  F.n_dynamic = cfg.n_dynamic_labels;
  {
    const unsigned pb = bits_for(n);
    F.point_mask = (1u << pb) - 1u;
    F.clear_bit = (cfg.method == KS_METHOD_MERGED) ? (1u << pb) : 0u;
    F.seq_bits = pb + (cfg.method == KS_METHOD_MERGED ? 1u : 0u);
  }
  F.order = nullptr;  // set below once the sorted order has been computed
  F.inv_order = F.sorted_order ? c->d_inv_order : nullptr;
  std::memcpy(F.dynamic_labels, cfg.dynamic_labels, 32);
  // the early-out can never fire if the threshold exceeds the longest possible ray
  const double max_steps = 3.0 * ((double)cfg.max_ray_length_m + 2.0 * cfg.truncation_distance) * c->voxel_size_inv + 8.0;
  F.early_out = (cfg.method == KS_METHOD_FAST) && ((double)cfg.max_consecutive_ray_collisions < max_steps);

  // march (+emit) writes pairs before their count is known: worst case = every point a full-length ray
  {
    const double max_len = (double)cfg.max_ray_length_m + 2.0 * (double)cfg.truncation_distance;
    const size_t steps_max = (size_t)std::ceil(1.7321 * max_len * (double)c->voxel_size_inv) + 8;
    if ((rc = ensure_pairs_in(c, S, n * steps_max))) return rc;
  }

  hipStream_t st = c->stream;
  if (S.tail_recorded && c->stream_tail != c->stream) HIPCHK(c, hipStreamWaitEvent(st, S.tail_done, 0));
  S.n = n;
  S.prof_set = -1;
  if (c->profiling) {
    const int set = (int)(c->frame_no % kProfSets);
    resolve_prof(c, set);
    ProfSet& P = c->pset[set];
    P.used = true;
    P.complete = P.applied = false;
    P.stages = c->profiling == 1;
    P.apply = c->profiling == 1 || (c->frame_no % 4) == 0;
    P.n_points = n;
    P.n_pairs = 0;
    S.prof_set = set;
  }
  ++c->frame_no;
  // S.d_counters are zero: cleared at create time / by k_publish of the slot's previous frame

  const uint32_t nb = (uint32_t)((n + 255) / 256);
  const uint32_t nb1k = (uint32_t)((n + 1023) / 1024);
  const uint32_t* order_ptr = nullptr;
  stage_mark(c, S.prof_set, 0);

  if (F.sorted_order) {
    hipLaunchKernelGGL(k_sqnorm, dim3(nb), dim3(256), 0, st, (uint32_t)n, d_xyz, c->d_okeys, c->d_ovals);
    uint32_t *ok = nullptr, *ov = nullptr;
    if ((rc = sort_pairs(c, c->d_okeys, c->d_okeys2, c->d_ovals, c->d_order, n, 32, &ok, &ov))) return rc;
    order_ptr = ov;  // position -> index
    hipLaunchKernelGGL(k_invert, dim3(nb), dim3(256), 0, st, (uint32_t)n, order_ptr, c->d_inv_order);
    F.order = order_ptr;
  }

  if (cfg.method == KS_METHOD_FAST) {
    hipLaunchKernelGGL(k_points_fast, dim3(nb1k), dim3(1024), 0, st, F, d_xyz, d_rgba, d_labels, c->d_color_lut,
                       S.d_rays, c->d_hash, c->d_skeys32, c->d_pvals, S.d_counters);
    stage_mark(c, S.prof_set, 1);
    // stable sort by slot only: position order inside a slot is preserved
    uint32_t *sk = nullptr, *sv = nullptr;
    if ((rc = sort_pairs(c, c->d_skeys32, c->d_skeys32b, c->d_pvals, c->d_pvals2, n, kSetBits + 1, &sk, &sv))) return rc;
    stage_mark(c, S.prof_set, 2);
    hipLaunchKernelGGL(k_dedup, dim3(nb1k), dim3(1024), 0, st, F, sk, sv, c->d_hash, c->d_start_set, S.d_ray_list,
                       S.d_counters);
    hipLaunchKernelGGL(k_dedup_commit, dim3(nb1k), dim3(1024), 0, st, F, sk, sv, c->d_hash, c->d_start_set,
                       S.d_counters);
  } else {
    hipLaunchKernelGGL(k_points_merged, dim3(nb1k), dim3(1024), 0, st, F, d_xyz, d_rgba, d_labels, c->d_color_lut,
                       c->d_pkeys, c->d_pvals, S.d_counters);
    stage_mark(c, S.prof_set, 1);
    uint64_t* sk = nullptr;
    uint32_t* sv = nullptr;
    if ((rc = sort_pairs(c, c->d_pkeys, c->d_pkeys2, c->d_pvals, c->d_pvals2, n, 64, &sk, &sv))) return rc;
    stage_mark(c, S.prof_set, 2);
    hipLaunchKernelGGL(k_gather_sorted, dim3(nb), dim3(256), 0, st, F, d_xyz, d_rgba, d_labels, c->d_color_lut,
                       order_ptr, sk, sv, c->d_gpw, c->d_glc);
    uint64_t* ray_keys = cfg.enable_anti_grazing ? c->d_ray_keys : nullptr;
    hipLaunchKernelGGL(k_bundles, dim3(nb), dim3(256), 0, st, F, sk, sv, c->d_gpw, c->d_glc, S.d_rays, S.d_deltas,
                       S.d_ray_list, c->d_blong, ray_keys, S.d_counters);
    hipLaunchKernelGGL(k_bundles_long, dim3((uint32_t)std::min<size_t>(n / kLongRun + 1, 2048)), dim3(64), 0, st, F,
                       sk, sv, c->d_gpw, c->d_glc, S.d_rays, S.d_deltas, S.d_ray_list, c->d_blong, ray_keys,
                       S.d_counters);
    if (cfg.enable_anti_grazing) {
      F.grazing_keys = sk;
      F.ray_keys = c->d_ray_keys;
    }
  }
  stage_mark(c, S.prof_set, 3);
  // ---- stage B: march (+emit) over an upper bound of rays (<= n); the live ray count stays on
  // the device.  Anti-grazing reads stage A's sorted point keys, which the next frame's stage A
  // overwrites: with it the march stays on stage A's stream.
  hipStream_t sm = cfg.enable_anti_grazing ? c->stream : c->stream_march;
  if (sm != st) {
    HIPCHK(c, hipEventRecord(S.a_done, st));
    HIPCHK(c, hipStreamWaitEvent(sm, S.a_done, 0));
  }
  if (S.prof_set >= 0 && c->pset[S.prof_set].stages) (void)hipEventRecord(c->pset[S.prof_set].ev[4], sm);
  hipLaunchKernelGGL(k_march, dim3(nb), dim3(256), 0, sm, F, S.d_ray_list, S.d_rays, c->table, c->pool,
                     c->d_observed_set, S.d_pairs, (unsigned long long)S.cap_pairs_in, S.d_counters);
  // the frame's only device->host traffic: pair / ray / tile counts and error flags
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, sm, S.d_counters, (const uint32_t*)c->table.n_tiles,
                     (uint32_t*)S.h_snap);
  HIPCHK(c, hipEventRecord(S.ready, sm));
  if (S.prof_set >= 0 && c->pset[S.prof_set].stages) (void)hipEventRecord(c->pset[S.prof_set].ev[5], sm);
  S.pending = true;
  return KS_OK;
}

// ---- tail half: sized by the snapshot --------------------------------------------------------
int frame_tail(ks_ctx* c, FrameSlot& S, ks_frame_stats* stats) {
  if (!S.pending) return KS_OK;
  S.pending = false;
  hipStream_t st = c->stream_tail;  // the host wait below orders the tail after the slot's front
  const FrameParams& F = S.F;
  {
    const auto w0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipEventSynchronize(S.ready));  // the frame's only host wait
    if (c->profiling) c->prof.host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }
  const Counters cnt = S.counters();
  const uint32_t new_tiles = std::min(S.n_tiles(), c->cfg.max_tiles);
  const uint32_t tiles_before = c->tiles_initialised;
  const int set = S.prof_set;
  stage_mark(c, set, 6);
  // tiles allocated by the front exist in the table whatever happens next: make them valid
  if (new_tiles > c->tiles_initialised) {
    hipLaunchKernelGGL(k_init_tiles, dim3(new_tiles - c->tiles_initialised), dim3(512), 0, st, c->pool,
                       c->tiles_initialised);
    c->tiles_initialised = new_tiles;
  }
  auto finish_prof = [&](uint64_t n_pairs) {
    if (set < 0) return;
    ProfSet& P = c->pset[set];
    P.n_pairs = n_pairs;
    (void)hipEventRecord(P.ev[kStageEvents - 1], st);
    P.complete = true;
  };
  if (cnt.err & kErrLabel) {
    for (int e = 7; e < kStageEvents - 1; ++e) stage_mark(c, set, e);
    finish_prof(0);
    c->err = "semantic label >= 21 (CHECK_LT in the reference)";
    return KS_ERR_LABEL_RANGE;
  }
  if (cnt.err) {
    for (int e = 7; e < kStageEvents - 1; ++e) stage_mark(c, set, e);
    finish_prof(0);
    c->fatal = true;
    if (cnt.err & kErrPool) {
      c->err = "voxel tile pool exhausted: raise ks_config.max_tiles";
      return KS_ERR_POOL_FULL;
    }
    c->err = "voxel index out of the packed range / tile table full";
    return KS_ERR_INDEX_RANGE;
  }
  const unsigned long long n_pairs = cnt.n_pairs;
  if (n_pairs > 0) {
    int rc;
    if ((rc = ensure_pairs_out(c, n_pairs))) return rc;
    stage_mark(c, set, 7);
    const unsigned end_bit = F.seq_bits + 9 + bits_for(new_tiles);
    uint64_t* sp = nullptr;
    // Deterministic modes sort by (voxel, ray sequence): every voxel replays its updates in
    // the reference's single-thread order.  With the racy early-out of `fast` the set of
    // updates is already schedule dependent (as in the multi-threaded reference, whose per-
    // voxel order is whatever the mutex grants), so grouping by voxel suffices: the sort
    // skips the sequence bits (2 fewer passes); the order inside a voxel is then the stable
    // emission order.
    const unsigned begin_bit = F.early_out ? F.seq_bits : 0u;
    if ((rc = sort_keys(c, S.d_pairs, c->d_pairs2, n_pairs, std::min(64u, end_bit), &sp, begin_bit, /*tail=*/true))) return rc;
    stage_mark(c, set, 8);
    const uint32_t ab = (uint32_t)((n_pairs + 255) / 256);
    const uint32_t lb = (uint32_t)std::min<unsigned long long>(n_pairs / kLongRun + 1, 4096);
    const bool time_apply = set >= 0 && c->pset[set].apply;
    if (time_apply) c->pset[set].applied = true;
#define KS_LAUNCH_APPLY(MODE)                                                                                        \
  if (time_apply)                                                                                                    \
    hipExtLaunchKernelGGL(k_apply<MODE>, dim3(ab), dim3(256), 0, st, c->pset[set].k0, c->pset[set].k1, 0, F, n_pairs, \
                          sp, S.d_rays, S.d_deltas, c->table, c->pool, c->d_label_lut, c->d_long_list, S.d_counters); \
  else                                                                                                               \
    hipLaunchKernelGGL(k_apply<MODE>, dim3(ab), dim3(256), 0, st, F, n_pairs, sp, S.d_rays, S.d_deltas, c->table,     \
                       c->pool, c->d_label_lut, c->d_long_list, S.d_counters);                                        \
  stage_mark(c, set, 9);                                                                                             \
  hipLaunchKernelGGL(k_apply_long<MODE>, dim3(lb), dim3(64), 0, st, F, n_pairs, sp, S.d_rays, S.d_deltas, c->table,   \
                     c->pool, c->d_label_lut, c->d_long_list, S.d_counters)
    switch (c->cfg.color_mode) {
      case KS_COLOR_MODE_COLOR: KS_LAUNCH_APPLY(KS_COLOR_MODE_COLOR); break;
      case KS_COLOR_MODE_SEMANTIC: KS_LAUNCH_APPLY(KS_COLOR_MODE_SEMANTIC); break;
      default: KS_LAUNCH_APPLY(KS_COLOR_MODE_SEMANTIC_PROBABILITY); break;
    }
#undef KS_LAUNCH_APPLY
  } else {
    stage_mark(c, set, 7);
    stage_mark(c, set, 8);
    stage_mark(c, set, 9);
  }
  finish_prof(n_pairs);
  HIPCHK(c, hipEventRecord(S.tail_done, st));
  S.tail_recorded = true;
  HIPCHK(c, hipGetLastError());
  c->last_stats = ks_frame_stats{};
  c->last_stats.n_points = S.n;
  c->last_stats.n_valid_points = cnt.n_valid;
  c->last_stats.n_rays_cast = cnt.n_rays;
  c->last_stats.n_voxel_updates = n_pairs;
  c->last_stats.n_blocks_allocated = new_tiles - tiles_before;
  c->stats_undelivered = stats == nullptr;
  if (stats) *stats = c->last_stats;
  return KS_OK;
}

// run the tail of a frame whose front is still waiting for it (pipelined mode)
int flush_pending(ks_ctx* c, ks_frame_stats* stats) {
  // at most one slot is pending between calls; the older frame first in any case
  for (int k = 0; k < kSlots; ++k) {
    FrameSlot& S = c->slot[(c->frame_no + k) % kSlots];
    if (S.pending) {
      const int rc = frame_tail(c, S, stats);
      if (rc) return rc;
    }
  }
  return KS_OK;
}

// complete every outstanding frame and drain both streams
int quiesce(ks_ctx* c) {
  const int rc = flush_pending(c, nullptr);
  if (c->stream_tail != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_tail));
  if (c->stream_march != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_march));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return rc;
}

int integrate_device_impl(ks_ctx* c, const float Tq[7], const float* d_xyz, const uint8_t* d_rgba, const uint8_t* d_labels,
                          size_t n, int freespace, ks_frame_stats* stats) {
  if (c->fatal) {
    c->err = "context is in a failed state (earlier pool/index error)";
    return KS_ERR_INVALID_ARG;
  }
  if (n >= (1u << 23)) {
    c->err = "more than 2^23-1 points per call";
    return KS_ERR_INVALID_ARG;
  }
  const ks_config& cfg = c->cfg;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  // sorted integration order keeps its permutation in single buffers: not pipelined
  const bool pipelined = cfg.pipeline_frames && cfg.integration_order_mode != KS_ORDER_SORTED;
  int rc;

  // frame-level bookkeeping of the fast integrator [K:src/semantic_tsdf_integrator_fast.cpp:165-170]
  if (cfg.method == KS_METHOD_FAST) {
    if ((++c->reset_counter) >= cfg.clear_checks_every_n_frames) {
      c->reset_counter = 0;
      if ((rc = reset_set(c, c->d_start_set, &c->start_offset))) return rc;
      if ((rc = reset_set(c, c->d_observed_set, &c->observed_offset))) return rc;
    }
  }
  if (n == 0) {
    if ((rc = quiesce(c))) return rc;
    if (stats) stats->n_points = 0;
    return KS_OK;
  }
  if (n > c->cap_points) {  // growing frees buffers a pending tail still needs
    if ((rc = quiesce(c))) return rc;
    if ((rc = ensure_points(c, n))) return rc;
  }
  if (!pipelined) {
    if ((rc = quiesce(c))) return rc;
    FrameSlot& S = c->slot[0];
    if ((rc = frame_front(c, S, Tq, d_xyz, d_rgba, d_labels, n, freespace))) return rc;
    return frame_tail(c, S, stats);
  }
  // pipelined: front of this frame first, then the tail of the previous one; the statistics
  // returned are those of the frame whose tail ran here (the previous frame)
  FrameSlot& S = c->slot[c->frame_no % kSlots];
  FrameSlot& prev = c->slot[(c->frame_no + kSlots - 1) % kSlots];
  if (S.pending && (rc = frame_tail(c, S, nullptr))) return rc;  // cannot happen: slots alternate
  if ((rc = frame_front(c, S, Tq, d_xyz, d_rgba, d_labels, n, freespace))) return rc;
  if (prev.pending) return frame_tail(c, prev, stats);
  if (c->stats_undelivered && stats) {
    *stats = c->last_stats;
    c->stats_undelivered = false;
  }
  return KS_OK;
}

int integrate_device(ks_ctx* c, const float Tq[7], const float* d_xyz, const uint8_t* d_rgba, const uint8_t* d_labels,
                     size_t n, int freespace, ks_frame_stats* stats) {
  if (!c->profiling) return integrate_device_impl(c, Tq, d_xyz, d_rgba, d_labels, n, freespace, stats);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = integrate_device_impl(c, Tq, d_xyz, d_rgba, d_labels, n, freespace, stats);
  c->prof.host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int collect_block_indices(ks_ctx* c, bool only_updated, bool reset, std::vector<int32_t>* out) {
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint32_t nt = c->tiles_initialised;
  std::vector<uint64_t> keys(nt);
  std::vector<uint8_t> upd(nt);
  if (nt) {
    HIPCHK(c, hipMemcpy(keys.data(), c->table.slot_keys, nt * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(upd.data(), c->pool.updated, nt, hipMemcpyDeviceToHost));
    if (only_updated && reset) HIPCHK(c, hipMemset(c->pool.updated, 0, nt));
  }
  std::set<std::tuple<int32_t, int32_t, int32_t>> s;
  for (uint32_t i = 0; i < nt; ++i) {
    if (only_updated && !upd[i]) continue;
    const uint64_t k = keys[i];
    const int tx = (int)((k >> 36) & 0x3ffffu) - kTileBias, ty = (int)((k >> 18) & 0x3ffffu) - kTileBias,
              tz = (int)(k & 0x3ffffu) - kTileBias;
    s.insert({tx >> c->vps_shift, ty >> c->vps_shift, tz >> c->vps_shift});
  }
  out->clear();
  for (const auto& t : s) {
    out->push_back(std::get<0>(t));
    out->push_back(std::get<1>(t));
    out->push_back(std::get<2>(t));
  }
  return KS_OK;
}

}  // namespace

// find-or-insert n tile keys (device array) and initialise the newly allocated tiles
static int insert_tiles(ks_ctx* c, const uint64_t* d_keys, size_t n) {
  int rc;
  if ((rc = quiesce(c))) return rc;
  FrameSlot& S = c->slot[0];
  HIPCHK(c, hipMemsetAsync(S.d_counters, 0, sizeof(Counters), c->stream));
  hipLaunchKernelGGL(k_insert_tiles, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, c->table, S.d_counters,
                     d_keys, (uint32_t)n);
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, S.d_counters, (const uint32_t*)c->table.n_tiles,
                     (uint32_t*)S.h_snap);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint32_t new_tiles = std::min(S.n_tiles(), c->cfg.max_tiles);
  if (new_tiles > c->tiles_initialised) {
    hipLaunchKernelGGL(k_init_tiles, dim3(new_tiles - c->tiles_initialised), dim3(512), 0, c->stream, c->pool,
                       c->tiles_initialised);
    c->tiles_initialised = new_tiles;
  }
  if (S.counters().err) {
    c->fatal = true;
    c->err = "voxel tile pool exhausted: raise ks_config.max_tiles";
    return KS_ERR_POOL_FULL;
  }
  return KS_OK;
}

extern "C" {

int ks_default_config(ks_config* c) {
  if (!c) return KS_ERR_INVALID_ARG;
  std::memset(c, 0, sizeof(*c));
  c->voxel_size = 0.05f;
  c->voxels_per_side = 16;
  c->truncation_distance = 4 * 0.05f;
  c->max_weight = 10000.0f;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->voxel_carving_enabled = 1;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->enable_anti_grazing = 0;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->integration_order_mode = KS_ORDER_MIXED;
  c->integrator_threads = 1;
  c->method = KS_METHOD_FAST;
  c->bundle_order = 1;
  c->semantic_measurement_probability = 0.9f;
  c->color_mode = KS_COLOR_MODE_SEMANTIC;
  c->n_dynamic_labels = 0;
  c->device_id = 0;
  c->max_tiles = 1u << 16;
  c->max_points = 1u << 20;
  return KS_OK;
}

int ks_create(const ks_config* cfg, ks_ctx** out) {
  if (!cfg || !out) {
    g_create_error = "null argument";
    return KS_ERR_INVALID_ARG;
  }
  const int vps = cfg->voxels_per_side;
  if (!(vps == 8 || vps == 16 || vps == 32 || vps == 64)) {
    g_create_error = "voxels_per_side must be 8, 16, 32 or 64";
    return KS_ERR_INVALID_ARG;
  }
  if (cfg->n_dynamic_labels < 0 || cfg->n_dynamic_labels > 32 || cfg->max_tiles == 0 || cfg->max_tiles >= (1u << 23)) {
    g_create_error = "bad n_dynamic_labels / max_tiles";
    return KS_ERR_INVALID_ARG;
  }
  // setSemanticProbabilities CHECKs [K:src/semantic_integrator_base.cpp:93-107]
  const float match = cfg->semantic_measurement_probability;
  const float non_match = 1.0f - cfg->semantic_measurement_probability;
  if (!(match > 0.0f) || !(non_match > 0.0f) || !(match < 1.0f) || !(non_match < 1.0f)) {
    g_create_error = "semantic_measurement_probability must be in (0,1)";
    return KS_ERR_PROBABILITY;
  }
  const float lm = std::log(match), lnm = std::log(non_match);  // host libm, as the reference
  if (!(lm > lnm)) {
    g_create_error = "log(p) must exceed log(1-p)";
    return KS_ERR_PROBABILITY;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device_id >= ndev) {
    g_create_error = "no HIP device (the MI355X path has no CPU fallback)";
    return KS_ERR_NO_DEVICE;
  }
  ks_ctx* c = new ks_ctx();
  c->cfg = *cfg;
  c->log_match = lm;
  c->log_non_match = lnm;
  c->voxel_size_inv = (float)(1.0 / cfg->voxel_size);  // TsdfIntegratorBase::setLayer
  c->vps_shift = vps == 8 ? 0 : vps == 16 ? 1 : vps == 32 ? 2 : 3;
#define CRCHK(expr)                                                        \
  do {                                                                     \
    hipError_t e_ = (expr);                                                \
    if (e_ != hipSuccess) {                                                \
      g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_);  \
      ks_destroy(c);                                                       \
      return KS_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)
  CRCHK(hipSetDevice(cfg->device_id));
  CRCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  if (cfg->pipeline_frames) {
    CRCHK(hipStreamCreateWithFlags(&c->stream_march, hipStreamNonBlocking));
    CRCHK(hipStreamCreateWithFlags(&c->stream_tail, hipStreamNonBlocking));
  } else {
    c->stream_march = c->stream_tail = c->stream;
  }
  for (auto& P : c->pset) {
    for (auto& e : P.ev) CRCHK(hipEventCreate(&e));
    CRCHK(hipEventCreate(&P.k0));
    CRCHK(hipEventCreate(&P.k1));
  }
  uint32_t cap = 1024;
  while (cap < 2u * cfg->max_tiles) cap <<= 1;
  c->table.mask = cap - 1;
  c->table.max_tiles = cfg->max_tiles;
  const size_t mt = cfg->max_tiles;
  CRCHK(hipMalloc((void**)&c->table.ent, cap * sizeof(TileEntry)));
  CRCHK(hipMalloc((void**)&c->table.slot_keys, mt * sizeof(uint64_t)));
  CRCHK(hipMemset(c->table.ent, 0xff, cap * sizeof(TileEntry)));  // key = empty, val = kSlotPending
  CRCHK(hipMalloc((void**)&c->pool.vox, mt * kTileVoxels * 8 * sizeof(uint4)));
  CRCHK(hipMalloc((void**)&c->pool.updated, mt));
  CRCHK(hipMemset(c->pool.updated, 0, mt));
  CRCHK(hipMalloc((void**)&c->d_start_set, sizeof(uint64_t) << kSetBits));
  CRCHK(hipMalloc((void**)&c->d_observed_set, sizeof(uint64_t) << kSetBits));
  CRCHK(hipMemset(c->d_start_set, 0, sizeof(uint64_t) << kSetBits));
  CRCHK(hipMemset(c->d_observed_set, 0, sizeof(uint64_t) << kSetBits));
  const uint64_t poison = ~0ull;  // ApproxHashSet ctor: slot[offset_=0] = SIZE_MAX
  CRCHK(hipMemcpy(c->d_start_set, &poison, 8, hipMemcpyHostToDevice));
  CRCHK(hipMemcpy(c->d_observed_set, &poison, 8, hipMemcpyHostToDevice));
  CRCHK(hipMalloc((void**)&c->d_label_lut, 256 * sizeof(uint32_t)));
  CRCHK(hipMemcpy(c->d_label_lut, cfg->label_rgba, 1024, hipMemcpyHostToDevice));
  static_assert(sizeof(Counters) == 32, "snapshot layout");
  CRCHK(hipMalloc((void**)&c->d_state, 64 * (kSlots + 1)));
  CRCHK(hipMemset(c->d_state, 0, 64 * (kSlots + 1)));
  c->table.n_tiles = (uint32_t*)(c->d_state + 64 * kSlots);
  for (int i = 0; i < kSlots; ++i) {
    FrameSlot& S = c->slot[i];
    S.index = i;
    S.d_counters = (Counters*)(c->d_state + 64 * i);
    CRCHK(hipHostMalloc((void**)&S.h_snap, sizeof(HostSnap)));
    std::memset(S.h_snap, 0, sizeof(HostSnap));
    CRCHK(hipEventCreateWithFlags(&S.a_done, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.ready, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.tail_done, hipEventDisableTiming));
  }
#undef CRCHK
  if (ensure_points(c, cfg->max_points) != KS_OK) {
    g_create_error = c->err;
    ks_destroy(c);
    return KS_ERR_HIP;
  }
  *out = c;
  return KS_OK;
}

void ks_destroy(ks_ctx* c) {
  if (!c) return;
  if (c->stream_tail && c->stream_tail != c->stream) (void)hipStreamSynchronize(c->stream_tail);
  if (c->stream_march && c->stream_march != c->stream) (void)hipStreamSynchronize(c->stream_march);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->table.ent, c->table.slot_keys, c->pool.vox, c->pool.updated, c->d_start_set, c->d_observed_set, c->d_color_lut,
                  c->d_label_lut, c->d_xyz, c->d_rgba, c->d_labels, c->slot[0].d_rays, c->slot[1].d_rays, c->slot[2].d_rays, c->slot[0].d_deltas, c->slot[1].d_deltas, c->slot[2].d_deltas, c->slot[0].d_ray_list, c->slot[1].d_ray_list, c->slot[2].d_ray_list, c->d_hash, c->d_skeys32, c->d_skeys32b, c->d_gpw, c->d_glc, c->d_ray_keys, c->d_long_list, c->d_blong, c->d_pkeys,
                  c->d_pkeys2, c->d_pvals, c->d_pvals2, c->d_order, c->d_inv_order, c->d_okeys, c->d_okeys2, c->d_ovals,
                  c->slot[0].d_pairs, c->slot[1].d_pairs, c->slot[2].d_pairs, c->d_pairs2, c->d_state,
                  c->d_block_idx, c->d_tsdf_out, c->d_sem_out, c->d_depth_blocks, c->d_img_depth, c->d_img_aux};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  ksrs::release(c->sort_ws);
  ksrs::release(c->sort_ws_tail);
  for (auto& S : c->slot) {
    if (S.h_snap) (void)hipHostFree(S.h_snap);
    if (S.ready) (void)hipEventDestroy(S.ready);
    if (S.tail_done) (void)hipEventDestroy(S.tail_done);
    if (S.a_done) (void)hipEventDestroy(S.a_done);
  }
  for (auto& P : c->pset) {
    for (auto& e : P.ev)
      if (e) (void)hipEventDestroy(e);
    if (P.k0) (void)hipEventDestroy(P.k0);
    if (P.k1) (void)hipEventDestroy(P.k1);
  }
  if (c->stream_tail && c->stream_tail != c->stream) (void)hipStreamDestroy(c->stream_tail);
  if (c->stream_march && c->stream_march != c->stream) (void)hipStreamDestroy(c->stream_march);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* ks_last_error(ks_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ks_set_color_to_label(ks_ctx* c, const uint8_t* rgba_keys, const uint8_t* labels, size_t n) {
  if (!c || (n && (!rgba_keys || !labels))) return KS_ERR_INVALID_ARG;
  // 16 MiB direct-mapped rgb -> label table; lookups force alpha = 255
  // ([K:src/semantic_tsdf_integrator_fast.cpp:157]), so only keys with alpha 255 can match
  // (HashableColor::operator== compares alpha, [K:src/color.cpp:25-27]); unknown -> 0.
  std::vector<uint8_t> lut(1u << 24, 0);
  for (size_t i = 0; i < n; ++i) {
    if (rgba_keys[4 * i + 3] != 255) continue;
    const uint32_t rgb = rgba_keys[4 * i] | (rgba_keys[4 * i + 1] << 8) | (rgba_keys[4 * i + 2] << 16);
    lut[rgb] = labels[i];
  }
  if (int rc = quiesce(c)) return rc;  // frames in flight still read the old table
  if (!c->d_color_lut) HIPCHK(c, hipMalloc((void**)&c->d_color_lut, 1u << 24));
  HIPCHK(c, hipMemcpy(c->d_color_lut, lut.data(), 1u << 24, hipMemcpyHostToDevice));
  return KS_OK;
}

int ks_integrate_points_device(ks_ctx* c, const float T[7], const float* d_xyz, const uint8_t* d_rgba,
                               const uint8_t* d_labels, size_t n, int freespace, ks_frame_stats* stats) {
  if (!c || !T || (n && !d_xyz)) return KS_ERR_INVALID_ARG;
  if (!d_labels && !(d_rgba && c->d_color_lut)) {
    c->err = "labels == NULL requires rgba and a colour map (ks_set_color_to_label)";
    return KS_ERR_INVALID_ARG;
  }
  return integrate_device(c, T, d_xyz, d_rgba, d_labels, n, freespace, stats);
}

int ks_integrate_points(ks_ctx* c, const float T[7], const float* xyz, const uint8_t* rgba, const uint8_t* labels,
                        size_t n, int freespace, ks_frame_stats* stats) {
  if (!c || !T || (n && !xyz)) return KS_ERR_INVALID_ARG;
  if (!labels && !(rgba && c->d_color_lut)) {
    c->err = "labels == NULL requires rgba and a colour map (ks_set_color_to_label)";
    return KS_ERR_INVALID_ARG;
  }
  int rc = ensure_points(c, n);
  if (rc) return rc;
  if (n) {
    HIPCHK(c, hipMemcpyAsync(c->d_xyz, xyz, n * 12, hipMemcpyHostToDevice, c->stream));
    if (rgba) HIPCHK(c, hipMemcpyAsync(c->d_rgba, rgba, n * 4, hipMemcpyHostToDevice, c->stream));
    if (labels) HIPCHK(c, hipMemcpyAsync(c->d_labels, labels, n, hipMemcpyHostToDevice, c->stream));
  }
  return integrate_device(c, T, c->d_xyz, rgba ? c->d_rgba : nullptr, labels ? c->d_labels : nullptr, n, freespace, stats);
}

static int integrate_depth_impl(ks_ctx* c, const float T[7], DepthParams D, int freespace, ks_frame_stats* stats) {
  const size_t n_px = (size_t)D.width * D.height;
  int rc = ensure_points(c, n_px);
  if (rc) return rc;
  const uint32_t nb = (uint32_t)((n_px + 1023) / 1024);
  if (nb + 1 > c->cap_depth_blocks) {
    if ((rc = dev_alloc(c, &c->d_depth_blocks, (size_t)nb + 1))) return rc;
    c->cap_depth_blocks = nb + 1;
  }
  hipStream_t st = c->stream;
  hipLaunchKernelGGL(k_depth_count, dim3(nb), dim3(1024), 0, st, D, (uint32_t)n_px, c->d_depth_blocks);
  hipLaunchKernelGGL(k_depth_scan, dim3(1), dim3(1024), 0, st, c->d_depth_blocks, nb);
  const bool have_labels = D.label_img != nullptr;
  hipLaunchKernelGGL(k_depth_compact, dim3(nb), dim3(1024), 0, st, D, (uint32_t)n_px, c->d_depth_blocks, c->d_label_lut,
                     c->d_xyz, c->d_rgba, c->d_labels);
  uint32_t n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->d_depth_blocks + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (!have_labels && !(D.rgba_img && c->d_color_lut)) {
    c->err = "ks_integrate_depth: need a label image, or a colour image plus ks_set_color_to_label";
    return KS_ERR_INVALID_ARG;
  }
  return integrate_device(c, T, c->d_xyz, c->d_rgba, have_labels ? c->d_labels : nullptr, n, freespace, stats);
}

int ks_integrate_depth(ks_ctx* c, const float T[7], const void* depth, int depth_fmt, const uint8_t* label_img,
                       const uint8_t* rgba_img, int width, int height, const float K[4], int freespace,
                       ks_frame_stats* stats) {
  if (!c || !T || !depth || !K || width <= 0 || height <= 0 || (depth_fmt != 0 && depth_fmt != 1)) return KS_ERR_INVALID_ARG;
  const size_t n_px = (size_t)width * height;
  const size_t dbytes = n_px * (depth_fmt == 0 ? 4 : 2);
  if (dbytes > c->cap_img_depth) {
    int rc = dev_alloc(c, &c->d_img_depth, dbytes);
    if (rc) return rc;
    c->cap_img_depth = dbytes;
  }
  if (n_px * 4 > c->cap_img_aux) {
    int rc = dev_alloc(c, &c->d_img_aux, n_px * 4);
    if (rc) return rc;
    c->cap_img_aux = n_px * 4;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_img_depth, depth, dbytes, hipMemcpyHostToDevice, c->stream));
  DepthParams D{};
  D.depth = c->d_img_depth;
  if (label_img) {
    HIPCHK(c, hipMemcpyAsync(c->d_img_aux, label_img, n_px, hipMemcpyHostToDevice, c->stream));
    D.label_img = c->d_img_aux;
  } else if (rgba_img) {
    HIPCHK(c, hipMemcpyAsync(c->d_img_aux, rgba_img, n_px * 4, hipMemcpyHostToDevice, c->stream));
    D.rgba_img = c->d_img_aux;
  }
  D.fmt = depth_fmt;
  D.width = width;
  D.height = height;
  D.cx = K[2];
  D.cy = K[3];
  // unit_scaling / fx evaluated in double then narrowed, as depth_map_to_pointcloud.h:227-229
  const double unit = depth_fmt == 0 ? 1.0 : 0.001;
  D.constant_x = (float)(unit / (double)K[0]);
  D.constant_y = (float)(unit / (double)K[1]);
  return integrate_depth_impl(c, T, D, freespace, stats);
}

int ks_integrate_depth_device(ks_ctx* c, const float T[7], const void* d_depth, int depth_fmt, const uint8_t* d_label_img,
                              const uint8_t* d_rgba_img, int width, int height, const float K[4], int freespace,
                              ks_frame_stats* stats) {
  if (!c || !T || !d_depth || !K || width <= 0 || height <= 0 || (depth_fmt != 0 && depth_fmt != 1)) return KS_ERR_INVALID_ARG;
  DepthParams D{};
  D.depth = d_depth;
  D.label_img = d_label_img;
  D.rgba_img = d_label_img ? nullptr : d_rgba_img;
  D.fmt = depth_fmt;
  D.width = width;
  D.height = height;
  D.cx = K[2];
  D.cy = K[3];
  const double unit = depth_fmt == 0 ? 1.0 : 0.001;
  D.constant_x = (float)(unit / (double)K[0]);
  D.constant_y = (float)(unit / (double)K[1]);
  return integrate_depth_impl(c, T, D, freespace, stats);
}

int ks_num_blocks(ks_ctx* c, size_t* n) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  std::vector<int32_t> v;
  int rc = collect_block_indices(c, false, false, &v);
  if (rc) return rc;
  *n = v.size() / 3;
  return KS_OK;
}

int ks_get_block_indices(ks_ctx* c, int32_t* out, size_t cap, size_t* n) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  std::vector<int32_t> v;
  int rc = collect_block_indices(c, false, false, &v);
  if (rc) return rc;
  *n = v.size() / 3;
  if (out) std::memcpy(out, v.data(), std::min(cap, *n) * 3 * sizeof(int32_t));
  return KS_OK;
}

int ks_get_updated_block_indices(ks_ctx* c, int32_t* out, size_t cap, size_t* n, int reset) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  std::vector<int32_t> v;
  int rc = collect_block_indices(c, true, reset != 0, &v);
  if (rc) return rc;
  *n = v.size() / 3;
  if (out) std::memcpy(out, v.data(), std::min(cap, *n) * 3 * sizeof(int32_t));
  return KS_OK;
}

int ks_download_blocks(ks_ctx* c, const int32_t* idx, size_t n, void* tsdf_out, void* sem_out) {
  if (!c || (n && !idx)) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (int rc = quiesce(c)) return rc;
  const int vps = c->cfg.voxels_per_side;
  const size_t nv = (size_t)vps * vps * vps;
  // chunk so staging buffers stay bounded (<= ~256 MiB of semantic voxels)
  const size_t chunk = std::max<size_t>(1, (size_t(256) << 20) / (nv * 92));
  if (c->cap_out_blocks < std::min(chunk, n)) {
    const size_t cb = std::min(chunk, std::max<size_t>(n, 16));
    int rc;
    if ((rc = dev_alloc(c, &c->d_tsdf_out, cb * nv * 12))) return rc;
    if ((rc = dev_alloc(c, &c->d_sem_out, cb * nv * 92))) return rc;
    if ((rc = dev_alloc(c, &c->d_block_idx, cb * 3))) return rc;
    c->cap_out_blocks = cb;
  }
  for (size_t off = 0; off < n; off += c->cap_out_blocks) {
    const size_t m = std::min(c->cap_out_blocks, n - off);
    HIPCHK(c, hipMemcpyAsync(c->d_block_idx, idx + 3 * off, m * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_download, dim3((uint32_t)((nv + 255) / 256), (uint32_t)m), dim3(256), 0, c->stream, c->table,
                       c->pool, c->d_block_idx, vps, c->d_label_lut, tsdf_out ? c->d_tsdf_out : nullptr,
                       sem_out ? c->d_sem_out : nullptr);
    if (tsdf_out)
      HIPCHK(c, hipMemcpyAsync((uint8_t*)tsdf_out + off * nv * 12, c->d_tsdf_out, m * nv * 12, hipMemcpyDeviceToHost, c->stream));
    if (sem_out)
      HIPCHK(c, hipMemcpyAsync((uint8_t*)sem_out + off * nv * 92, c->d_sem_out, m * nv * 92, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return KS_OK;
}

int ks_upload_blocks(ks_ctx* c, const int32_t* idx, size_t n, const void* tsdf_in, const void* sem_in) {
  if (!c || (n && !idx) || (!tsdf_in && !sem_in)) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (c->fatal) return KS_ERR_INVALID_ARG;
  const int vps = c->cfg.voxels_per_side;
  const int tpb = vps / 8;  // tiles per block edge
  const size_t nv = (size_t)vps * vps * vps;
  std::vector<uint64_t> keys;
  keys.reserve(n * tpb * tpb * tpb);
  const int lim = kTileBias / tpb;
  for (size_t b = 0; b < n; ++b) {
    const int bx = idx[3 * b], by = idx[3 * b + 1], bz = idx[3 * b + 2];
    if (bx < -lim || bx >= lim || by < -lim || by >= lim || bz < -lim || bz >= lim) {
      c->err = "block index outside the packed tile-key range";
      return KS_ERR_INDEX_RANGE;
    }
    for (int z = 0; z < tpb; ++z)
      for (int y = 0; y < tpb; ++y)
        for (int x = 0; x < tpb; ++x) keys.push_back(pack_tile(bx * tpb + x, by * tpb + y, bz * tpb + z));
  }
  uint64_t* d_keys = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_keys, keys.size() * sizeof(uint64_t)));
  HIPCHK(c, hipMemcpyAsync(d_keys, keys.data(), keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  int rc = insert_tiles(c, d_keys, keys.size());
  (void)hipFree(d_keys);
  if (rc) return rc;
  // staging buffers shared with ks_download_blocks
  const size_t chunk = std::max<size_t>(1, (size_t(256) << 20) / (nv * 92));
  if (c->cap_out_blocks < std::min(chunk, n)) {
    const size_t cb = std::min(chunk, std::max<size_t>(n, 16));
    if ((rc = dev_alloc(c, &c->d_tsdf_out, cb * nv * 12))) return rc;
    if ((rc = dev_alloc(c, &c->d_sem_out, cb * nv * 92))) return rc;
    if ((rc = dev_alloc(c, &c->d_block_idx, cb * 3))) return rc;
    c->cap_out_blocks = cb;
  }
  for (size_t off = 0; off < n; off += c->cap_out_blocks) {
    const size_t m = std::min(c->cap_out_blocks, n - off);
    HIPCHK(c, hipMemcpyAsync(c->d_block_idx, idx + 3 * off, m * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (tsdf_in)
      HIPCHK(c, hipMemcpyAsync(c->d_tsdf_out, (const uint8_t*)tsdf_in + off * nv * 12, m * nv * 12, hipMemcpyHostToDevice, c->stream));
    if (sem_in)
      HIPCHK(c, hipMemcpyAsync(c->d_sem_out, (const uint8_t*)sem_in + off * nv * 92, m * nv * 92, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_upload, dim3((uint32_t)((nv + 255) / 256), (uint32_t)m), dim3(256), 0, c->stream, c->table, c->pool,
                       c->d_block_idx, vps, tsdf_in ? (const uint8_t*)c->d_tsdf_out : nullptr,
                       sem_in ? (const uint8_t*)c->d_sem_out : nullptr);
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, hipGetLastError());
  return KS_OK;
}

int ks_debug_radix_sort(ks_ctx* c, void* keys, uint32_t* vals, size_t n, int key_bits, unsigned end_bit) {
  if (!c || (n && !keys) || (key_bits != 32 && key_bits != 64)) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  const size_t kb = key_bits / 8;
  void *ka = nullptr, *kbuf = nullptr;
  uint32_t *va = nullptr, *vb = nullptr;
  HIPCHK(c, hipMalloc(&ka, n * kb));
  HIPCHK(c, hipMalloc(&kbuf, n * kb));
  HIPCHK(c, hipMemcpy(ka, keys, n * kb, hipMemcpyHostToDevice));
  if (vals) {
    HIPCHK(c, hipMalloc((void**)&va, n * 4));
    HIPCHK(c, hipMalloc((void**)&vb, n * 4));
    HIPCHK(c, hipMemcpy(va, vals, n * 4, hipMemcpyHostToDevice));
  }
  void* kres = nullptr;
  uint32_t* vres = nullptr;
  int rc = KS_OK;
  if (key_bits == 32) {
    uint32_t* r = nullptr;
    rc = vals ? sort_pairs(c, (uint32_t*)ka, (uint32_t*)kbuf, va, vb, n, end_bit, &r, &vres)
              : sort_keys(c, (uint32_t*)ka, (uint32_t*)kbuf, n, end_bit, &r);
    kres = r;
  } else {
    uint64_t* r = nullptr;
    rc = vals ? sort_pairs(c, (uint64_t*)ka, (uint64_t*)kbuf, va, vb, n, end_bit, &r, &vres)
              : sort_keys(c, (uint64_t*)ka, (uint64_t*)kbuf, n, end_bit, &r);
    kres = r;
  }
  if (rc == KS_OK) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(keys, kres, n * kb, hipMemcpyDeviceToHost));
    if (vals) HIPCHK(c, hipMemcpy(vals, vres, n * 4, hipMemcpyDeviceToHost));
  }
  (void)hipFree(ka);
  (void)hipFree(kbuf);
  if (va) (void)hipFree(va);
  if (vb) (void)hipFree(vb);
  return rc;
}

int ks_get_tile_keys(ks_ctx* c, uint64_t* out, size_t cap, size_t* n) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n = c->tiles_initialised;
  const size_t m = std::min<size_t>(cap, *n);
  if (out && m) HIPCHK(c, hipMemcpy(out, c->table.slot_keys, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return KS_OK;
}

int ks_export_tiles_device(ks_ctx* c, const uint32_t* slots, size_t n, void* d_payload) {
  if (!c || (n && (!slots || !d_payload))) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (int rc = quiesce(c)) return rc;
  for (size_t i = 0; i < n; ++i)
    if (slots[i] >= c->tiles_initialised) return KS_ERR_INVALID_ARG;
  uint32_t* d_slots = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_slots, n * sizeof(uint32_t)));
  HIPCHK(c, hipMemcpyAsync(d_slots, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_export_tiles, dim3((uint32_t)n), dim3(512), 0, c->stream, c->pool, d_slots, (uint4*)d_payload);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_slots);
  return KS_OK;
}

int ks_merge_tiles_device(ks_ctx* c, const uint64_t* keys, size_t n, const void* d_payload) {
  if (!c || (n && (!keys || !d_payload))) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (c->fatal) return KS_ERR_INVALID_ARG;
  uint64_t* d_keys = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_keys, n * sizeof(uint64_t)));
  HIPCHK(c, hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  // allocate tiles this rank has not seen yet
  if (int rc = insert_tiles(c, d_keys, n)) {
    (void)hipFree(d_keys);
    return rc;
  }
  switch (c->cfg.color_mode) {
    case KS_COLOR_MODE_COLOR:
      hipLaunchKernelGGL(k_merge_tiles<KS_COLOR_MODE_COLOR>, dim3((uint32_t)n), dim3(512), 0, c->stream, c->table, c->pool,
                         d_keys, (const uint4*)d_payload, c->cfg.max_weight, c->d_label_lut);
      break;
    case KS_COLOR_MODE_SEMANTIC:
      hipLaunchKernelGGL(k_merge_tiles<KS_COLOR_MODE_SEMANTIC>, dim3((uint32_t)n), dim3(512), 0, c->stream, c->table,
                         c->pool, d_keys, (const uint4*)d_payload, c->cfg.max_weight, c->d_label_lut);
      break;
    default:
      hipLaunchKernelGGL(k_merge_tiles<KS_COLOR_MODE_SEMANTIC_PROBABILITY>, dim3((uint32_t)n), dim3(512), 0, c->stream,
                         c->table, c->pool, d_keys, (const uint4*)d_payload, c->cfg.max_weight, c->d_label_lut);
      break;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  (void)hipFree(d_keys);
  return KS_OK;
}

int ks_clear(ks_ctx* c) {
  if (!c) return KS_ERR_INVALID_ARG;
  for (auto& S : c->slot) S.pending = false;  // a frame that was never applied is dropped with the map
  if (c->stream_tail != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_tail));
  if (c->stream_march != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_march));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemset(c->table.ent, 0xff, ((size_t)c->table.mask + 1) * sizeof(TileEntry)));
  HIPCHK(c, hipMemset(c->pool.updated, 0, c->cfg.max_tiles));
  HIPCHK(c, hipMemset(c->d_state, 0, 64 * (kSlots + 1)));
  c->tiles_initialised = 0;
  c->fatal = false;
  return KS_OK;
}

int ks_flush(ks_ctx* c, ks_frame_stats* stats) {
  if (!c) return KS_ERR_INVALID_ARG;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const bool had = c->slot[0].pending || c->slot[1].pending;
  const int rc = flush_pending(c, stats);
  if (rc == KS_OK && !had && c->stats_undelivered && stats) *stats = c->last_stats;
  if (stats) c->stats_undelivered = false;
  return rc;
}

int ks_synchronize(ks_ctx* c) {
  if (!c) return KS_ERR_INVALID_ARG;
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KS_OK;
}

void* ks_stream(ks_ctx* c) { return c ? (void*)c->stream : nullptr; }

int ks_profile_enable(ks_ctx* c, int level) {
  if (!c || level < 0 || level > 2) return KS_ERR_INVALID_ARG;
  c->profiling = level;
  return KS_OK;
}

int ks_profile_get(ks_ctx* c, ks_profile* out, int reset) {
  if (!c || !out) return KS_ERR_INVALID_ARG;
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < kProfSets; ++i) resolve_prof(c, i);
  *out = c->prof;
  if (reset) c->prof = ks_profile{};
  return KS_OK;
}

}  // extern "C"
