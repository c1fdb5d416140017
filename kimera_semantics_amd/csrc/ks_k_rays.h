// ks_k_rays.h — stage A kernels: per-point work, start-voxel dedup (fast), bundle merge (merged).
#pragma once
#include "ks_types.h"
#include "ks_k_bundle_order.h"

namespace ksk {
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
                                                      uint32_t* __restrict__ cnt, uint8_t* __restrict__ live,
                                                      Counters* C) {
  // One lane per point in MEMORY order (coalesced reads, coalesced descriptor writes); the
  // integration position p of the point is arithmetic, only the 4-byte sort key is scattered.
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  bool counted = false;
  if (idx < F.n) {
    cnt[idx] = 0u;   // per-position update counts / live flags of stage B (set by k_dedup for the kept rays)
    live[idx] = 0;
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
                                                const RayDesc* __restrict__ rays, uint32_t* __restrict__ cnt,
                                                uint8_t* __restrict__ live, Counters* C) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = F.n;
  bool kept = false;
  uint32_t p = 0;
  if (i < n && C->err == 0) {
    const uint32_t slot = skeys[i];
    if (slot != kInvalidSlot) {
      p = svals[i];
      const uint32_t idx = point_order(F, F.order, p);
      const uint64_t h = hash[idx];
      const bool first = (i == 0) || (skeys[i - 1] != slot);
      uint64_t prev;
      // the slot's persistent content is only READ here; k_dedup_commit writes it afterwards
      if (first) prev = start_set[slot];
      else prev = hash[point_order(F, F.order, svals[i - 1])];
      kept = prev != h;
      if (kept) {
        // the ray's full length (the early-out of stage B can only shorten it)
        const RayDesc d = rays[idx];
        Dda dda;
        dda.setup(F.T.t, {d.px, d.py, d.pz}, ((d.info >> 10) & 1u) != 0, F.carving != 0, F.max_ray, F.voxel_size_inv,
                  F.trunc, /*cast_from_origin=*/false);
        if (!dda.in_range) atomicOr(&C->err, kErrIndex);
        cnt[p] = (uint32_t)dda.steps + 1u;
        live[p] = 1;
      }
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
// validity + end-voxel key of point idx: kEmpty64 = not integrated, else clearing << 63 | x << 42 | y << 21 | z (biased by kCoordBias)
__device__ __forceinline__ uint64_t merged_point_key(const FrameParams& F, const float* __restrict__ xyz, const uint8_t* __restrict__ rgba,
                                                     const uint8_t* __restrict__ labels, const uint8_t* __restrict__ color_lut,
                                                     uint32_t idx, Counters* C, bool report) {
  const f3 pc = {xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
  uint32_t label;
  if (labels) label = labels[idx];
  else label = (color_lut && rgba) ? color_lut[((const uint32_t*)rgba)[idx] & 0xffffffu] : 0u;
  if (label >= (uint32_t)kNumLabels) {
    if (report) atomicOr(&C->err, kErrLabel);
    return kEmpty64;
  }
  const int valid = point_validity(pc, F.min_ray, F.max_ray, F.allow_clear != 0, F.freespace != 0);
  if (!valid) return kEmpty64;
  const f3 pg = transform_point(F.T, pc);
  const float gx = grid_coord(pg.x, F.voxel_size_inv), gy = grid_coord(pg.y, F.voxel_size_inv), gz = grid_coord(pg.z, F.voxel_size_inv);
  const float lim = (float)(kCoordBias - 1);
  if (!(fabsf(gx) < lim && fabsf(gy) < lim && fabsf(gz) < lim)) {
    if (report) atomicOr(&C->err, kErrIndex);
    return kEmpty64;
  }
  return ((uint64_t)(valid == 2 ? 1u : 0u) << 63) | ((uint64_t)(uint32_t)((int)gx + kCoordBias) << 42) |
         ((uint64_t)(uint32_t)((int)gy + kCoordBias) << 21) | (uint64_t)(uint32_t)((int)gz + kCoordBias);
}

// The voxels outside the key window (FrameParams::key_base / key_bits): a slot per distinct end-voxel key in an open-addressing
// table of >= 2 n entries (0 = free: no valid key is 0), the slot number is the group's key.  Which slot a voxel gets depends on
// who came first; that it is the same slot for every point of the voxel, and no other voxel's, does not.  k_gather_sorted frees
// the slots again.
__device__ __forceinline__ uint32_t key_overflow_slot(uint64_t* __restrict__ tab, uint32_t mask, uint64_t key) {
  uint64_t x = key + 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  uint32_t h = (uint32_t)(x ^ (x >> 31)) & mask;
  for (;;) {
    const unsigned long long prev = atomicCAS((unsigned long long*)&tab[h], 0ull, (unsigned long long)key);
    if (prev == 0ull || prev == (unsigned long long)key) return h;
    h = (h + 1u) & mask;
  }
}

__global__ void __launch_bounds__(1024) k_points_merged(FrameParams F, const float* __restrict__ xyz,
                                                        const uint8_t* __restrict__ rgba,
                                                        const uint8_t* __restrict__ labels,
                                                        const uint8_t* __restrict__ color_lut,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ keys32,
                                                        uint32_t* __restrict__ vals, uint32_t* __restrict__ cnt,
                                                        uint32_t* __restrict__ bo_flag, uint64_t* __restrict__ overflow_tab,
                                                        uint32_t overflow_mask, Counters* C) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  bool counted = false;
  if (idx < F.n) {
    cnt[idx] = 0u;        // update counts of stage B: normal bundles at their integration id,
    cnt[idx + F.n] = 0u;  // clearing bundles n further on (they integrate after all normal ones)
    if (bo_flag) {        // reference bundle order: first-point marks of this frame's bundles (ks_k_bundle_order.h)
      bo_flag[idx] = 0u;
      bo_flag[idx + F.n] = 0u;
    }
    const uint64_t key = merged_point_key(F, xyz, rgba, labels, color_lut, idx, C, true);
    counted = key != kEmpty64;
    const uint32_t pos = point_position(F, F.inv_order, idx);
    if (F.key_bits) {
      uint32_t kw = kEmpty32;
      if (counted) {
        const uint32_t w = F.key_bits;
        const uint32_t rx = (uint32_t)((int)((key >> 42) & 0x1fffffu) - kCoordBias - F.key_base[0]);
        const uint32_t ry = (uint32_t)((int)((key >> 21) & 0x1fffffu) - kCoordBias - F.key_base[1]);
        const uint32_t rz = (uint32_t)((int)(key & 0x1fffffu) - kCoordBias - F.key_base[2]);
        if (((rx | ry | rz) >> w) == 0u) kw = ((uint32_t)(key >> 63) << (3u * w)) | (rx << (2u * w)) | (ry << w) | rz;
        else kw = 0x80000000u | key_overflow_slot(overflow_tab, overflow_mask, key);
      }
      keys32[pos] = kw;
    } else {
      keys[pos] = key;
    }
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
                                                       const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ skeys32,
                                                       uint64_t* __restrict__ skeys_out,
                                                       const uint32_t* __restrict__ svals, float4* __restrict__ g_pw,
                                                       uint2* __restrict__ g_lc, uint32_t* __restrict__ bo_flag,
                                                       uint32_t* __restrict__ long_list, uint64_t* __restrict__ overflow_tab, Counters* C) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  uint64_t key;
  bool head, is_long;
  if (skeys32) {
    // compact grouping keys: the 64-bit end-voxel key of the sorted point is computed again (the same arithmetic on the same
    // operands as k_points_merged) for everything downstream
    const uint32_t kw = skeys32[i];
    if (kw == kEmpty32) {
      skeys_out[i] = kEmpty64;
      return;
    }
    key = merged_point_key(F, xyz, rgba, labels, color_lut, point_order(F, order, svals[i]), C, false);
    skeys_out[i] = key;
    if (kw >> 31) overflow_tab[kw & 0x7fffffffu] = 0ull;   // (the slot is free again for the next frame)
    head = i == 0 || skeys32[i - 1] != kw;
    is_long = head && i + kLongRun < F.n && skeys32[i + kLongRun] == kw;
  } else {
    key = skeys[i];
    if (key == kEmpty64) return;
    head = i == 0 || skeys[i - 1] != key;
    is_long = head && i + kLongRun < F.n && skeys[i + kLongRun] == key;
  }
  if (head) {
    // a bundle's first point in integration order: its insertion
    if (bo_flag) bo_flag[svals[i] + (uint32_t)(key >> 63) * F.n] = 1u;
    // the bundles of >= kLongRun points (a handful per frame), listed here so that their merge can start before the
    // bundle order is known (k_bundles_long beside k_bo_* and k_bundles)
    if (is_long) long_list[atomicAdd(&C->n_long_bundles, 1u)] = i;
  }
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
                                              bool clearing, int n_labels, int the_label, float c, RayDesc* out,
                                              uint32_t* cnt_out, Counters* C) {
  const f3 pg = transform_point(F.T, mp);
  {
    // length of the bundle's ray (origin -> surface, [K:src/semantic_tsdf_integrator_merged.cpp:288-294])
    Dda dda;
    dda.setup(F.T.t, pg, clearing, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc, /*cast_from_origin=*/true);
    if (!dda.in_range) atomicOr(&C->err, kErrIndex);
    *cnt_out = (uint32_t)dda.steps + 1u;
  }
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

__device__ __forceinline__ void bundles_body(const FrameParams& F, const uint64_t* __restrict__ skeys,
                                             const uint32_t* __restrict__ svals, const float4* __restrict__ g_pw,
                                             const uint2* __restrict__ g_lc, RayDesc* __restrict__ rays,
                                             float* __restrict__ deltas, uint32_t* __restrict__ ray_list,
                                             uint64_t* __restrict__ ray_keys, uint32_t* __restrict__ cnt, const BoCtx& X,
                                             bool use_rank, Counters* C, uint32_t block) {
  const uint32_t i = block * blockDim.x + threadIdx.x;
  bool head = false, is_long = false;
  uint32_t first_p = 0;
  uint64_t key = 0;
  if (i < F.n && C->err == 0) {
    key = skeys[i];
    head = (key != kEmpty64) && (i == 0 || skeys[i - 1] != key);
    if (head) is_long = (i + kLongRun < F.n) && (skeys[i + kLongRun] == key);
  }
  const bool work = head && !is_long;   // (the long ones: listed by k_gather_sorted, merged by k_bundles_long)
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

    first_p = bundle_id(X, use_rank, svals, i, clearing);  // the bundle's integration id
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
    finish_bundle(F, mp, mw, merged_color, clearing, n_labels, the_label, c, &rays[first_p],
                  &cnt[first_p + (clearing ? F.n : 0u)], C);
    if (ray_keys) ray_keys[first_p] = key & ~(1ull << 63);
  }
  const uint32_t pos = block_append(work, &C->n_rays);
  if (work) ray_list[pos] = first_p;
}
__global__ void __launch_bounds__(256) k_bundles(FrameParams F, const uint64_t* __restrict__ skeys,
                                                 const uint32_t* __restrict__ svals, const float4* __restrict__ g_pw,
                                                 const uint2* __restrict__ g_lc, RayDesc* __restrict__ rays,
                                                 float* __restrict__ deltas, uint32_t* __restrict__ ray_list,
                                                 uint64_t* __restrict__ ray_keys,
                                                 uint32_t* __restrict__ cnt, BoCtx X, bool use_rank, Counters* C) {
  bundles_body(F, skeys, svals, g_pw, g_lc, rays, deltas, ray_list, ray_keys, cnt, X, use_rank, C, blockIdx.x);
}

// A merged long bundle -> its ray (one wavefront; freq: lane l < kNumLabels holds the count of label l).
__device__ __forceinline__ void bundle_long_finish(const FrameParams& F, uint64_t key, uint32_t start, float freq, f3 mp, uint32_t merged_color,
                                                   float mw, const uint32_t* __restrict__ svals, RayDesc* __restrict__ rays,
                                                   float* __restrict__ deltas, uint32_t* __restrict__ ray_list,
                                                   uint64_t* __restrict__ ray_keys, uint32_t* __restrict__ cnt, const BoCtx& X,
                                                   bool use_rank, Counters* C) {
  const int lane = (int)lane_id();
  const bool clearing = (key >> 63) != 0;
  const uint32_t first_p = bundle_id(X, use_rank, svals, start, clearing);
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
    finish_bundle(F, mp, mw, merged_color, clearing, n_labels, the_label, c, &rays[first_p], &cnt[first_p + (clearing ? F.n : 0u)], C);
    if (ray_keys) ray_keys[first_p] = key & ~(1ull << 63);
    ray_list[atomicAdd(&C->n_rays, 1u)] = first_p;
  }
}

constexpr uint32_t kBundleLongRec = 32;   // floats per merged long bundle between k_bundles_long and k_bundles_long_finish

// Two wavefronts per long bundle (>= kLongRun points in one voxel: thousands when a wall is close).  The merge is the
// reference's serial recurrence [K:src/semantic_tsdf_integrator_merged.cpp:231-262 via voxblox's weighted mean]:
//     den = w + pw;  mean = (mean * w + p * pw) / den;  w = den           (per point, in input order, f32, no FMA)
// and what bounds the kernel is the number of instructions ONE wave issues per point of the longest bundle, so:
//   wave 0  per 64-point batch: the weight recurrence runs through the lanes — 63 DPP adds (lane k takes lane k-1's sum),
//           after which lane k holds the weight before and after its point; one reciprocal per lane; the label histogram;
//           the operands of the batch go to an LDS table (double-buffered, one workgroup barrier per batch);
//   wave 1  lanes 0/1/2 walk the x/y/z chains of the PREVIOUS batch at the same time: one 16-byte broadcast read + one
//           4-byte read per point, requested eight points ahead, then five dependent operations per point (multiply, add,
//           and the three of the division by a known reciprocal).  The exponent-window test of that division is taken
//           off the chain: eight points are applied without it, and repeated one by one if any of them fell outside.
template <bool FINISH>
__device__ __forceinline__ void bundles_long_body(const FrameParams& F, const uint64_t* __restrict__ skeys,
                                                  const uint32_t* __restrict__ svals, const float4* __restrict__ g_pw,
                                                  const uint2* __restrict__ g_lc, const uint32_t* __restrict__ long_list,
                                                  float* __restrict__ merged, RayDesc* __restrict__ rays,
                                                  float* __restrict__ deltas, uint32_t* __restrict__ ray_list,
                                                  uint64_t* __restrict__ ray_keys, uint32_t* __restrict__ cnt, const BoCtx& X,
                                                  bool use_rank, Counters* C, uint32_t block, uint32_t n_blocks) {
  __shared__ float4 s_w[2][64];  // per point of the batch: weight before it, weight after it, reciprocal of that, its own weight
  __shared__ float4 s_a[2][64];  // x * w, y * w, z * w, colour
  __shared__ unsigned long long s_use[2];  // the points of the batch that are merged
  __shared__ int s_last[2];                // 1: the bundle ends with this batch
  __shared__ float s_mp[4];                // the merged point (+ the blended colour) back to wave 0
  const uint32_t n_long = C->n_long_bundles;
  const int lane = (int)lane_id();
  const bool back = (threadIdx.x >> 6) != 0u;
  const int comp = lane < 3 ? lane : 0;
  const bool colour = F.color_mode == KS_COLOR_MODE_COLOR;
  for (uint32_t run = block; run < n_long; run += n_blocks) {
    if (back) {
      // ---- wave 1: the weighted-mean recurrence, lanes 0,1,2 one component each ----
      float mpc = 0.0f;
      uint32_t merged_color = 0;
      for (int buf = 0;; buf ^= 1) {
        __syncthreads();  // batch `buf` is complete
        const unsigned long long vmask = s_use[buf];
        const int last = s_last[buf];
        const float4* tw = s_w[buf];
        const float4* ta = s_a[buf];
        if (vmask == ~0ull && !colour) {
          // the common batch: every point used.  Operands of the next eight points are requested while eight are applied.
          float4 w0[8], w1[8];
          float a0[8], a1[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            w0[i] = tw[i];
            a0[i] = ((const float*)&ta[i])[comp];
          }
          auto eight = [&](const float4 (&w)[8], const float (&a)[8]) {
            const float at_start = mpc;
            // the exponent window of div_by_recip over the eight numerators as a running min / max: vector instructions only
            // (a compare per point writes scalar registers, and the scalar AND that collects them waits for the vector pipe in the
            // middle of the chain).  A NaN numerator passes — and gives NaN on either path.
            float an_lo = 1.0f, an_hi = 1.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float num = mpc * w[i].x + a[i];
              an_lo = fminf(an_lo, fabsf(num));
              an_hi = fmaxf(an_hi, fabsf(num));
              const float q0 = num * w[i].z;
              const float rem = __builtin_fmaf(-q0, w[i].y, num);
              mpc = __builtin_fmaf(rem, w[i].z, q0);
            }
            const bool in_window = (an_lo >= 1e-20f) && (an_hi <= 1e20f);
            if (__ballot(!in_window) != 0ull) {  // (a coordinate that is exactly 0, ...): one by one, with the test
              mpc = at_start;
#pragma unroll
              for (int i = 0; i < 8; ++i) mpc = div_by_recip(mpc * w[i].x + a[i], w[i].y, w[i].z);
            }
          };
#pragma unroll
          for (int g = 0; g < 8; g += 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              w1[i] = tw[8 * (g + 1) + i];
              a1[i] = ((const float*)&ta[8 * (g + 1) + i])[comp];
            }
            eight(w0, a0);
            if (g + 2 < 8) {
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                w0[i] = tw[8 * (g + 2) + i];
                a0[i] = ((const float*)&ta[8 * (g + 2) + i])[comp];
              }
            }
            eight(w1, a1);
          }
        } else {
          for (unsigned long long m = vmask; m; m &= m - 1ull) {
            const int k = __ffsll((long long)m) - 1;
            const float4 w = tw[k], ac = ta[k];
            mpc = div_by_recip(mpc * w.x + ((lane == 0) ? ac.x : (lane == 1) ? ac.y : ac.z), w.y, w.z);
            if (colour) merged_color = blend_two_colors(merged_color, w.x, __float_as_uint(ac.w), w.w);
          }
        }
        if (last) break;
      }
      if (lane < 3) s_mp[lane] = mpc;
      if (lane == 3) s_mp[3] = __uint_as_float(merged_color);
      __syncthreads();  // the merged point for wave 0
      __syncthreads();  // LDS free for the next bundle
      continue;
    }
    // ---- wave 0: everything else ----
    const uint32_t start = long_list[run];
    const uint64_t key = skeys[start];
    const bool clearing = (key >> 63) != 0;
    float mw = 0.0f;
    float freq = 0.0f;  // lane l < 21 counts label l
    uint32_t base = start;
    int buf = 0;
    // prefetch one batch ahead (contiguous, coalesced; two ahead measured the same: the chain, not memory, is what a batch waits for).
    // (key, point and label of a batch are requested TOGETHER, from a clamped index, and masked afterwards: a point load that waits
    // for its key's comparison puts two memory latencies on every batch of the chain: 171 -> 150 us at 640x480)
    uint32_t j = base + (uint32_t)lane;
    const uint32_t jc0 = j < F.n ? j : F.n - 1u;
    const uint64_t k0 = skeys[jc0];
    const float4 q0 = g_pw[jc0];
    const uint2 lc0 = g_lc[jc0];
    bool in = (j < F.n) && (k0 == key);
    float4 q = in ? q0 : make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 lc = in ? lc0 : make_uint2(0u, 0u);
    for (;;) {
      const int n_in = (int)__popcll(__ballot(in));  // >= 1 in the first batch (a long bundle has >= kLongRun points)
      const uint32_t jn = base + 64u + (uint32_t)lane;
      const uint32_t jc = jn < F.n ? jn : F.n - 1u;
      const uint64_t k_n = skeys[jc];
      const float4 q_r = g_pw[jc];
      const uint2 lc_r = g_lc[jc];

      const bool valid = in && !(q.w < kEps);
      unsigned long long vmask = __ballot(valid);
      bool last = n_in < 64;
      if (clearing && vmask) {  // only the first usable point of a clearing bundle is integrated
        vmask &= (~vmask + 1ull);
        last = true;
      }
      const bool use = valid && ((vmask >> lane) & 1ull);
      // s_k = s_(k-1) + pw_k in lane order (a point that is not used adds +0: the sum stays what it was, bit for bit)
      const float pw = use ? q.w : 0.0f;
      const float step = (lane == 0) ? 0.0f : pw;
      float s = (lane == 0) ? mw + pw : pw;
#pragma unroll
      for (int t = 0; t < 63; ++t) s = step + lane_below_f(s, s);
      const float before = lane_below_f(s, mw);
      mw = bcast_f(s, 63);
      s_w[buf][lane] = make_float4(before, s, 1.0f / s, q.w);
      s_a[buf][lane] = make_float4(q.x * q.w, q.y * q.w, q.z * q.w, __uint_as_float(lc.y));
      if (lane == 0) {
        s_use[buf] = vmask;
        s_last[buf] = last ? 1 : 0;
      }
      // label histogram: counts are order independent and exact in f32
#pragma unroll
      for (int l = 0; l < kNumLabels; ++l) {
        const unsigned long long lm = __ballot(use && lc.x == (uint32_t)l);
        if (lane == l) freq += (float)__popcll(lm);
      }
      __syncthreads();  // hand the batch to wave 1
      buf ^= 1;
      if (last) break;
      in = (jn < F.n) && (k_n == key);
      q = in ? q_r : make_float4(0.f, 0.f, 0.f, 0.f);
      lc = in ? lc_r : make_uint2(0u, 0u);
      base += 64u;
      if (__ballot(in) == 0ull) {  // the bundle ended on a batch boundary: an empty last batch
        if (lane == 0) {
          s_use[buf] = 0ull;
          s_last[buf] = 1;
        }
        __syncthreads();
        break;
      }
    }
    __syncthreads();  // wave 1 has the merged point
    if (FINISH) {
      // (the bundle order is known: k_bundles_all, after k_bo_*)
      const f3 mp = {s_mp[0], s_mp[1], s_mp[2]};
      bundle_long_finish(F, skeys[start], start, freq, mp, __float_as_uint(s_mp[3]), mw, svals, rays, deltas, ray_list, ray_keys, cnt, X, use_rank, C);
    } else {
      // the merged bundle for k_bundles_long_finish: label counts in lanes 0..20, then the point, the weight, the colour
      float v = freq;
      if (lane >= kNumLabels) v = (lane < kNumLabels + 4) ? s_mp[lane - kNumLabels] : mw;
      if (lane <= kNumLabels + 4) merged[(size_t)run * kBundleLongRec + (uint32_t)lane] = v;
    }
    __syncthreads();  // LDS free for the next bundle
  }
}
__global__ void __launch_bounds__(128) k_bundles_long(FrameParams F, const uint64_t* __restrict__ skeys,
                                                      const float4* __restrict__ g_pw, const uint2* __restrict__ g_lc,
                                                      const uint32_t* __restrict__ long_list, float* __restrict__ merged,
                                                      Counters* C) {
  bundles_long_body<false>(F, skeys, nullptr, g_pw, g_lc, long_list, merged, nullptr, nullptr, nullptr, nullptr, nullptr, BoCtx{}, false, C,
                           blockIdx.x, gridDim.x);
}

// Both in one launch, after the bundle order: the first n_long_blocks workgroups walk the long bundles (a serial chain each,
// 0.13-0.16 ms at 640x480 with a wall close to the sensor), the others merge the short ones under it (0.04 ms).
__global__ void __launch_bounds__(128) k_bundles_all(FrameParams F, const uint64_t* __restrict__ skeys,
                                                     const uint32_t* __restrict__ svals, const float4* __restrict__ g_pw,
                                                     const uint2* __restrict__ g_lc, const uint32_t* __restrict__ long_list,
                                                     RayDesc* __restrict__ rays, float* __restrict__ deltas,
                                                     uint32_t* __restrict__ ray_list, uint64_t* __restrict__ ray_keys,
                                                     uint32_t* __restrict__ cnt, BoCtx X, bool use_rank, Counters* C,
                                                     uint32_t n_long_blocks) {
  if (blockIdx.x < n_long_blocks)
    bundles_long_body<true>(F, skeys, svals, g_pw, g_lc, long_list, nullptr, rays, deltas, ray_list, ray_keys, cnt, X, use_rank, C, blockIdx.x,
                            n_long_blocks);
  else
    bundles_body(F, skeys, svals, g_pw, g_lc, rays, deltas, ray_list, ray_keys, cnt, X, use_rank, C, blockIdx.x - n_long_blocks);
}

// ... and what needs the bundle's integration id (k_bo_*): the ray descriptor, the increments, the lists.  A wavefront per bundle.
__global__ void __launch_bounds__(256) k_bundles_long_finish(FrameParams F, const uint64_t* __restrict__ skeys,
                                                             const uint32_t* __restrict__ svals,
                                                             const uint32_t* __restrict__ long_list,
                                                             const float* __restrict__ merged, RayDesc* __restrict__ rays,
                                                             float* __restrict__ deltas, uint32_t* __restrict__ ray_list,
                                                             uint64_t* __restrict__ ray_keys, uint32_t* __restrict__ cnt,
                                                             BoCtx X, bool use_rank, Counters* C) {
  const uint32_t n_long = C->n_long_bundles;
  const int lane = (int)lane_id();
  for (uint32_t run = blockIdx.x * 4u + (threadIdx.x >> 6); run < n_long; run += gridDim.x * 4u) {
    const uint32_t start = long_list[run];
    const float* rec = merged + (size_t)run * kBundleLongRec;
    const float freq = lane < kNumLabels ? rec[lane] : 0.0f;
    const f3 mp = {rec[kNumLabels], rec[kNumLabels + 1], rec[kNumLabels + 2]};
    bundle_long_finish(F, skeys[start], start, freq, mp, __float_as_uint(rec[kNumLabels + 3]), rec[kNumLabels + 4], svals, rays, deltas, ray_list,
                       ray_keys, cnt, X, use_rank, C);
  }
}

}  // namespace ksk
