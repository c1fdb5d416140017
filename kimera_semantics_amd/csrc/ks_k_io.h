// ks_k_io.h — kernels around the hot path: multi-GPU tile exchange/merge, the depth-image
// front end, sorted integration order, host-layout download / upload.
#pragma once
#include "ks_types.h"

namespace ksk {
// ------------------------------------------------------------------------------------------
// Multi-GPU exchange (new functionality): tiles travel as raw 64 KiB records.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_export_tiles(Pool P, const uint32_t* __restrict__ slots, uint4* __restrict__ out) {
  const uint4* src = P.vox + (size_t)slots[blockIdx.x] * kTileVoxels * 8;
  uint4* dst = out + (size_t)blockIdx.x * kTileVoxels * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) dst[r * 512 + threadIdx.x] = src[r * 512 + threadIdx.x];
}

// tiles handed to their owner rank start over as empty deltas (same content as k_init_tiles writes)
__global__ void __launch_bounds__(512) k_reset_tiles(Pool P, const uint32_t* __restrict__ slots) {
  const size_t slot = slots[blockIdx.x];
  uint4* tile = P.vox + slot * (size_t)kTileVoxels * 8;
  const uint32_t pi = __float_as_uint(kPriorInit);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t q = r * 512u + threadIdx.x;
    const uint32_t sub = q & 7u;
    uint4 v;
    if (sub == 0) v = make_uint4(0u, 0u, 0u, 255u);
    else if (sub < 6) v = make_uint4(pi, pi, pi, pi);
    else if (sub == 6) v = make_uint4(pi, 0u, 0u, 0u);
    else v = make_uint4(0u, 0u, 0u, 0u);
    tile[q] = v;
  }
  if (threadIdx.x == 0) {
    P.updated[slot] = 1;
    P.dirty[slot] = 0;
  }
}

// pool growth: the tile table is rebuilt at twice the capacity; slot numbers are kept
__global__ void __launch_bounds__(256) k_rehash_tiles(TileTable T, const uint64_t* __restrict__ slot_keys, uint32_t n) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const uint64_t key = slot_keys[slot];
  uint32_t h = mix64(key) & T.mask;
  for (uint32_t probes = 0; probes <= T.mask; ++probes) {
    const uint64_t old = atomicCAS((unsigned long long*)&T.ent[h].key, (unsigned long long)kEmpty64, (unsigned long long)key);
    if (old == kEmpty64) {
      T.ent[h].val = slot;
      return;
    }
    h = (h + 1) & T.mask;
  }
}

// ks_reduce: the tiles touched since the last reduce that ANOTHER rank owns, grouped by owner.  Owner of a tile =
// splitmix64(key) % world (ks_tile_owner).  Pass 0 counts per owner; pass 1 (offsets known) writes slots and keys.
__device__ __forceinline__ uint32_t tile_owner_dev(uint64_t x, uint32_t world) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x % (uint64_t)world);
}
__global__ void __launch_bounds__(256) k_dirty_by_owner(Pool P, const uint64_t* __restrict__ slot_keys, uint32_t nt, uint32_t rank,
                                                        uint32_t world, int pass, int32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor,
                                                        uint32_t* __restrict__ out_slots, uint64_t* __restrict__ out_keys) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nt || !P.dirty[s]) return;
  const uint64_t key = slot_keys[s];
  const uint32_t o = tile_owner_dev(key, world);
  if (o == rank) return;
  if (pass == 0) {
    atomicAdd(&counts[o], 1);
  } else {
    const uint32_t at = offs[o] + atomicAdd(&cursor[o], 1u);
    out_slots[at] = s;
    out_keys[at] = key;
  }
}

__global__ void __launch_bounds__(256) k_insert_tiles(TileTable T, Counters* C, const uint64_t* __restrict__ keys, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tile_insert(T, C, keys[i]);
}

// Merge incoming tiles into the resident map; 8 lanes per voxel.
//   TSDF: Voxblox's layer-merge rule (mergeVoxelAIntoVoxelB): weight-averaged distance and
//         colour, summed weight (clamped to max_weight);
//   semantics: log-likelihoods are additive: priors += (incoming - initial), then argmax/colour
//         exactly as updateSemanticVoxel ends ([K:src/semantic_integrator_base.cpp:164-191]).
template <int COLOR_MODE>
__global__ void __launch_bounds__(512) k_merge_tiles(TileTable T, Pool P, const uint64_t* __restrict__ ukeys,
                                                     const uint32_t* __restrict__ offs, const uint32_t* __restrict__ idx,
                                                     const uint4* __restrict__ in, float max_weight,
                                                     const uint32_t* __restrict__ label_lut) {
  // One workgroup per DISTINCT tile key; the incoming tiles of that key (idx[offs[b] .. offs[b+1]),
  // in the caller's order = ascending source rank) are folded in one after the other, so one
  // launch merges everything a rank received and the result does not depend on scheduling.
  const uint32_t slot = tile_lookup(T, ukeys[blockIdx.x]);
  if (slot == 0xffffffffu) return;
  const uint32_t j0 = offs[blockIdx.x], j1 = offs[blockIdx.x + 1];
  uint4* dst = P.vox + (size_t)slot * kTileVoxels * 8;
  const uint32_t lane = lane_id(), sub = lane & 7u;
  const uint32_t cbase = (sub - 1u) * 4u;
  for (uint32_t r = 0; r < 8; ++r) {
    const uint32_t q = r * 512u + threadIdx.x;  // uint4 index in the tile; voxel = q >> 3
    uint4 b = dst[q];
    bool any = false;
    for (uint32_t j = j0; j < j1; ++j) {
      const uint4 a = in[(size_t)idx[j] * kTileVoxels * 8 + q];
      // every lane of the voxel's group needs A's label (dword 3 of sub 0)
      const uint32_t a_label = perm_u(a.w, lane & ~7u);
      if (a_label == 255u) continue;  // the sender never updated this voxel
      any = true;
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
        b.x = __float_as_uint(__uint_as_float(b.x) + (__uint_as_float(a.x) - kPriorInit));
        if (sub < 6u) {
          b.y = __float_as_uint(__uint_as_float(b.y) + (__uint_as_float(a.y) - kPriorInit));
          b.z = __float_as_uint(__uint_as_float(b.z) + (__uint_as_float(a.z) - kPriorInit));
          b.w = __float_as_uint(__uint_as_float(b.w) + (__uint_as_float(a.w) - kPriorInit));
        }
      }
    }
    // calculateMaximumLikelihoodLabel on the merged priors: first strict maximum
    float bv = -INFINITY;
    uint32_t bi = 1000u;
    if (any && sub >= 1u && sub < 7u) {
      bv = __uint_as_float(b.x); bi = cbase;
      if (sub < 6u) {
        const float p1 = __uint_as_float(b.y), p2 = __uint_as_float(b.z), p3 = __uint_as_float(b.w);
        if (p1 > bv) { bv = p1; bi = cbase + 1u; }
        if (p2 > bv) { bv = p2; bi = cbase + 2u; }
        if (p3 > bv) { bv = p3; bi = cbase + 3u; }
      } else {
        b.y = 1u;  // dword 25: written since the last voxel-level host sync (merged voxels are reported by it too)
        b.z = b.w = 0u;
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const float ov = perm_f(bv, lane ^ (uint32_t)o);
      const uint32_t oi = perm_u(bi, lane ^ (uint32_t)o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (any && sub < 7u) {
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


// Voxel-level host sync (strict drop-in: the host Layers are brought up to date after every frame).  Only the
// voxels the update kernels wrote since the last sync travel: record dword 25 is their dirty mark.
//   k_list_updated_tiles : slots of tiles flagged `updated` -> compact list
//   k_export_dirty       : one workgroup per listed tile, one lane per voxel; dirty voxels are packed as 120-byte
//                          host records {int32 block x, y, z; uint32 linear index in the block; TsdfVoxel 12 B;
//                          SemanticVoxel 92 B}, a tile's voxels contiguous; count_only: just the total.
__global__ void __launch_bounds__(256) k_list_updated_tiles(Pool P, uint32_t n_tiles, uint32_t* __restrict__ list,
                                                            uint32_t* __restrict__ counters) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = s < n_tiles && P.updated[s] != 0;
  const uint32_t pos = block_append(on, &counters[0]);
  if (on) list[pos] = s;
}
constexpr uint32_t kVoxRecBytes = 120;
__global__ void __launch_bounds__(512) k_export_dirty(TileTable T, Pool P, const uint32_t* __restrict__ list,
                                                      const uint32_t* __restrict__ label_lut, int vps_shift, int count_only,
                                                      uint32_t* __restrict__ counters, uint8_t* __restrict__ out,
                                                      uint32_t* __restrict__ runs) {
  __shared__ uint32_t s_wave[8];
  __shared__ uint32_t s_base, s_total;
  __shared__ __attribute__((aligned(16))) uint32_t s_rec[kTileVoxels * (kVoxRecBytes / 4)];
  const uint32_t slot = list[blockIdx.x];
  const uint32_t local = threadIdx.x;
  uint32_t* rec = (uint32_t*)(P.vox + ((size_t)slot * kTileVoxels + local) * 8);
  const bool dirty = rec[25] != 0u;
  // the tile's dirty voxels get one contiguous range of records (one atomic per workgroup)
  const unsigned long long m = __ballot(dirty);
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < 8; ++w) {
      const uint32_t t = s_wave[w];
      s_wave[w] = total;
      total += t;
    }
    s_total = total;
    s_base = total ? atomicAdd(&counters[1], total) : 0u;
  }
  __syncthreads();
  const uint32_t pos = s_base + s_wave[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (count_only) return;
  int tx, ty, tz;
  unpack_tile(T.slot_keys[slot], tx, ty, tz);
  if (threadIdx.x == 0) {
    P.updated[slot] = 0;
    if (runs) {  // {block x, y, z, first record, records}: a device tile lies inside one host block
      uint32_t* r = runs + (size_t)blockIdx.x * 5;
      r[0] = (uint32_t)(tx >> vps_shift);
      r[1] = (uint32_t)(ty >> vps_shift);
      r[2] = (uint32_t)(tz >> vps_shift);
      r[3] = s_base;
      r[4] = s_total;
    }
  }
  // records are assembled in LDS in the tile's record order, then leave as one contiguous, coalesced range
  // (the target may be pinned host memory: scattered dword stores would crawl over PCIe)
  if (dirty) {
    rec[25] = 0u;
    const int vps = 8 << vps_shift;
    const int vx = tx * 8 + (int)(local & 7u), vy = ty * 8 + (int)((local >> 3) & 7u), vz = tz * 8 + (int)(local >> 6);
    uint32_t* o = s_rec + (size_t)(pos - s_base) * (kVoxRecBytes / 4);
    o[0] = (uint32_t)(vx >> (3 + vps_shift));
    o[1] = (uint32_t)(vy >> (3 + vps_shift));
    o[2] = (uint32_t)(vz >> (3 + vps_shift));
    o[3] = (uint32_t)((vx & (vps - 1)) + vps * ((vy & (vps - 1)) + vps * (vz & (vps - 1))));
    const uint32_t label = rec[3];
    const bool touched = label != 255u;
    o[4] = rec[0];
    o[5] = rec[1];
    o[6] = rec[2];
    o[7] = touched ? label : 0u;
#pragma unroll
    for (int k = 0; k < kNumLabels; ++k) o[8 + k] = rec[4 + k];
    o[29] = touched ? label_lut[label] : (127u | (127u << 8) | (127u << 16) | (255u << 24));
  }
  __syncthreads();
  const uint32_t n2 = s_total * (kVoxRecBytes / 8);  // 120-byte records: the range starts 8-byte aligned
  uint2* dst = (uint2*)(out + (size_t)s_base * kVoxRecBytes);
  const uint2* src = (const uint2*)s_rec;
  for (uint32_t i = threadIdx.x; i < n2; i += 512) dst[i] = src[i];
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

}  // namespace ksk
