// ks_k_march.h — stage B kernels: the ray march (+ pair emission), the counter snapshot, and
// the initialisation of freshly allocated tiles.
#pragma once
#include "ks_types.h"

namespace ksk {
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
  uint64_t info_hi = 0;  // label | kind << 5 | clearing << 7 in the key's top byte (rides through the sort)
  if (r < n_rays && (C->err & (kErrLabel | kErrIndex)) == 0) {
    const uint32_t p = ray_list[r];
    const RayDesc d = rays[ray_index(F, p)];
    clearing = ((d.info >> 10) & 1u) != 0;
    info_hi = (uint64_t)((d.info & 0x1fu) | (((d.info >> 8) & 3u) << 5) | (((d.info >> 10) & 1u) << 7)) << 56;
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
        else atomicOr(&C->err, kErrPool);  // a tile an earlier frame failed to allocate: this frame must not be applied either
      }
      const unsigned long long m = __ballot(emit);
      if (emit) {
        const uint32_t local = (uint32_t)(vx[j] & 7) + 8u * ((uint32_t)(vy[j] & 7) + 8u * (uint32_t)(vz[j] & 7));
        const uint32_t pos = wcount + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        buf[pos] = ((uint64_t)(slot * (uint32_t)kTileVoxels + local) << F.seq_bits) | seq | info_hi;
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

}  // namespace ksk
