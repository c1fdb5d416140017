// ks_k_apply.h — the voxel update: updateTsdfVoxel + updateSemanticVoxel replayed per voxel run.
#pragma once
#include "ks_types.h"

namespace ksk {
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
// HOT_ONLY: only the first 16 bytes of the ray descriptor (point_G, weight) are gathered; label /
// kind / clearing come from the top byte of the pair key, where k_march put them.  That is all the
// `fast` integrator needs unless colours are blended: one random 16-B gather per update, not two.
template <bool HOT_ONLY>
__device__ __forceinline__ UpdateOps load_update_ops(const FrameParams& F, const RayDesc* __restrict__ rays, uint64_t key,
                                                     const VoxelRef& v) {
  UpdateOps u;
  u.rp = (uint32_t)key & F.point_mask;
  const uint4* r4 = (const uint4*)rays + (size_t)ray_index(F, u.rp) * 2;
  const uint4 d0 = r4[0];
  tsdf_operands(F.tsdf, F.T.t, {__uint_as_float(d0.x), __uint_as_float(d0.y), __uint_as_float(d0.z)}, v.vx, v.vy, v.vz,
                __uint_as_float(d0.w), u.sdf, u.uw);
  if (HOT_ONLY) {
    const uint32_t b = (uint32_t)(key >> 56);
    u.color = 0u;
    u.dm = F.log_match;
    u.dn = F.log_non_match;
    u.info = (b & 0x1fu) | (((b >> 5) & 3u) << 8) | (((b >> 7) & 1u) << 10);
  } else {
    const uint4 d1 = r4[1];
    u.color = d1.x;
    u.dm = __uint_as_float(d1.y);
    u.dn = __uint_as_float(d1.z);
    u.info = d1.w;
  }
  return u;
}

// MERGED is a compile-time copy of F.method == KS_METHOD_MERGED (per-bundle increments, mixed-label
// increment vectors): uniform run-time tests in the step loop are not free.
template <int COLOR_MODE, bool MERGED>
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
  // label -> colour table in LDS: a global lookup per voxel would sit, with its full latency, between
  // the recurrence and the record store of every iteration
  __shared__ uint32_t s_lut[256];
  s_lut[threadIdx.x] = label_lut[threadIdx.x];  // 256 threads
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
    u = load_update_ops<!MERGED && COLOR_MODE != KS_COLOR_MODE_COLOR>(F, rays, key, voxel_ref(T, vox));
  }
  __syncthreads();  // s_lut

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
  // the records of the next TWO groups of heads are in flight while the current one is applied
  auto fetch = [&](int& pos, uint32_t& v, uint4& q) {
    pos = (8u * it < nh) ? next_head() : (++it, -1);
    v = perm_u(vox, pos >= 0 ? (uint32_t)pos : lane);
    // UNCONDITIONAL load (idle groups read record 0, sub 7 reads the spare 16 bytes): a load under a
    // divergent branch makes the compiler drain vmcnt at the join, which would serialise the prefetch
    q = (P.vox + (size_t)(pos >= 0 ? v : 0u) * 8)[sub];
  };
  const uint32_t n_it = (nh + 7u) >> 3;
  int nxt_pos, nxt2_pos;
  uint32_t nxt_vox, nxt2_vox;
  uint4 nxt_q, nxt2_q;
  fetch(nxt_pos, nxt_vox, nxt_q);
  fetch(nxt2_pos, nxt2_vox, nxt2_q);
  for (uint32_t cur = 0; cur < n_it; ++cur) {
    const int my_pos = nxt_pos;
    const bool active = my_pos >= 0;
    const uint32_t hp = active ? (uint32_t)my_pos : lane;
    const uint32_t hvox = nxt_vox;
    const uint4 q = nxt_q;
    nxt_pos = nxt2_pos;
    nxt_vox = nxt2_vox;
    nxt_q = nxt2_q;
    fetch(nxt2_pos, nxt2_vox, nxt2_q);
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
      constexpr bool per_ray_inc = MERGED;
      const float dm_s = per_ray_inc ? perm_f(u.dm, src) : F.log_match;
      const float dn_s = per_ray_inc ? perm_f(u.dn, src) : F.log_non_match;
      const uint32_t info_s = perm_u(u.info, src);
      uint32_t color_s = 0, rp_s = 0;
      if (COLOR_MODE == KS_COLOR_MODE_COLOR) color_s = perm_u(u.color, src);
      if (MERGED) rp_s = perm_u(u.rp, src);
      // The step is straight-line code with selects: instruction issue is one of k_apply's limits (one
      // wave per SIMD slot), and the nested divergent branches of the obvious formulation cost
      // more scalar/branch instructions than the arithmetic they skip.  Every lane evaluates the
      // TSDF recurrence (only sub 0 keeps it) and its four class sums (only subs 1..6 of a
      // pure-label update keep them).
      {
        // updateTsdfVoxel's state half (tsdf_combine), [K:src/semantic_tsdf_integrator_fast.cpp:128]
        const float nw = weight + uw_s;
        const bool upd = on && sub == 0u && !(nw < kEps);
        const float ns = (sdf_s * uw_s + dist * weight) / nw;
        // (ns > 0) ? min(trunc, ns) : max(-trunc, ns) is the median of (-trunc, ns, trunc); v_med3_f32
        // returns min3 of the non-NaN operands for a NaN input = -trunc, which is what the reference's
        // std::max(-trunc, NaN) yields too
        const float nd = __builtin_amdgcn_fmed3f(ns, -F.tsdf.trunc, F.tsdf.trunc);
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
      if (MERGED) {  // mixed-label bundles carry a 21-entry increment vector
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
        const UpdateOps t = load_update_ops<!MERGED && COLOR_MODE != KS_COLOR_MODE_COLOR>(F, rays, k, v);
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
    // ONE unconditional 16-byte store per lane and iteration: idle groups and the spare lane (sub 7)
    // write zeros into spare dwords (their own record's, or record 0's for an idle group).  With the
    // stores under divergent branches the compiler cannot count the memory operations in flight and
    // waits for ALL of them — including the previous iteration's stores — before it touches the
    // prefetched record.
    if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = s_lut[bi & 255u];
    else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) color = rainbow_color_map((double)(float)exp((double)bv));
    uint4 outv = make_uint4(0u, 0u, 0u, 0u);
    if (sub == 0u) outv = make_uint4(__float_as_uint(dist), __float_as_uint(weight), color, bi);
    else if (sub < 6u) outv = make_uint4(__float_as_uint(p0), __float_as_uint(p1), __float_as_uint(p2), __float_as_uint(p3));
    else if (sub == 6u) outv = make_uint4(__float_as_uint(p0), 1u, 0u, 0u);  // dword 25 = 1: updated since the last voxel-level host sync
    const bool wr = active && sub < 7u;
    rec[wr ? sub : 7u] = wr ? outv : make_uint4(0u, 0u, 0u, 0u);
  }
}

// ------------------------------------------------------------------------------------------
// k_apply_runs — the same update as k_apply, ONE LANE PER VOXEL RUN, runs bucketed by length inside a tile of pairs.
//
// Why (round 6, profiles/r06_sq_c4_merged.txt): k_apply is bound by instruction issue, not by HBM.  Eight lanes share a voxel
// and all eight groups of a wavefront step in lockstep to the LONGEST of their runs, 8 heads at a time: ~20 wave
// instructions per update at 1280x720 / 2 cm, of which the arithmetic itself is ~1.5.  Here a lane owns a whole voxel record
// (dist, weight, colour, 21 class sums in registers) and walks its own run; the runs of a tile of 2048 consecutive pairs are
// counting-sorted by length in LDS first, so the 64 lanes of a wavefront walk runs of (nearly) the same length and the
// lockstep costs little.  The state-independent half of every update of the tile (ray gather, computeDistance, weight
// drop-off) is evaluated lane-per-pair, coalesced, all gathers in flight at once, and parked in LDS — the step loop itself
// touches no global memory.
//   phase A   lane per pair of the tile (+ a halo of 32: a run of <= 32 updates that starts in the tile ends there):
//             operands -> LDS (structure of arrays), run boundaries -> a bit per pair
//   listing   lane per pair: a head finds its run's length in the bit mask (distance to the next set bit); runs of more
//             than 32 updates are k_apply_long's / k_apply_xlong's (listed by k_find_long, on their own streams);
//             counting sort by length, longest first
//   tasks     a wavefront takes 64 runs at a time (dynamic, longest first): records in (7 x 16 B per lane), the steps,
//             first-maximum label, colour, records out
// Same operations in the same order per voxel as k_apply: bit for bit the same records.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kRunPer = 4;       // pairs per thread: a tile is 4 x THREADS pairs
constexpr uint32_t kRunThreads = 512; // the default shape: eight wavefronts, tiles of 2048 pairs (~ 6 - 12 tasks of 64 runs); what bounds the kernel is how
                                      // many wavefronts a CU holds against the latency of gather -> record -> steps.  256 (tiles of 1024) is the other
                                      // instantiation: half the LDS and one wavefront per SIMD per workgroup — packs better beside the long-run kernels
constexpr uint32_t kRunTile = kRunPer * kRunThreads;
constexpr uint32_t kRunHalo = 32;     // = kLongRun: the longest run this kernel takes

template <int COLOR_MODE, bool MERGED, uint32_t THREADS = kRunThreads>
__global__ void __launch_bounds__(THREADS) k_apply_runs(FrameParams F, unsigned long long n_pairs,
                                                            const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                            const float* __restrict__ deltas, TileTable T, Pool P,
                                                            const uint32_t* __restrict__ label_lut, uint32_t max_len) {
  // max_len: the longest run this kernel takes (kLongRun, or kLongRunLanes where the lane-per-run long kernel takes over earlier);
  // k_find_long lists exactly the longer ones
  constexpr bool HOT_ONLY = !MERGED && COLOR_MODE != KS_COLOR_MODE_COLOR;
  constexpr bool BLEND = COLOR_MODE == KS_COLOR_MODE_COLOR;
  constexpr uint32_t kTile = kRunPer * THREADS, kSlots = kTile + kRunHalo;
  constexpr uint32_t kMixSlots = kTile / 32;   // merged: increment vectors of mixed-label bundles parked in LDS per tile (more: read from global in the step)
  static_assert(kRunHalo == kLongRun, "a short run must end inside the halo");
  static_assert(kRunHalo <= 64 && THREADS % 64 == 0 && THREADS >= 256, "tile shape");
  __shared__ float s_sdf[kSlots], s_uw[kSlots];
  __shared__ uint16_t s_info[kSlots];                     // (with 16-bit entries and kMixSlots = kTile / 32 a workgroup's LDS is 31 KB: five per CU instead of four)  [7:0] label, [9:8] kind (1 pure, 2 mixed: vector in s_mix, 3 mixed: vector in global memory)
  __shared__ float s_dm[MERGED ? kSlots : 1], s_dn[MERGED ? kSlots : 1];   // merged: per-bundle increments (mixed: s_dm = slot in s_mix / bundle position)
  __shared__ uint32_t s_col[BLEND ? kSlots : 1];
  __shared__ unsigned long long s_bounds[kSlots / 64 + 3 + THREADS / 64];  // bit j: pair j of the tile starts a run, or lies past the end of the list
  __shared__ uint32_t s_run[kTile];                        // the tile's short runs, longest first: start | length << 16
  __shared__ uint32_t s_run_vox[kTile];                    // ... and their voxels
  __shared__ uint32_t s_hist[kLongRun + 2];
  __shared__ uint32_t s_lut[256];
  __shared__ float s_mix[MERGED ? kMixSlots : 1][kNumLabels];
  __shared__ uint32_t s_n_mix, s_next_task;
  const uint32_t lane = lane_id(), tid = threadIdx.x;
  const unsigned long long base = (unsigned long long)blockIdx.x * kTile;
  if (tid < 256u) s_lut[tid] = label_lut[tid];
  if (tid < kLongRun + 2) s_hist[tid] = 0u;
  if (tid == 0) {
    s_n_mix = 0u;
    s_next_task = 0u;
  }
  if (tid < 2) s_bounds[kSlots / 64 + 1 + tid] = ~0ull;

  // ---- phase A1: the tile's keys -> where its runs begin (a bit per pair) ----
  constexpr uint32_t NA = kRunPer + 1;   // the last one is the halo: pairs [kTile, kTile + 32), threads 0..31
  uint64_t key[NA];
  bool valid[NA];
  {
    uint64_t pkey[NA];
#pragma unroll
    for (uint32_t k = 0; k < NA; ++k) {
      const uint32_t j = k * THREADS + tid;
      const unsigned long long i = base + j;
      valid[k] = j < kSlots && i < n_pairs;
      key[k] = valid[k] ? pairs[i] : 0ull;
      pkey[k] = (valid[k] && i > 0) ? pairs[i - 1] : ~0ull;
    }
    __syncthreads();   // (the initial values above)
#pragma unroll
    for (uint32_t k = 0; k < NA; ++k) {
      const uint32_t j = k * THREADS + tid;
      const bool head = valid[k] && ((base + j == 0) || ((uint32_t)(pkey[k] >> F.seq_bits) != (uint32_t)(key[k] >> F.seq_bits)));
      const unsigned long long m = __ballot(head || !valid[k]);
      if (lane == 0) s_bounds[j >> 6] = m;   // (the halo round: wavefront 0 writes word 32, the others all-ones past it)
    }
  }
  __syncthreads();

  // ---- listing: every pair finds the run it is part of — d1 pairs back to its head, d2 on to the next boundary.  A run of
  //      more than 32 updates is k_apply_long's / k_apply_xlong's (k_find_long lists exactly those), a run that began in
  //      the tile before this one is that tile's (its halo).  Heads of the tile's own short runs are counted by length. ----
  uint32_t my_len[kRunPer];
  bool need[NA];
#pragma unroll
  for (uint32_t k = 0; k < NA; ++k) {
    const uint32_t j = k * THREADS + tid;
    need[k] = false;
    if (k < kRunPer) my_len[k] = 0u;
    if (valid[k]) {
      // boundaries at j - 31 .. j (bit 31 = pair j itself) and at j + 1 .. j + 32 (bit 0 = pair j + 1)
      const uint32_t q = j + 1u, sh = q & 63u;
      const unsigned long long lo = s_bounds[q >> 6], hi = s_bounds[(q >> 6) + 1u];
      const uint32_t fwd = (uint32_t)(sh ? ((lo >> sh) | (hi << (64u - sh))) : lo);
      uint32_t back;
      {
        // 32 bits ending at j: positions before the tile's first pair hold no boundary (a run from before the tile: not ours)
        const int p0 = (int)j - 31;
        const uint32_t w0 = (uint32_t)(p0 >= 0 ? p0 : 0) >> 6;
        const unsigned long long a0 = s_bounds[w0], a1 = s_bounds[w0 + 1u];
        if (p0 >= 0) {
          const uint32_t s0 = (uint32_t)p0 & 63u;
          back = (uint32_t)(s0 ? ((a0 >> s0) | (a1 << (64u - s0))) : a0);
        } else {
          back = (uint32_t)(a0 << (uint32_t)(-p0));   // (j < 31: word 0 holds them all)
        }
      }
      if (back != 0u && fwd != 0u) {
        const uint32_t d1 = (uint32_t)__clz((int)back);        // pairs back to the head (0: j is the head)
        const uint32_t d2 = (uint32_t)__ffs((int)fwd);         // pairs on to the next boundary
        const uint32_t len = d1 + d2;
        const bool head_in_tile = j - d1 < kTile;
        if (len <= max_len && head_in_tile) {
          need[k] = true;
          if (d1 == 0u && k < kRunPer) {
            my_len[k] = len;
            atomicAdd(&s_hist[len], 1u);
          }
        }
      }
    }
  }

  // ---- phase A2: the operands of the pairs that are part of this tile's short runs (at 1280x720 / 2 cm most pairs are
  //      part of longer ones): every gather is requested before the first is used ----
  uint64_t tkey[NA];
  uint4 d0[NA], d1v[NA];
#pragma unroll
  for (uint32_t k = 0; k < NA; ++k) {
    const uint32_t vox = (uint32_t)(key[k] >> F.seq_bits);
    const uint32_t rp = (uint32_t)key[k] & F.point_mask;
    const uint4* r4 = (const uint4*)rays + (size_t)(need[k] ? ray_index(F, rp) : 0u) * 2;
    if (need[k]) {
      tkey[k] = T.slot_keys[vox >> 9];
      d0[k] = r4[0];
      if (!HOT_ONLY) d1v[k] = r4[1];
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < NA; ++k) {
    const uint32_t j = k * THREADS + tid;
    if (need[k]) {
      const uint32_t vox = (uint32_t)(key[k] >> F.seq_bits);
      int tx, ty, tz;
      unpack_tile(tkey[k], tx, ty, tz);
      const uint32_t local = vox & 511u;
      float sdf, uw;
      tsdf_operands(F.tsdf, F.T.t, {__uint_as_float(d0[k].x), __uint_as_float(d0[k].y), __uint_as_float(d0[k].z)},
                    tx * 8 + (int)(local & 7u), ty * 8 + (int)((local >> 3) & 7u), tz * 8 + (int)(local >> 6), __uint_as_float(d0[k].w), sdf, uw);
      s_sdf[j] = sdf;
      s_uw[j] = uw;
      uint32_t info;
      if (HOT_ONLY) {   // (load_update_ops: label / kind / clearing ride in the top byte of the pair key)
        const uint32_t b = (uint32_t)(key[k] >> 56);
        info = (b & 0x1fu) | (((b >> 5) & 3u) << 8);
      } else {
        info = d1v[k].w & 0x3ffu;
      }
      if (MERGED) {
        float dm = __uint_as_float(d1v[k].y);
        if (((info >> 8) & 3u) == 2u) {
          // mixed-label bundle: its 21 increments wait in LDS for the step that needs them
          const uint32_t rp = (uint32_t)key[k] & F.point_mask;
          const uint32_t slot = atomicAdd(&s_n_mix, 1u);
          const float* dl = deltas + (size_t)rp * kNumLabels;
          if (slot < kMixSlots) {
#pragma unroll
            for (int l = 0; l < kNumLabels; ++l) s_mix[slot][l] = dl[l];
            dm = __uint_as_float(slot);
          } else {
            info |= 0x300u;   // kind 3
            dm = __uint_as_float(rp);
          }
        }
        s_dm[j] = dm;
        s_dn[j] = __uint_as_float(d1v[k].z);
      }
      if (BLEND) s_col[j] = d1v[k].x;
      s_info[j] = (uint16_t)info;
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0u;
    for (uint32_t l = kLongRun; l >= 1u; --l) {
      const uint32_t c = s_hist[l];
      s_hist[l] = acc;
      acc += c;
    }
    s_hist[0] = acc;   // runs in all
  }
  __syncthreads();
  const uint32_t n_runs = s_hist[0];
  __syncthreads();     // (s_hist[0] read by everyone before the cursors move)
#pragma unroll
  for (uint32_t k = 0; k < kRunPer; ++k)
    if (my_len[k]) {
      const uint32_t pos = atomicAdd(&s_hist[my_len[k]], 1u);
      s_run[pos] = (k * THREADS + tid) | (my_len[k] << 16);
      s_run_vox[pos] = (uint32_t)(key[k] >> F.seq_bits);
    }
  __syncthreads();

  // ---- tasks ----
  const uint32_t n_tasks = (n_runs + 63u) >> 6;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&s_next_task, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= n_tasks) break;
    const uint32_t r = t * 64u + lane;
    const bool has = r < n_runs;
    const uint32_t rj = has ? s_run[r] : 0u;
    const uint32_t j0 = rj & 0xffffu, len = has ? (rj >> 16) : 0u;
    const uint32_t vox = has ? s_run_vox[r] : 0u;
    uint4* rec = P.vox + (size_t)vox * 8;
    float p[kNumLabels];
    float dist = 0.f, weight = 0.f;
    uint32_t color = 0u;
    if (has) {
      const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5];
      const uint32_t q6 = ((const uint32_t*)rec)[24];
      dist = __uint_as_float(q0.x);
      weight = __uint_as_float(q0.y);
      color = q0.z;
      p[0] = __uint_as_float(q1.x); p[1] = __uint_as_float(q1.y); p[2] = __uint_as_float(q1.z); p[3] = __uint_as_float(q1.w);
      p[4] = __uint_as_float(q2.x); p[5] = __uint_as_float(q2.y); p[6] = __uint_as_float(q2.z); p[7] = __uint_as_float(q2.w);
      p[8] = __uint_as_float(q3.x); p[9] = __uint_as_float(q3.y); p[10] = __uint_as_float(q3.z); p[11] = __uint_as_float(q3.w);
      p[12] = __uint_as_float(q4.x); p[13] = __uint_as_float(q4.y); p[14] = __uint_as_float(q4.z); p[15] = __uint_as_float(q4.w);
      p[16] = __uint_as_float(q5.x); p[17] = __uint_as_float(q5.y); p[18] = __uint_as_float(q5.z); p[19] = __uint_as_float(q5.w);
      p[20] = __uint_as_float(q6);
    } else {
#pragma unroll
      for (int l = 0; l < kNumLabels; ++l) p[l] = 0.f;
    }
    // the operands of step s + 1 are requested from LDS before step s is applied
    float sdf_n = s_sdf[j0], uw_n = s_uw[j0];
    uint32_t info_n = s_info[j0];
    float dm_n = MERGED ? s_dm[j0] : F.log_match, dn_n = MERGED ? s_dn[j0] : F.log_non_match;
    uint32_t col_n = BLEND ? s_col[j0] : 0u;
    for (uint32_t s = 0;; ++s) {
      const bool on = s < len;
      if (__ballot(on) == 0ull) break;
      const float sdf_s = sdf_n, uw_s = uw_n, dm_s = dm_n, dn_s = dn_n;
      const uint32_t info_s = info_n, col_s = col_n;
      const uint32_t jn = j0 + ((s + 1u < len) ? s + 1u : 0u);
      sdf_n = s_sdf[jn];
      uw_n = s_uw[jn];
      info_n = s_info[jn];
      if (MERGED) {
        dm_n = s_dm[jn];
        dn_n = s_dn[jn];
      }
      if (BLEND) col_n = s_col[jn];
      if (on) {
        // updateTsdfVoxel's state half (tsdf_combine), [K:src/semantic_tsdf_integrator_fast.cpp:128]
        const float nw = weight + uw_s;
        if (!(nw < kEps)) {
          const float ns = (sdf_s * uw_s + dist * weight) / nw;
          if (BLEND) {
            if (fabsf(sdf_s) < F.tsdf.trunc) color = blend_two_colors(color, weight, col_s, uw_s);
          }
          dist = __builtin_amdgcn_fmed3f(ns, -F.tsdf.trunc, F.tsdf.trunc);   // (k_apply: the median IS the reference's clamp, NaN included)
          weight = std_min(F.tsdf.max_weight, nw);
        }
        // updateSemanticVoxelProbabilities, [K:src/semantic_integrator_base.cpp:283-380]
        const uint32_t kind = (info_s >> 8) & 3u, lab = info_s & 0xffu;
        if (kind == 1u) {
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) p[l] += ((uint32_t)l == lab) ? dm_s : dn_s;
        } else if (MERGED && kind == 2u) {
          const float* dl = s_mix[__float_as_uint(dm_s)];
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) p[l] += dl[l];
        } else if (MERGED && kind == 3u) {
          const float* dl = deltas + (size_t)__float_as_uint(dm_s) * kNumLabels;
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) p[l] += dl[l];
        }
      }
    }
    if (has) {
      // calculateMaximumLikelihoodLabel: first strict maximum [K:src/semantic_integrator_base.cpp:352-367]
      float bv = p[0];
      uint32_t bi = 0u;
#pragma unroll
      for (int l = 1; l < kNumLabels; ++l)
        if (p[l] > bv) {
          bv = p[l];
          bi = (uint32_t)l;
        }
      if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = s_lut[bi & 255u];
      else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) color = rainbow_color_map((double)(float)exp((double)bv));
      rec[0] = make_uint4(__float_as_uint(dist), __float_as_uint(weight), color, bi);
      rec[1] = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), __float_as_uint(p[3]));
      rec[2] = make_uint4(__float_as_uint(p[4]), __float_as_uint(p[5]), __float_as_uint(p[6]), __float_as_uint(p[7]));
      rec[3] = make_uint4(__float_as_uint(p[8]), __float_as_uint(p[9]), __float_as_uint(p[10]), __float_as_uint(p[11]));
      rec[4] = make_uint4(__float_as_uint(p[12]), __float_as_uint(p[13]), __float_as_uint(p[14]), __float_as_uint(p[15]));
      rec[5] = make_uint4(__float_as_uint(p[16]), __float_as_uint(p[17]), __float_as_uint(p[18]), __float_as_uint(p[19]));
      rec[6] = make_uint4(__float_as_uint(p[20]), 1u, 0u, 0u);   // dword 25 = 1: updated since the last voxel-level host sync
    }
  }
}

// Heads of runs of >= kLongRun updates, found before either apply kernel runs: k_apply (short runs) and
// k_apply_long (long runs) touch disjoint voxels and are launched side by side on two streams.  Runs of more than
// kXLongRun updates go to a list of their own when the caller passes one (xlong_list != nullptr): the handful of voxels
// next to the sensor, whose single chains of 1e4 .. 1e5 updates bound the frame's update stage (k_apply_xlong).
constexpr uint32_t kFindLongItems = 8;  // pairs per thread
__global__ void __launch_bounds__(256) k_find_long(uint32_t seq_bits, unsigned long long n_pairs,
                                                   const uint64_t* __restrict__ pairs,
                                                   unsigned long long* __restrict__ long_list,
                                                   unsigned long long* __restrict__ xlong_list, Counters* C, uint32_t long_min = kLongRun) {
  // long_min: runs of MORE than long_min updates are listed (kLongRun, or kLongRunLanes)
  // heads of long runs are more than kLongRun apart: at most 2048 / 33 + 1 of them per workgroup
  __shared__ unsigned long long s_list[2048 / kLongRunLanes + 2];
  __shared__ unsigned long long s_xlist[2048 / kXLongRun + 2];
  __shared__ uint32_t s_n, s_base, s_xn, s_xbase;
  static_assert(kLongRun >= kLongRunLanes, "s_list size");
  if (threadIdx.x == 0) {
    s_n = 0u;
    s_xn = 0u;
  }
  __syncthreads();
  const unsigned long long base = (unsigned long long)blockIdx.x * (256ull * kFindLongItems);
#pragma unroll
  for (uint32_t k = 0; k < kFindLongItems; ++k) {
    const unsigned long long i = base + k * 256ull + threadIdx.x;
    if (i < n_pairs) {
      const uint32_t vox = (uint32_t)(pairs[i] >> seq_bits);
      const bool head = (i == 0) || ((uint32_t)(pairs[i - 1] >> seq_bits) != vox);
      if (head && (i + long_min < n_pairs) && ((uint32_t)(pairs[i + long_min] >> seq_bits) == vox)) {
        const bool xl = xlong_list != nullptr && (i + kXLongRun < n_pairs) && ((uint32_t)(pairs[i + kXLongRun] >> seq_bits) == vox);
        if (xl) s_xlist[atomicAdd(&s_xn, 1u)] = i;
        else s_list[atomicAdd(&s_n, 1u)] = i;
      }
    }
  }
  __syncthreads();
  const uint32_t n_l = s_n, n_x = s_xn;
  if (n_l == 0u && n_x == 0u) return;
  if (threadIdx.x == 0) {
    if (n_l) s_base = atomicAdd(&C->n_long, n_l);
    if (n_x) s_xbase = atomicAdd(&C->n_xlong, n_x);
  }
  __syncthreads();
  if (threadIdx.x < n_l) long_list[s_base + threadIdx.x] = s_list[threadIdx.x];
  if (threadIdx.x < n_x) xlong_list[s_xbase + threadIdx.x] = s_xlist[threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// The runs of 33 .. kXLongRun updates, A LANE PER RUN (round 6).  At 1280x720 / 2 cm they hold most of a frame's updates
// (every voxel within a few decimetres of the sensor), and k_apply_long spends two wavefronts on each of them — ~8 wave
// instructions per update.  Here the runs are bucketed by length over the whole frame (ten classes, lengths within a class
// within a factor 1.5), 64 runs of a class share a wavefront, and every lane walks its own run with its voxel's record in
// registers: ~4 wave instructions per update, nothing serial beyond a run's own 33 .. 1024 steps.
//   k_long_measure   thread per listed run (k_find_long lists where they start): its length by bisection, class counts
//   k_long_bucket    the list again, class by class (offsets from the ten counts; cursors per class)
//   k_apply_long_lanes  wavefront per 64 runs of a class; per step and lane: the pair key (the lane's own stream, two steps
//                    ahead), its ray (one step ahead), computeDistance + weight, the voxel-state update
// Same operations in the same order per voxel as k_apply_long.
// ------------------------------------------------------------------------------------------
constexpr int kLongClasses = 12;
constexpr int kLongLaneClasses = 8;   // classes 0 .. 7 (17 .. 256 updates) a lane per run; the longer ones two wavefronts per run (k_apply_long)
struct LongHdr {
  uint32_t count[kLongClasses];    // runs per class (k_long_measure)
  uint32_t cursor[kLongClasses];   // k_long_bucket's
  uint32_t pad[8];
};
__device__ __forceinline__ int long_class(uint32_t len) {   // len in kLongRunLanes + 1 .. kXLongRun
  return len <= 24u ? 0 : len <= 32u ? 1 : len <= 48u ? 2 : len <= 64u ? 3 : len <= 96u ? 4 : len <= 128u ? 5 : len <= 192u ? 6 : len <= 256u ? 7 :
         len <= 384u ? 8 : len <= 512u ? 9 : len <= 768u ? 10 : 11;
}

__global__ void __launch_bounds__(256) k_long_measure(uint32_t seq_bits, unsigned long long n_pairs, const uint64_t* __restrict__ pairs,
                                                      unsigned long long* __restrict__ long_list, const Counters* C, LongHdr* __restrict__ H,
                                                      uint32_t long_min) {
  __shared__ uint32_t s_cnt[kLongClasses];
  if (threadIdx.x < kLongClasses) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t n_long = C->n_long, r = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x * 256u >= n_long) return;
  if (r < n_long) {
    const unsigned long long start = long_list[r];
    const uint32_t vox = (uint32_t)(pairs[start] >> seq_bits);
    // element long_min is this voxel's, element kXLongRun (if it exists) is not (k_find_long): bisect in between
    unsigned long long lo = long_min, hi = kXLongRun;
    if (start + hi > n_pairs) hi = n_pairs - start;
    // invariant: element lo belongs to the run, element hi does not (or is the end of the list)
    while (hi - lo > 1ull) {
      const unsigned long long mid = lo + ((hi - lo) >> 1);
      if ((uint32_t)(pairs[start + mid] >> seq_bits) == vox) lo = mid;
      else hi = mid;
    }
    const uint32_t len = (uint32_t)hi;
    long_list[r] = start | ((unsigned long long)len << 48);
    atomicAdd(&s_cnt[long_class(len)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kLongClasses && s_cnt[threadIdx.x]) atomicAdd(&H->count[threadIdx.x], s_cnt[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_long_bucket(const unsigned long long* __restrict__ long_list, const Counters* C,
                                                     LongHdr* __restrict__ H, unsigned long long* __restrict__ sorted) {
  __shared__ uint32_t s_cnt[kLongClasses], s_base[kLongClasses];
  if (threadIdx.x < kLongClasses) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t n_long = C->n_long, r = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x * 256u >= n_long) return;
  unsigned long long e = 0ull;
  int c = 0;
  uint32_t rank = 0u;
  if (r < n_long) {
    e = long_list[r];
    c = long_class((uint32_t)(e >> 48));
    rank = atomicAdd(&s_cnt[c], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kLongClasses) {
    uint32_t off = 0u;
    for (int k = 0; k < (int)threadIdx.x; ++k) off += H->count[k];
    s_base[threadIdx.x] = off + (s_cnt[threadIdx.x] ? atomicAdd(&H->cursor[threadIdx.x], s_cnt[threadIdx.x]) : 0u);
  }
  __syncthreads();
  if (r < n_long) sorted[s_base[c] + rank] = e;
}

template <int COLOR_MODE, uint32_t D = 6>
__global__ void __launch_bounds__(256) k_apply_long_lanes(FrameParams F, const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                          const float* __restrict__ deltas, TileTable T, Pool P,
                                                          const uint32_t* __restrict__ label_lut, const LongHdr* __restrict__ H,
                                                          const unsigned long long* __restrict__ sorted) {
  constexpr bool BLEND = COLOR_MODE == KS_COLOR_MODE_COLOR;
  __shared__ uint32_t s_lut[256];
  s_lut[threadIdx.x] = label_lut[threadIdx.x];
  __syncthreads();
  // this wavefront's 64 runs: wavefronts are dealt class by class, ceil(count / 64) each
  const uint32_t lane = lane_id();
  uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), first = 0u, n_mine = 0u;
  bool found = false;
#pragma unroll
  for (int c = 0; c < kLongLaneClasses; ++c) {
    const uint32_t cnt = H->count[c], waves = (cnt + 63u) >> 6;
    if (!found) {
      if (w < waves) {
        found = true;
        first += w * 64u;
        n_mine = cnt - w * 64u < 64u ? cnt - w * 64u : 64u;
      } else {
        w -= waves;
        first += cnt;
      }
    }
  }
  if (!found) return;
  const bool has = lane < n_mine;
  const unsigned long long e = has ? sorted[first + lane] : 0ull;
  const unsigned long long start = e & 0xffffffffffffull;
  const uint32_t len = has ? (uint32_t)(e >> 48) : 0u;
  // A lane's step is a chain key -> ray -> arithmetic, and a run is up to 1024 of them: the keys of the next 2 D steps and the
  // rays of the next D are in flight while a step is applied (with one step of look-ahead a step cost the latency of its
  // gather: 0.85 us — a 640x480 frame waited 0.9 ms for its handful of 1000-update runs).  Compile-time slots: the step loop
  // is unrolled D times.
  uint64_t kq[2 * D];
  RayDesc dq[D];
#pragma unroll
  for (uint32_t i = 0; i < 2 * D; ++i) kq[i] = (i < len) ? pairs[start + i] : 0ull;
  const uint32_t vox = (uint32_t)(kq[0] >> F.seq_bits);
  uint4* rec = P.vox + (size_t)vox * 8;
#pragma unroll
  for (uint32_t i = 0; i < D; ++i) dq[i] = rays[(i < len) ? ray_index(F, (uint32_t)kq[i] & F.point_mask) : 0u];
  float p[kNumLabels];
  float dist = 0.f, weight = 0.f;
  uint32_t color = 0u;
  f3 v_voxel_origin = {0.f, 0.f, 0.f};
  if (has) {
    const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5];
    const uint32_t q6 = ((const uint32_t*)rec)[24];
    dist = __uint_as_float(q0.x);
    weight = __uint_as_float(q0.y);
    color = q0.z;
    p[0] = __uint_as_float(q1.x); p[1] = __uint_as_float(q1.y); p[2] = __uint_as_float(q1.z); p[3] = __uint_as_float(q1.w);
    p[4] = __uint_as_float(q2.x); p[5] = __uint_as_float(q2.y); p[6] = __uint_as_float(q2.z); p[7] = __uint_as_float(q2.w);
    p[8] = __uint_as_float(q3.x); p[9] = __uint_as_float(q3.y); p[10] = __uint_as_float(q3.z); p[11] = __uint_as_float(q3.w);
    p[12] = __uint_as_float(q4.x); p[13] = __uint_as_float(q4.y); p[14] = __uint_as_float(q4.z); p[15] = __uint_as_float(q4.w);
    p[16] = __uint_as_float(q5.x); p[17] = __uint_as_float(q5.y); p[18] = __uint_as_float(q5.z); p[19] = __uint_as_float(q5.w);
    p[20] = __uint_as_float(q6);
    const VoxelRef v = voxel_ref(T, vox);
    const float vs = F.tsdf.voxel_size;
    v_voxel_origin = sub3({((float)v.vx + 0.5f) * vs, ((float)v.vy + 0.5f) * vs, ((float)v.vz + 0.5f) * vs}, F.T.t);
  } else {
#pragma unroll
    for (int l = 0; l < kNumLabels; ++l) p[l] = 0.f;
  }
  const TsdfParams& Pm = F.tsdf;
  for (uint32_t g = 0; __ballot(g < len) != 0ull; g += D) {
#pragma unroll
    for (uint32_t j = 0; j < D; ++j) {
      const uint32_t s = g + j;
      const bool on = s < len;
      const RayDesc d = dq[j];
      const uint32_t rp = (uint32_t)kq[j] & F.point_mask;
      // the slot takes the ray of step s + D (its key arrived a group ago)
      dq[j] = rays[(s + D < len) ? ray_index(F, (uint32_t)kq[j + D] & F.point_mask) : 0u];
      if (on) {
        // computeDistance + weight drop-off (tsdf_operands with the voxel's origin vector hoisted)
        const f3 v_point_origin = sub3({d.px, d.py, d.pz}, F.T.t);
        const float dist_G = norm3(v_point_origin);
        const float dist_G_V = dot3(v_voxel_origin, v_point_origin) / dist_G;
        const float sdf = dist_G - dist_G_V;
        float uw = d.weight;
        if (Pm.use_dropoff && sdf < -Pm.voxel_size) {
          uw = d.weight * (Pm.trunc + sdf) / Pm.dropoff_denominator;
          uw = std_max(uw, 0.0f);
        }
        if (Pm.use_sparsity) {
          if (fabsf(sdf) < Pm.trunc) uw *= Pm.sparsity_factor;
        }
        // updateTsdfVoxel's state half
        const float nw = weight + uw;
        if (!(nw < kEps)) {
          const float ns = (sdf * uw + dist * weight) / nw;
          if (BLEND) {
            if (fabsf(sdf) < Pm.trunc) color = blend_two_colors(color, weight, d.color, uw);
          }
          dist = __builtin_amdgcn_fmed3f(ns, -Pm.trunc, Pm.trunc);
          weight = std_min(Pm.max_weight, nw);
        }
        const uint32_t kind = (d.info >> 8) & 3u, lab = d.info & 0xffu;
        if (kind == 1u) {
          const float dm = d.d_match, dn = d.d_non;   // (as k_apply_long reads them)
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) p[l] += ((uint32_t)l == lab) ? dm : dn;
        } else if (kind == 2u) {
          const float* dl = deltas + (size_t)rp * kNumLabels;
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) p[l] += dl[l];
        }
      }
    }
    // the keys move up a group; the keys of the group after the next are requested
#pragma unroll
    for (uint32_t j = 0; j < D; ++j) {
      kq[j] = kq[j + D];
      kq[j + D] = (g + 2 * D + j < len) ? pairs[start + g + 2 * D + j] : 0ull;
    }
  }
  if (has) {
    float bv = p[0];
    uint32_t bi = 0u;
#pragma unroll
    for (int l = 1; l < kNumLabels; ++l)
      if (p[l] > bv) {
        bv = p[l];
        bi = (uint32_t)l;
      }
    if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = s_lut[bi & 255u];
    else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) color = rainbow_color_map((double)(float)exp((double)bv));
    rec[0] = make_uint4(__float_as_uint(dist), __float_as_uint(weight), color, bi);
    rec[1] = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), __float_as_uint(p[3]));
    rec[2] = make_uint4(__float_as_uint(p[4]), __float_as_uint(p[5]), __float_as_uint(p[6]), __float_as_uint(p[7]));
    rec[3] = make_uint4(__float_as_uint(p[8]), __float_as_uint(p[9]), __float_as_uint(p[10]), __float_as_uint(p[11]));
    rec[4] = make_uint4(__float_as_uint(p[12]), __float_as_uint(p[13]), __float_as_uint(p[14]), __float_as_uint(p[15]));
    rec[5] = make_uint4(__float_as_uint(p[16]), __float_as_uint(p[17]), __float_as_uint(p[18]), __float_as_uint(p[19]));
    rec[6] = make_uint4(__float_as_uint(p[20]), 1u, 0u, 0u);   // dword 25 = 1: updated since the last voxel-level host sync
  }
}

template <int COLOR_MODE>
__global__ void __launch_bounds__(128) k_apply_long(FrameParams F, unsigned long long n_pairs,
                                                   const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                   const float* __restrict__ deltas, TileTable T, Pool P,
                                                   const uint32_t* __restrict__ label_lut,
                                                   const unsigned long long* __restrict__ long_list, const Counters* C,
                                                   const LongHdr* __restrict__ H = nullptr) {
  // TWO wavefronts per run.  A lone wave is bound by instruction issue (one instruction every four
  // cycles): wave 0 (producer) gathers the rays of a 64-update batch, evaluates the state-
  // independent half of the update per lane, walks the weight / distance recurrences and writes
  // the 64 x 21 class increments to LDS; wave 1 (consumer) owns the 21 class sums and folds the
  // increments of the PREVIOUS batch in, in order, at the same time.  One workgroup barrier per
  // batch hands a double-buffered increment block over.
  __shared__ float s_inc[2][64][kNumLabels];  // class increments of the 64 updates of a batch
  __shared__ int s_cnt[2];                    // updates in the batch (0: the run has ended)
  __shared__ uint32_t s_best;
  __shared__ float s_best_val;
  constexpr int PF = 4;  // batches (of 64 updates) whose ray descriptors are in flight while one batch is applied (8 / 16: measured no faster)
  // H != nullptr: `long_list` is k_long_bucket's list, by length class; this kernel takes the classes from kLongLaneClasses on
  // (the runs of more than 256 updates: a lane per run would walk them for up to 0.6 ms), k_apply_long_lanes the others
  uint32_t n_long = C->n_long, list_base = 0u;
  if (H) {
    n_long = 0u;
    for (int c = 0; c < kLongClasses; ++c) {
      if (c < kLongLaneClasses) list_base += H->count[c];
      else n_long += H->count[c];
    }
  }
  const int lane = (int)lane_id();
  const bool consumer = (threadIdx.x >> 6) != 0u;
  const int cls = lane < kNumLabels ? lane : 0;
  const TsdfParams& Pm = F.tsdf;
  for (uint32_t run = blockIdx.x; run < n_long; run += gridDim.x) {
    const unsigned long long start = long_list[list_base + run] & 0xffffffffffffull;   // (bits 48..: the length, in k_long_bucket's list)
    const uint32_t vox = (uint32_t)(pairs[start] >> F.seq_bits);
    const VoxelRef v = voxel_ref(T, vox);
    uint32_t* rec = (uint32_t*)(P.vox + (size_t)vox * 8);
    float dist = __uint_as_float(rec[0]), weight = __uint_as_float(rec[1]);
    uint32_t color = rec[2];
    if (consumer) {
      // ---- wave 1: the semantic log-likelihood sums, lane l owns class l ----
      float pri = (lane < kNumLabels) ? __uint_as_float(rec[4 + lane]) : 0.0f;
      for (int buf = 0;; buf ^= 1) {
        __syncthreads();  // batch `buf` is complete
        const int cnt = s_cnt[buf];
        if (cnt == 0) break;
        if (cnt == 64) {
          // full batch: all 64 increments are requested from LDS before the first dependent add
          float x[64];
#pragma unroll
          for (int k = 0; k < 64; ++k) x[k] = s_inc[buf][k][cls];
#pragma unroll
          for (int k = 0; k < 64; ++k) pri += x[k];
        } else {
#pragma unroll 8
          for (int k = 0; k < cnt; ++k) pri += s_inc[buf][k][cls];
        }
        if (cnt < 64) break;
      }
      // argmax over lanes 0..20, first strict maximum
      int best = 0;
      float m = bcast_f(pri, 0);
#pragma unroll
      for (int l = 1; l < kNumLabels; ++l) {
        const float x = bcast_f(pri, l);
        if (x > m) { m = x; best = l; }
      }
      if (lane < kNumLabels) rec[4 + lane] = __float_as_uint(pri);
      if (lane == 0) {
        s_best = (uint32_t)best;
        s_best_val = m;
      }
      __syncthreads();  // label for the producer's record head
      __syncthreads();  // LDS free for the next run
      continue;
    }
    // ---- wave 0: everything else ----
    int buf = 0;
    // voxel centre and the origin->centre vector are constant over the run
    const f3 c = {((float)v.vx + 0.5f) * Pm.voxel_size, ((float)v.vy + 0.5f) * Pm.voxel_size,
                  ((float)v.vz + 0.5f) * Pm.voxel_size};
    const f3 v_voxel_origin = sub3(c, F.T.t);

    // software pipeline: the ray descriptors of the next PF batches and the pair keys of the PF after them are in
    // flight while batch b is applied (a batch is applied in well under the latency of its random 32-byte gathers:
    // with one batch of look-ahead the voxel next to the sensor — one run of ~1e4 .. 1e5 updates — ran at memory
    // latency per 64 updates).  All loads of the pipeline are UNCONDITIONAL (indices clamped): a load under a
    // divergent branch makes the compiler drain vmcnt at the join, which serialises the prefetch.  The queue is a
    // ring with compile-time slot numbers (the batch loop is unrolled PF times): nothing moves between registers.
    const unsigned long long last = n_pairs - 1ull;
    unsigned long long base = start;
    uint64_t key_q[PF], key_n[PF];  // key_q[j]: the batch in slot j; key_n[j]: the batch PF after it
    RayDesc d_q[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      key_q[i] = pairs[min(base + 64ull * i + lane, last)];
      key_n[i] = pairs[min(base + 64ull * (PF + i) + lane, last)];
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) d_q[i] = rays[ray_index(F, (uint32_t)key_q[i] & F.point_mask)];

    // one batch; returns false when the run has ended
    auto batch = [&](const uint64_t key_cur, const RayDesc& d) -> bool {
      const bool in = (base + lane < n_pairs) && ((uint32_t)(key_cur >> F.seq_bits) == vox);
      const int cnt = (int)__popcll(__ballot(in));  // sorted => the in-lanes form a prefix
      if (cnt == 0) {
        if (lane == 0) s_cnt[buf] = 0;
        __syncthreads();
        return false;
      }
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
          for (int l = 0; l < kNumLabels; ++l) s_inc[buf][lane][l] = dl[l];
        } else {
          const float a = (kind == 1u) ? d.d_match : 0.0f, b = (kind == 1u) ? d.d_non : 0.0f;
#pragma unroll
          for (int l = 0; l < kNumLabels; ++l) s_inc[buf][lane][l] = ((uint32_t)l == lab) ? a : b;
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
      // ---- hand the increments of this batch to the consumer wave ----
      if (lane == 0) s_cnt[buf] = cnt;
      __syncthreads();
      buf ^= 1;
      return cnt == 64;
    };
    for (bool more = true; more;) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const uint64_t key_cur = key_q[j];
        const RayDesc d = d_q[j];
        // slot j takes the batch PF after the one it held: its keys arrived a ring ago, its descriptors are requested
        // now, and the keys of the batch after that one follow
        key_q[j] = key_n[j];
        d_q[j] = rays[ray_index(F, (uint32_t)key_n[j] & F.point_mask)];
        key_n[j] = pairs[min(base + 64ull * (2 * PF) + lane, last)];
        more = batch(key_cur, d);
        if (!more) break;
        base += 64;
      }
    }
    __syncthreads();  // the consumer has the label
    const uint32_t best = s_best;
    const float m = s_best_val;
    if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = label_lut[best];
    else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY)
      color = rainbow_color_map((double)(float)exp((double)m));
    if (lane == 0) {
      *(uint4*)rec = make_uint4(__float_as_uint(dist), __float_as_uint(weight), color, best);
      rec[25] = 1u;  // updated since the last voxel-level host sync
    }
    __syncthreads();  // LDS free for the next run
  }
}

// The runs of more than kXLongRun updates — the handful of voxels next to the sensor, ONE chain of 1e4 .. 1e5 updates each
// in `merged` (every bundle's ray starts there) — bound the update stage, and what bounds such a chain is the number of
// instructions ONE wave issues per 64 updates (k_apply_long's producer: ~350).  Here the work of a batch is spread over
// FOUR waves of a workgroup that meet at one barrier per batch ("step"):
//   waves 2, 3  take the batches alternately.  A batch is prepared over two steps: first the gather of its ray descriptors
//               (requested four batches ahead) and the state-independent half of the TSDF update per lane — sdf, weight —
//               into a table in LDS; in the next step its 64 x 21 class increments (laid out so that the consumer reads the
//               increments of four consecutive updates of one class with one 16-byte load);
//   wave 0      walks the weight / distance recurrences of the batch prepared two steps ago (operands from the table);
//   wave 1      folds that batch's increments into the 21 class sums, in order.
// Same operations in the same order as k_apply_long: the voxel's record is bit for bit what the single chain leaves.
template <int COLOR_MODE>
__global__ void __launch_bounds__(256) k_apply_xlong(FrameParams F, unsigned long long n_pairs,
                                                    const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                    const float* __restrict__ deltas, TileTable T, Pool P,
                                                    const uint32_t* __restrict__ label_lut,
                                                    const unsigned long long* __restrict__ xlong_list, const uint32_t* __restrict__ n_runs_ptr) {
  __shared__ float s_inc[2][16][kNumLabels][4];  // [batch parity][update / 4][class][update % 4]
  __shared__ float4 s_tab[3][64];                // per update of a batch: sdf, update weight, colour
  __shared__ int s_cnt[3];                       // updates in the batch (< 64: the run ends with it)
  __shared__ uint32_t s_best;
  __shared__ float s_best_val;
  const uint32_t n_x = *n_runs_ptr;   // (C->n_xlong, or what k_xl_plan / k_xl_walk left for this kernel: ks_k_apply_xl.h)
  const int lane = (int)lane_id();
  const int role = (int)(threadIdx.x >> 6);
  const int cls = lane < kNumLabels ? lane : 0;
  const TsdfParams& Pm = F.tsdf;
  for (uint32_t run = blockIdx.x; run < n_x; run += gridDim.x) {
    const unsigned long long start = xlong_list[run];
    const uint32_t vox = (uint32_t)(pairs[start] >> F.seq_bits);
    uint32_t* rec = (uint32_t*)(P.vox + (size_t)vox * 8);
    if (role >= 2) {
      // ---- waves 2, 3: the batches f, f + 2, f + 4, ... ----
      const int f = role - 2;
      const VoxelRef v = voxel_ref(T, vox);
      const f3 c = {((float)v.vx + 0.5f) * Pm.voxel_size, ((float)v.vy + 0.5f) * Pm.voxel_size,
                    ((float)v.vz + 0.5f) * Pm.voxel_size};
      const f3 v_voxel_origin = sub3(c, F.T.t);
      const unsigned long long last = n_pairs - 1ull;
      // own batch j is batch f + 2 j of the run.  Descriptors of two own batches in flight, keys of the two after them
      // (all loads unconditional, indices clamped: ks_k_apply.h, k_apply_long)
      auto key_of = [&](unsigned long long own) { return pairs[min(start + 64ull * ((unsigned long long)f + 2ull * own) + lane, last)]; };
      uint64_t key_q[2] = {key_of(0), key_of(1)}, key_n[2] = {key_of(2), key_of(3)};
      RayDesc d_q[2] = {rays[ray_index(F, (uint32_t)key_q[0] & F.point_mask)], rays[ray_index(F, (uint32_t)key_q[1] & F.point_mask)]};
      unsigned long long own = 0;  // the own batch whose first half comes next
      // what the second half of a batch needs of the first
      bool h_in = false;
      uint32_t h_kind = 0, h_lab = 0, h_rp = 0;
      float h_a = 0.f, h_b = 0.f;
      auto first_half = [&]() {
        const unsigned long long b = (unsigned long long)f + 2ull * own;
        const uint64_t key_cur = key_q[0];
        const RayDesc d = d_q[0];
        const RayDesc d_new = rays[ray_index(F, (uint32_t)key_n[0] & F.point_mask)];
        const uint64_t key_new = key_of(own + 4ull);
        const bool in = (start + 64ull * b + lane < n_pairs) && ((uint32_t)(key_cur >> F.seq_bits) == vox);
        const int cnt = (int)__popcll(__ballot(in));  // sorted => the in-lanes form a prefix
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
        }
        s_tab[b % 3ull][lane] = make_float4(sdf, uw, __uint_as_float(d.color), 0.0f);
        if (lane == 0) s_cnt[b % 3ull] = cnt;
        h_in = in;
        h_kind = (d.info >> 8) & 3u;
        h_lab = d.info & 0xffu;
        h_rp = (uint32_t)key_cur & F.point_mask;
        h_a = (h_kind == 1u) ? d.d_match : 0.0f;
        h_b = (h_kind == 1u) ? d.d_non : 0.0f;
        key_q[0] = key_q[1];
        d_q[0] = d_q[1];
        key_q[1] = key_n[0];
        d_q[1] = d_new;
        key_n[0] = key_n[1];
        key_n[1] = key_new;
      };
      auto second_half = [&]() {
        const unsigned long long b = (unsigned long long)f + 2ull * own;
        float* inc = &s_inc[b & 1ull][lane >> 2][0][lane & 3];
        if (h_in) {
          if (h_kind == 2u) {
            const float* dl = deltas + (size_t)h_rp * kNumLabels;
#pragma unroll
            for (int l = 0; l < kNumLabels; ++l) inc[4 * l] = dl[l];
          } else {
            float a = h_a, b = h_b;   // values in registers: a select between two captured variables is otherwise folded into
            KS_VALUE_BARRIER(a);      // ONE load from a selected ADDRESS, which pins both to scratch memory (ks_types.h)
            KS_VALUE_BARRIER(b);
#pragma unroll
            for (int l = 0; l < kNumLabels; ++l) inc[4 * l] = ((uint32_t)l == h_lab) ? a : b;
          }
        }
        ++own;
      };
      if (f == 0) first_half();  // batch 0, before step 0
      for (unsigned long long s = 0;; ++s) {
        if ((int)(s & 1ull) == f) second_half();  // batch s
        else first_half();                        // batch s + 1
        const int cnt_prev = s >= 1ull ? s_cnt[(s - 1ull) % 3ull] : 64;
        __syncthreads();  // step s
        if (cnt_prev < 64) break;
      }
      __syncthreads();  // (the label for the record head)
      __syncthreads();  // LDS free for the next run
      continue;
    }
    if (role == 1) {
      // ---- wave 1: the semantic log-likelihood sums, lane l owns class l ----
      float pri = (lane < kNumLabels) ? __uint_as_float(rec[4 + lane]) : 0.0f;
      __syncthreads();  // step 0
      for (unsigned long long s = 1;; ++s) {
        const unsigned long long b = s - 1ull;
        const int cnt = s_cnt[b % 3ull];
        if (cnt == 64) {
          // full batch: all increments are requested from LDS before the first dependent add
          float4 x[16];
#pragma unroll
          for (int g = 0; g < 16; ++g) x[g] = *(const float4*)&s_inc[b & 1ull][g][cls][0];
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            pri += x[g].x;
            pri += x[g].y;
            pri += x[g].z;
            pri += x[g].w;
          }
        } else {
#pragma unroll 4
          for (int k = 0; k < cnt; ++k) pri += s_inc[b & 1ull][k >> 2][cls][k & 3];
        }
        __syncthreads();  // step s
        if (cnt < 64) break;
      }
      // argmax over lanes 0..20, first strict maximum
      int best = 0;
      float m = bcast_f(pri, 0);
#pragma unroll
      for (int l = 1; l < kNumLabels; ++l) {
        const float x = bcast_f(pri, l);
        if (x > m) { m = x; best = l; }
      }
      if (lane < kNumLabels) rec[4 + lane] = __float_as_uint(pri);
      if (lane == 0) {
        s_best = (uint32_t)best;
        s_best_val = m;
      }
      __syncthreads();  // label for the record head
      __syncthreads();  // LDS free for the next run
      continue;
    }
    // ---- wave 0: the weight and distance recurrences ----
    float dist = __uint_as_float(rec[0]), weight = __uint_as_float(rec[1]);
    uint32_t color = rec[2];
    __syncthreads();  // step 0
    for (unsigned long long s = 1;; ++s) {
      const unsigned long long b = s - 1ull;
      const int cnt = s_cnt[b % 3ull];
      const float4 t = s_tab[b % 3ull][lane];
      const bool in = lane < cnt;
      const float sdf = t.x, uw = t.y;
      const uint32_t d_color = __float_as_uint(t.z);
      if (cnt > 0) {
        // ---- pass 1: the weight recurrence (independent of the distance) ----
        float my_w = 0.0f, my_nw = 1.0f;
        if (weight == Pm.max_weight && __ballot(in && !(uw >= 0.0f)) == 0ull) {
          my_w = weight;   // (k_apply_long: weight clamped at max_weight, increments non-negative: 64 independent additions)
          my_nw = weight + uw;
        } else {
          float w_run = weight;
          for (int k = 0; k < cnt; ++k) {
            const float nw = w_run + bcast_f(uw, k);
            if (lane == k) { my_w = w_run; my_nw = nw; }
            if (!(nw < kEps)) w_run = std_min(Pm.max_weight, nw);
          }
          weight = w_run;
        }
        const bool my_skip = my_nw < kEps;
        const float my_r = 1.0f / my_nw;
        const float my_p = sdf * uw;
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
                color = blend_two_colors(color, w_k, bcast_u(d_color, k), bcast_f(uw, k));
            }
            dist = (q > 0.0f) ? std_min(Pm.trunc, q) : std_max(-Pm.trunc, q);
          }
        }
      }
      __syncthreads();  // step s
      if (cnt < 64) break;
    }
    __syncthreads();  // the consumer has the label
    const uint32_t best = s_best;
    const float m = s_best_val;
    if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = label_lut[best];
    else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY)
      color = rainbow_color_map((double)(float)exp((double)m));
    if (lane == 0) {
      *(uint4*)rec = make_uint4(__float_as_uint(dist), __float_as_uint(weight), color, best);
      rec[25] = 1u;  // updated since the last voxel-level host sync
    }
    __syncthreads();  // LDS free for the next run
  }
}

}  // namespace ksk
