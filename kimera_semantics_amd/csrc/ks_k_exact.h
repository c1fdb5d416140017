// ks_k_exact.h — `fast` with the early-out enabled, in the reference's SERIAL order (opt-in:
// ks_config.early_out_phase_growth = KS_EARLY_OUT_EXACT).
//
// The reference's loop [K:src/semantic_tsdf_integrator_fast.cpp:110-122] is serial by construction: ray s
// stops after max_consecutive_ray_collisions + 1 consecutive voxels whose slot of
// voxel_observed_approx_set_ already holds their hash — and what a slot holds is the hash the LAST visit
// before (s, step) left there.  Every visited voxel (hit or not) leaves its hash in its slot (replaceHash
// stores on a miss, and on a hit the value is already there), so the set's content at any time is a function
// of the VISITED LENGTH L[s'] of the rays before it.  That makes the serial result the unique fixed point of
//     L[s] = stop( hits of ray s against { marks (s', k'), k' < L[s'] } that precede (s, k) )
// (uniqueness by induction over s: ray s depends only on rays s' < s and on its own earlier steps).
// The GPU iterates that map from the ordered-phase schedule's result (ks_k_march.h) as the seed:
//   k_eo_scan + k_eo_emit : every ray writes its marks  [slot | position | step] -> hash, in (position,
//                           step) order (offsets from a scan of the visited lengths)
//   radix sort by slot    : stable, so every slot's marks stay in time order          (ks_radix_sort.h)
//   k_eo_index            : slot -> its range of marks
//   k_eo_eval             : every ray walks again; a voxel's slot content is the hash of the last mark that
//                           precedes (position, step) — binary search in the slot's range — or, if none, what
//                           earlier frames left in the reference's table (kept verbatim in `plain`, including
//                           the zero-initialised slots that "contain" hash 0 and the SIZE_MAX poison)
// until no ray's length changes (640x480 / 5 cm: ~10 iterations from the seed; 2 cm voxels: ~25).  Errors
// die out geometrically: the effective dependency chains between rays are short.  The iteration count is
// data dependent, so the host reads one counter back per iteration: this mode is not pipelined.
#pragma once
#include "ks_k_march.h"

namespace ksk {

struct EoState {
  unsigned long long n_marks;       // marks emitted this iteration (= sum of the visited lengths)
  unsigned long long n_marks_next;  // sum of the visited lengths k_eo_eval leaves
  uint32_t changed;                 // rays whose length changed in k_eo_eval
  uint32_t pad[3];
};

struct EoBuf {
  const uint64_t* keys;  // sorted marks: [63:44] slot | [43:22] integration position | [21:0] step
  const uint32_t* vals;  // ... their voxel hashes
  uint2* range;          // per slot: [x, y) in keys / vals (x == y: no mark this frame)
  uint64_t* plain;       // the reference's table content as earlier frames left it (ApproxHashSet::pseudo_set_)
};
constexpr uint64_t kEoLow44 = (1ull << 44) - 1ull;

__device__ __forceinline__ uint32_t eo_visited(uint32_t cv) { return (cv & ~kCntBroke) + ((cv & kCntBroke) ? 1u : 0u); }

// sum of the visited lengths (the seed's mark count) -> state->n_marks_next
__global__ void __launch_bounds__(256) k_eo_total(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ cnt,
                                                  EoState* __restrict__ st) {
  const uint32_t n = Fp->n;
  unsigned long long acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += eo_visited(cnt[i]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0 && acc) atomicAdd(&st->n_marks_next, acc);
}

// exclusive scan of the visited lengths over integration positions, blocks of kScanBlock (as k_scan_local)
__global__ void __launch_bounds__(1024) k_eo_scan(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ cnt,
                                                  uint32_t* __restrict__ lp, unsigned long long* __restrict__ bt,
                                                  EoState* __restrict__ st) {
  __shared__ uint32_t s_wave[16];
  const uint32_t n = Fp->n;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // this iteration's counters
    st->changed = 0u;
    st->n_marks_next = 0ull;
  }
  if (blockIdx.x * kScanBlock >= n) return;
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const uint32_t i0 = blockIdx.x * kScanBlock + threadIdx.x * 4u;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? eo_visited(cnt[i0 + k]) : 0u;
  const uint32_t mine = v[0] + v[1] + v[2] + v[3];
  uint32_t x = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += y;
  }
  if (lane == 63) s_wave[wave] = x;
  __syncthreads();
  uint32_t wbase = 0;
  for (uint32_t w = 0; w < wave; ++w) wbase += s_wave[w];
  uint32_t run = wbase + x - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) lp[i0 + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 1023) bt[blockIdx.x] = (unsigned long long)(wbase + x);
}

// prefix of the scan's block totals into LDS (every workgroup redundantly); returns the grand total
__device__ __forceinline__ unsigned long long eo_fold_totals(const unsigned long long* __restrict__ bt, uint32_t nb,
                                                             unsigned long long* s_bt) {
  __shared__ unsigned long long s_carry;
  __shared__ unsigned long long s_w[4];
  if (threadIdx.x == 0) s_carry = 0ull;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
    const uint32_t b = b0 + threadIdx.x;
    const unsigned long long v = b < nb ? bt[b] : 0ull;
    unsigned long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long y = __shfl_up(x, o);
      if (lane_id() >= (uint32_t)o) x += y;
    }
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned long long wb = s_carry;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wb += s_w[w];
    if (b < nb) s_bt[b] = wb + x - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = wb + x;
    __syncthreads();
  }
  return s_carry;
}

// every live ray writes the marks of its visited voxels at its offset of the scan (work split as k_mark)
template <int RPW>
__global__ void __launch_bounds__(256) k_eo_emit(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ ray_list,
                                                 const RayDesc* __restrict__ rays, const uint32_t* __restrict__ cnt,
                                                 const uint32_t* __restrict__ lp, const unsigned long long* __restrict__ bt,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                 unsigned long long cap, const Counters* C, EoState* __restrict__ st) {
  extern __shared__ unsigned long long s_bt[];
  __shared__ float s_e[4][3 * kES];
  const FrameParams F = *Fp;  // a COPY: through the pointer every loop iteration would re-load the fields it uses (they may alias the stores)
  if (blockIdx.x != 0 && blockIdx.x * 4u * (uint32_t)RPW >= C->n_rays) return;
  const unsigned long long total = eo_fold_totals(bt, (F.n + kScanBlock - 1u) / kScanBlock, s_bt);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->n_marks = total;
  if (total > cap) return;  // (the host sized the buffers from the count it read back: cannot happen)
  const uint32_t lane = lane_id();
  const uint32_t r = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (uint32_t)RPW + lane;
  uint32_t pos = 0, visited = 0;
  unsigned long long base = 0;
  Dda dda{};
  if (lane < (uint32_t)RPW && r < C->n_rays) {
    pos = ray_list[r];
    visited = eo_visited(cnt[pos]);
    base = s_bt[pos / kScanBlock] + lp[pos];
    const RayDesc d = rays[ray_index(F, pos)];
    dda.setup(F.T.t, {d.px, d.py, d.pz}, ((d.info >> 10) & 1u) != 0, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc, false);
  }
  auto put = [&](unsigned long long at, uint32_t p, uint32_t step, int vx, int vy, int vz) {
    const uint32_t h = index_hash(vx, vy, vz);
    const uint64_t slot = ((uint64_t)h + F.observed_offset) & kSetMask;
    keys[at] = (slot << 44) | ((uint64_t)p << 22) | (uint64_t)step;
    vals[at] = h;
  };
  const unsigned long long long_mask = __ballot(visited > kLaneWalk);
  const bool by_wave = tails_by_wavefront(long_mask, visited);
  const uint32_t own = (by_wave && visited > kLaneWalk) ? kLaneWalk : visited;
  for (uint32_t s = 0; __ballot(s < own) != 0ull; ++s) {
    if (s < own) put(base + s, pos, s, dda.cx, dda.cy, dda.cz);
    dda.advance(s < own);
  }
  float* escr = s_e[threadIdx.x >> 6];
  for (unsigned long long todo = by_wave ? long_mask : 0ull; todo != 0ull; todo &= todo - 1ull) {
    const int j = __ffsll((long long)todo) - 1;
    Dda ust = dda_bcast(dda, j);  // state at step kLaneWalk
    const uint32_t v_j = __shfl(visited, j), pos_j = __shfl(pos, j);
    const unsigned long long base_j = __shfl(base, j);
    if (dda_parallel_ok(ust)) {
      for (uint32_t s0 = kLaneWalk; s0 < v_j; s0 += 64) {
        dda_round64(ust, escr, lane, [&](uint32_t rr, int vx, int vy, int vz) {
          if (s0 + rr < v_j) put(base_j + s0 + rr, pos_j, s0 + rr, vx, vy, vz);
        });
      }
    } else if ((int)lane == j) {
      for (uint32_t s = kLaneWalk; s < visited; ++s) {
        put(base + s, pos, s, dda.cx, dda.cy, dda.cz);
        dda.advance();
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_eo_index(unsigned long long n_marks, const uint64_t* __restrict__ keys,
                                                  uint2* __restrict__ range) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (i >= n_marks) return;
  const uint32_t slot = (uint32_t)(keys[i] >> 44);
  if (i == 0 || (uint32_t)(keys[i - 1] >> 44) != slot) range[slot].x = (uint32_t)i;
  if (i + 1 == n_marks || (uint32_t)(keys[i + 1] >> 44) != slot) range[slot].y = (uint32_t)(i + 1);
}

// does the slot of voxel hash h hold h at time t = position << 22 | step ?
__device__ __forceinline__ bool eo_hit(const EoBuf& E, uint32_t slot, uint32_t h, uint64_t t) {
  const uint2 r = E.range[slot];
  uint32_t lo = r.x, hi = r.y;  // first mark of the slot at or after t
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((E.keys[mid] & kEoLow44) < t) lo = mid + 1;
    else hi = mid;
  }
  if (lo > r.x) return E.vals[lo - 1] == h;
  return E.plain[slot] == (uint64_t)h;  // nothing this frame yet: what earlier frames (or the constructor) left
}

// every live ray decides again how far it gets, against the marks of the previous iteration
template <int RPW>
__global__ void __launch_bounds__(256) k_eo_eval(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ ray_list,
                                                 const RayDesc* __restrict__ rays, uint32_t* __restrict__ cnt, EoBuf E,
                                                 const Counters* C, EoState* __restrict__ st) {
  __shared__ float s_e[4][3 * kES];
  __shared__ unsigned long long s_keys[4][64];
  const FrameParams F = *Fp;  // a COPY: through the pointer every loop iteration would re-load the fields it uses (they may alias the stores)
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const uint32_t r0 = (blockIdx.x * 4u + wave) * (uint32_t)RPW;
  if (r0 >= C->n_rays) return;
  const uint32_t r = r0 + lane;
  const int lim = F.max_collisions;
  const bool mine = lane < (uint32_t)RPW && r < C->n_rays;
  uint32_t pos = 0, old = 0, full = 0;
  Dda dda{};
  if (mine) {
    pos = ray_list[r];
    old = cnt[pos];
    const RayDesc d = rays[ray_index(F, pos)];
    dda.setup(F.T.t, {d.px, d.py, d.pz}, ((d.info >> 10) & 1u) != 0, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc, false);
    full = (uint32_t)dda.steps + 1u;
  }
  // the owner lane tests its first kLaneWalk voxels one after the other
  int c = 0;
  int stop = -1;  // step the ray breaks on (visited, not updated)
  const uint32_t own = full < kLaneWalk ? full : kLaneWalk;
  for (uint32_t s = 0; __ballot(s < own && stop < 0) != 0ull; ++s) {
    const bool on = s < own && stop < 0;
    if (on) {
      const uint32_t h = index_hash(dda.cx, dda.cy, dda.cz);
      const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
      c = eo_hit(E, slot, h, ((uint64_t)pos << 22) | s) ? c + 1 : 0;
      if (c > lim) stop = (int)s;
    }
    dda.advance(on && stop < 0);  // (a ray that breaks keeps its state; it is not used again)
  }
  // rays not decided within kLaneWalk voxels: one at a time, 64 voxels per round by the whole wavefront
  unsigned long long todo = __ballot(mine && stop < 0 && full > kLaneWalk);
  float* escr = s_e[wave];
  unsigned long long* wkeys = s_keys[wave];
  for (; todo != 0ull; todo &= todo - 1ull) {
    const int j = __ffsll((long long)todo) - 1;
    Dda ust = dda_bcast(dda, j);  // state at step kLaneWalk
    const uint32_t full_j = __shfl(full, j), pos_j = __shfl(pos, j);
    int c_j = __shfl(c, j);
    int stop_j = -1;
    if (dda_parallel_ok(ust)) {
      for (uint32_t s0 = kLaneWalk; s0 < full_j; s0 += 64) {
        const uint32_t n_round = full_j - s0 < 64u ? full_j - s0 : 64u;
        dda_round64(ust, escr, lane, [&](uint32_t rr, int vx, int vy, int vz) {
          if (rr < n_round) {
            const uint32_t h = index_hash(vx, vy, vz);
            const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
            wkeys[rr] = ((unsigned long long)slot << 32) | h;
          }
        });
        __builtin_amdgcn_wave_barrier();
        const bool v64 = lane < n_round;
        bool hit = false;
        if (v64) {
          const unsigned long long k64 = wkeys[lane];
          hit = eo_hit(E, (uint32_t)(k64 >> 32), (uint32_t)k64, ((uint64_t)pos_j << 22) | (uint64_t)(s0 + lane));
        }
        __builtin_amdgcn_wave_barrier();
        const int st_r = early_out_stop(__ballot(v64 && hit), __ballot(v64), lim, c_j);
        if (st_r >= 0) {
          stop_j = (int)s0 + st_r;
          break;
        }
      }
      if ((int)lane == j) stop = stop_j;
    } else if ((int)lane == j) {  // axis-parallel ray (inf / NaN crossing times): its owner walks on
      for (uint32_t s = kLaneWalk; s < full && stop < 0; ++s) {
        const uint32_t h = index_hash(dda.cx, dda.cy, dda.cz);
        const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
        c = eo_hit(E, slot, h, ((uint64_t)pos << 22) | s) ? c + 1 : 0;
        if (c > lim) stop = (int)s;
        dda.advance();
      }
    }
  }
  uint32_t visited = 0;
  bool diff = false;
  if (mine) {
    const uint32_t now = stop >= 0 ? ((uint32_t)stop | kCntBroke) : full;
    visited = eo_visited(now);
    diff = now != old;
    if (diff) cnt[pos] = now;
  }
  const unsigned long long dm = __ballot(diff);
  unsigned long long vs = visited;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) vs += __shfl_xor(vs, o);
  if (lane == 0) {
    if (dm) atomicAdd(&st->changed, (uint32_t)__popcll(dm));
    if (vs) atomicAdd(&st->n_marks_next, vs);
  }
}

// the frame's marks enter the reference's table: per slot, the hash of the last mark in time order
__global__ void __launch_bounds__(256) k_eo_commit(unsigned long long n_marks, const uint64_t* __restrict__ keys,
                                                   const uint32_t* __restrict__ vals, uint64_t* __restrict__ plain) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (i >= n_marks) return;
  const uint32_t slot = (uint32_t)(keys[i] >> 44);
  if (i + 1 == n_marks || (uint32_t)(keys[i + 1] >> 44) != slot) plain[slot] = (uint64_t)vals[i];
}

}  // namespace ksk
