// ks_k_exact.h — `fast` with the early-out enabled, in the reference's SERIAL order: the library default
// (ks_config.early_out_phase_growth = 0 / KS_EARLY_OUT_EXACT).  Three loops live here, all reaching the same fixed point:
//   * the HOST-DRIVEN loop (k_eo_*: this comment and the first third of the file) — the fallback when a buffer of the device
//     loops overflows, and KS_EXACT_HOST_LOOP=1;
//   * the EVENT-DRIVEN loop on the device (k_eo2_hits .. k_eo2_commit): what runs at 5 cm / 5 m, pipelined;
//   * the SWEEPS along the chains of the integration order (k_eo2_full .. k_eo2_sweep_done): long rays (2 cm / 10 m).
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
// until no ray's length changes (640x480 / 5 cm: ~10 iterations from the seed; 1280x720 / 2 cm: ~90 — a change travels
// one ray per iteration).  The iteration count is data dependent, so the host reads one counter back per iteration: this
// loop is not pipelined.
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
// (cnt_b / ux / dirty: the event-driven path's per-position state, initialised here — next lengths = current lengths,
// steps covered by marks = visited length, nobody dirty)
// (view: long rays — per position {-, pad, ray length, -}: the marks emitted for a ray cover its VIEW, min(ray length,
// visited + pad) steps, so that an evaluation over the sorted marks can let the ray get further than it does now; see
// the dense iterations below.  nullptr: exactly the visited steps.)
__device__ __forceinline__ uint32_t eo_view_length(uint32_t visited, const uint4 ri) {
  const uint32_t v = visited + ri.y;
  return v < ri.z ? v : ri.z;
}
__device__ __forceinline__ void eo_scan_body(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ cnt,
                                             uint32_t* __restrict__ lp, unsigned long long* __restrict__ bt,
                                             EoState* __restrict__ st, uint32_t* __restrict__ cnt_b,
                                             uint32_t* __restrict__ ux, uint32_t* __restrict__ dirty,
                                             const uint4* __restrict__ view = nullptr) {
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
  for (int k = 0; k < 4; ++k) {
    const uint32_t cv = (i0 + k < n) ? cnt[i0 + k] : 0u;
    v[k] = eo_visited(cv);
    if (view && i0 + k < n) v[k] = eo_view_length(v[k], view[i0 + k]);
    if (cnt_b && i0 + k < n) {
      cnt_b[i0 + k] = cv;
      ux[i0 + k] = v[k];
      dirty[i0 + k] = 0u;
    }
  }
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
__global__ void __launch_bounds__(1024) k_eo_scan(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ cnt,
                                                  uint32_t* __restrict__ lp, unsigned long long* __restrict__ bt,
                                                  EoState* __restrict__ st) {
  eo_scan_body(Fp, cnt, lp, bt, st, nullptr, nullptr, nullptr);
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
__device__ __forceinline__ void eo_emit_body(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ ray_list,
                                             const RayDesc* __restrict__ rays, const uint32_t* __restrict__ cnt,
                                             const uint32_t* __restrict__ lp, const unsigned long long* __restrict__ bt,
                                             uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                             unsigned long long cap, const Counters* C, EoState* __restrict__ st,
                                             uint32_t* __restrict__ fail, unsigned long long* __restrict__ btp,
                                             uint32_t* __restrict__ hseq, uint4* __restrict__ rinfo,
                                             uint4* __restrict__ ckpt, bool view = false) {
  extern __shared__ unsigned long long s_bt[];
  __shared__ float s_e[4][3 * kES];
  const FrameParams F = *Fp;  // a COPY: through the pointer every loop iteration would re-load the fields it uses (they may alias the stores)
  if (blockIdx.x != 0 && blockIdx.x * 4u * (uint32_t)RPW >= C->n_rays) return;
  const unsigned long long total = eo_fold_totals(bt, (F.n + kScanBlock - 1u) / kScanBlock, s_bt);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->n_marks = total;
  if (btp && blockIdx.x == 0)   // the exclusive prefix of the block totals, for the kernels that map a mark back to its place in emission order
    for (uint32_t b = threadIdx.x; b < (F.n + kScanBlock - 1u) / kScanBlock; b += 256) btp[b] = s_bt[b];
  if (total > cap) {  // (host-driven loop: the host sized the buffers from the count it read back; event-driven: the frame falls back to it)
    if (fail && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(fail, 1u /* kEoFailMarks */);
    return;
  }
  const uint32_t lane = lane_id();
  const uint32_t r = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (uint32_t)RPW + lane;
  uint32_t pos = 0, visited = 0, pad = 0;   // visited: the steps marks are emitted for (the ray's view, see eo_scan_body)
  unsigned long long base = 0;
  Dda dda{};
  if (lane < (uint32_t)RPW && r < C->n_rays) {
    pos = ray_list[r];
    visited = eo_visited(cnt[pos]);
    if (view) {
      const uint4 ri = rinfo[pos];
      pad = ri.y;
      visited = eo_view_length(visited, ri);
    }
    base = s_bt[pos / kScanBlock] + lp[pos];
    const RayDesc d = rays[ray_index(F, pos)];
    dda.setup(F.T.t, {d.px, d.py, d.pz}, ((d.info >> 10) & 1u) != 0, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc, false);
  }
  auto put = [&](unsigned long long at, uint32_t p, uint32_t step, int vx, int vy, int vz) {
    const uint32_t h = index_hash(vx, vy, vz);
    const uint64_t slot = ((uint64_t)h + F.observed_offset) & kSetMask;
    keys[at] = (slot << 44) | ((uint64_t)p << 22) | (uint64_t)step;
    vals[at] = h;
    if (hseq) hseq[at] = h;   // (stays in emission order: a ray's hashes are contiguous)
  };
  // event-driven path: what a later look at the ray needs without casting it again — how many steps have a mark, its
  // length, and the caster's state at step cs <= visited (from where it is cast on if it outgrows its marks)
  auto checkpoint = [&](const Dda& d, uint32_t cs) {
    if (!rinfo) return;
    rinfo[pos] = make_uint4(visited, pad, (uint32_t)d.steps + 1u, cs);
    ckpt[3u * pos] = make_uint4((uint32_t)d.cx, (uint32_t)d.cy, (uint32_t)d.cz, (uint32_t)(d.sx + 1) | ((uint32_t)(d.sy + 1) << 2) | ((uint32_t)(d.sz + 1) << 4));
    ckpt[3u * pos + 1u] = make_uint4(__float_as_uint(d.tx), __float_as_uint(d.ty), __float_as_uint(d.tz), __float_as_uint(d.dx));
    ckpt[3u * pos + 2u] = make_uint4(__float_as_uint(d.dy), __float_as_uint(d.dz), 0u, 0u);
  };
  const unsigned long long long_mask = __ballot(visited > kLaneWalk);
  const bool by_wave = tails_by_wavefront(long_mask, visited);
  const uint32_t own = (by_wave && visited > kLaneWalk) ? kLaneWalk : visited;
  for (uint32_t s = 0; __ballot(s < own) != 0ull; ++s) {
    if (s < own) put(base + s, pos, s, dda.cx, dda.cy, dda.cz);
    dda.advance(s < own);
  }
  const bool is_ray = lane < (uint32_t)RPW && r < C->n_rays;
  if (is_ray) checkpoint(dda, own);   // (state after `own` steps; a ray whose tail is walked below by its owner lane overwrites it)
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
      checkpoint(dda, visited);
    }
  }
}
template <int RPW>
__global__ void __launch_bounds__(256) k_eo_emit(const FrameParams* __restrict__ Fp, const uint32_t* __restrict__ ray_list,
                                                 const RayDesc* __restrict__ rays, const uint32_t* __restrict__ cnt,
                                                 const uint32_t* __restrict__ lp, const unsigned long long* __restrict__ bt,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                 unsigned long long cap, const Counters* C, EoState* __restrict__ st) {
  eo_emit_body<RPW>(Fp, ray_list, rays, cnt, lp, bt, keys, vals, cap, C, st, nullptr, nullptr, nullptr, nullptr, nullptr);
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


// ==========================================================================================================
// EVENT-DRIVEN fix point (the default of the exact mode since round 4): no host in the loop, work proportional to what changes.
//
// The host-driven loop above re-emits, re-sorts and re-evaluates ALL marks per iteration and reads a counter back.
// Here the marks of the SEED are emitted and sorted ONCE (M; where[] maps a mark's place in emission order — a ray's steps
// are contiguous there — to its index in M), and a mark COUNTS while its step is below its ray's current visited length:
// one validity bit per mark, in two bitmaps (A: under the current lengths, B: under the next ones).
//   * the FIRST iteration is a full one and streams: every seed mark is valid, so whether a visit is a hit is a property of
//     the sorted list alone (k_eo2_hits), the stop rule runs over a ray's contiguous hit bits (k_eo2_stop0), and
//     k_eo2_bits builds the bitmaps from the new lengths and finds the readers whose predecessor in their slot went away;
//   * a round evaluates only DIRTY rays: a visit's slot content is the mark of the highest set bit below the visit's own
//     index in M — a bit scan, however many dead marks of rays that went too far in the seed sit in between (near the
//     sensor hundreds of seed marks share a slot) — or a mark of the slot's chain of X marks, or what earlier frames left;
//   * a ray that outgrows the steps the seed walked appends X marks (a lock-free list per slot; few: the seed's lengths
//     are almost right) and is cast on from a checkpoint of its caster; nothing is ever re-sorted;
//   * a ray whose length changed flips the bits of its marks between the old and the new length in B; each flipped mark
//     can change what exactly one reader sees — the next valid mark of its slot — so the propagation dirties the owner of
//     the next set bit of B in the slot (and the owners of the slot's later X marks), then flips A and makes the new length
//     current.
// Rounds are Jacobi steps (evaluate against A, propagate against B), every hand-over crosses a kernel boundary (or a
// workgroup barrier in the finisher): only bit flips, dirty flags, list counters and the X chains are touched by atomics.
// The fixed point is unique (ks_k_exact.h, top), so the order of the lists does not matter.
// A fixed number of bulk rounds is enqueued; k_eo2_finish (ONE workgroup) then iterates until nothing is dirty.
// (tools/fixpoint_study.py is the CPU study of this scheme: round counts, list sizes, X marks, scan lengths.)
// ==========================================================================================================
constexpr uint32_t kEoBulkMax = 40;          // bulk rounds a launch sequence can hold
constexpr uint32_t kEoFailMarks = 1u, kEoFailX = 2u, kEoFailRounds = 4u, kEoFailChain = 8u;
constexpr uint32_t kEoFinishRounds = 256;    // rounds of the finisher before it gives up (the host-driven loop takes over)
constexpr uint32_t kEoFinishRays = 2048;     // ... and the longest list ONE workgroup takes on (a frame whose lists are still long after the bulk
                                             // rounds is not converging like a sparse problem: 2 cm voxels / 10 m rays, where the approximate set is overwhelmed)

struct EoCtl {
  EoState st;                     // n_marks (k_eo_emit) ...
  uint32_t n_x;                   // X nodes handed out (node 0 = end of chain)
  uint32_t fail;                  // kEoFail*
  uint32_t rounds;                // rounds run (statistics)
  uint32_t n_consulted;           // rays whose result depends on what EARLIER FRAMES left in the table
  uint32_t fin_in[2], fin_chg;    // the finisher's list counters
  uint32_t pad;
  unsigned long long assumed0;    // what the slot of hash 0 held (as earlier frames left it) when the fix point began — see k_eo2_finish
  uint32_t n_in[kEoBulkMax + 2];  // dirty rays entering bulk round r
  uint32_t n_chg[kEoBulkMax + 2]; // rays whose length changed in bulk round r
  uint32_t dense_chg[32];         // long rays: rays whose length changed in sweep i (mod 32; statistics)
  uint32_t sw_prev, sw_cur;       // ... 0: a full sweep changed nothing — the fixed point / rays changed by the sweep that is running
  uint32_t sw_full, sw_pad;       // the next sweep looks at every ray (else: at the flagged ones)
};

struct EoView {
  const FrameParams* F;
  const uint32_t* ray_list;
  const RayDesc* rays;
  Counters* C;
  uint32_t* cnt_a;                // current lengths (the slot's cnt[]: updates | kCntBroke) — the result
  uint32_t* cnt_b;                // next lengths
  uint32_t* ux;                   // per position: steps that have a mark (in M or X)
  uint32_t* dirty;                // per position: queued for the next round
  const uint64_t* keys;           // M, sorted by slot (time order inside a slot): [63:44] slot | [43:22] position | [21:0] step
  const uint32_t* vals;           //    ... voxel hashes
  unsigned long long* bits_a;     // per mark of M: it counts under the current lengths
  unsigned long long* bits_b;     //                ... under the next lengths
  uint4* tab;                     // per set slot: {begin, end} in M, head of the X chain, unused
  unsigned long long* xnode;      // X marks, 2 words per node: position << 22 | step ;  hash | next << 32
  uint32_t cap_x;
  uint32_t* list[2];              // dirty rays (positions), ping-pong
  uint32_t* chg;                  // rays whose length changed this round
  uint32_t* consulted;
  uint64_t* plain;                // the reference's table as earlier frames left it
  uint32_t* committed;            // frames [0, *committed) have entered `plain` (this frame's number: FrameParams::eo_frame)
  uint32_t* lp;                   // scan of the seed's visited lengths (block-local) ...
  unsigned long long* bt;         // ... its block totals ...
  unsigned long long* btp;        // ... and their exclusive prefix: a ray's place in emission order = btp[pos / 4096] + lp[pos]
  uint64_t* keys0;                // the marks as emitted (the sort's input buffers)
  uint32_t* vals0;
  unsigned long long cap_marks;
  uint8_t* hitb;                  // first iteration: per mark in emission order, the visit is a hit
  const uint32_t* hseq;           // voxel hashes of the seed's marks in emission order
  uint32_t* hseq_w;               // (hseq, for the kernel that writes it)
  uint32_t* where;                // per seed mark in emission order: its index in M
  uint4* rinfo;                   // per position: {u0 = steps M has marks for (the view), pad of the next view, ray length, checkpoint step}
  uint4* ckpt;                    // per position: the caster's state at the checkpoint step (3 words of 16 bytes)
  uint32_t wide;                  // long rays (2 cm voxels): 8 rays per wavefront in the mark emission, views + dense iterations
  const uint8_t* live;            // per position: it holds a ray
  EoCtl* ctl;
};

// The frames of a batch (ks_k_march.h: BatchView) share every launch of the fix point up to the finisher: blockIdx.y = frame.
struct EoBatch {
  EoView v[kBatchMax];
};
__global__ void __launch_bounds__(1024) k_eo2_scan(EoBatch Bt) {
  const EoView& E = Bt.v[blockIdx.y];
  eo_scan_body(E.F, E.cnt_a, E.lp, E.bt, &E.ctl->st, E.cnt_b, E.ux, E.dirty, E.wide ? E.rinfo : nullptr);
}
template <int RPW>
__global__ void __launch_bounds__(256) k_eo2_emit(EoBatch Bt) {
  const EoView& E = Bt.v[blockIdx.y];
  eo_emit_body<RPW>(E.F, E.ray_list, E.rays, E.cnt_a, E.lp, E.bt, E.keys0, E.vals0, E.cap_marks, E.C, &E.ctl->st, &E.ctl->fail, E.btp, E.hseq_w,
                    E.rinfo, E.ckpt, E.wide != 0u);
}

__device__ __forceinline__ uint32_t eo2_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long eo2_ld64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the FIRST iteration is a full one, and streams -----------------------------------------------------------------
constexpr uint32_t kEoStepMax = 0x3fffffu;
__device__ __forceinline__ void eo2_mark_dirty(const EoView& E, uint32_t p, uint32_t step, uint32_t* out, uint32_t* n_out);

// per sorted mark: is the visit a hit (the mark before it in its slot holds the same hash; first of its slot: what earlier
// frames left there)?  -> hitb[its place in emission order]; and where[] of that place = the mark's index; and the slot's
// range in M (its first and its last mark write them)
__global__ void __launch_bounds__(256) k_eo2_hits(EoBatch Bt) {
  const EoView& E = Bt.v[blockIdx.y];
  const unsigned long long n = E.ctl->st.n_marks;
  if (E.ctl->fail) return;
  for (unsigned long long j = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * 256ull) {
    const uint64_t key = E.keys[j];
    const uint32_t slot = (uint32_t)(key >> 44), pos = (uint32_t)(key >> 22) & 0x3fffffu, step = (uint32_t)key & 0x3fffffu;
    const uint32_t h = E.vals[j];
    const bool first_of_slot = j == 0 || (uint32_t)(E.keys[j - 1] >> 44) != slot;
    // per slot: [begin, end) of its marks in M (the table is clear: k_eo2_commit leaves it so)
    if (first_of_slot) E.tab[slot].x = (uint32_t)j;
    if (j + 1 == n || (uint32_t)(E.keys[j + 1] >> 44) != slot) E.tab[slot].y = (uint32_t)(j + 1);
    bool hit;
    if (!first_of_slot) {
      hit = E.vals[j - 1] == h;
    } else {
      // (the only hash an entry of an EARLIER offset can equal is that of the zero-initialised slot; what that slot held
      // when the fix point began is the assumption every evaluation uses — k_eo2_finish checks it once the frames before
      // this one have entered their marks, and looks at these rays again if it was wrong)
      hit = h == 0u ? E.ctl->assumed0 == 0ull : E.plain[slot] == (uint64_t)h;
      if (h == 0u && !(atomicOr(&E.ux[pos], 0x80000000u) >> 31)) E.consulted[atomicAdd(&E.ctl->n_consulted, 1u)] = pos;
    }
    const unsigned long long at = E.btp[pos / kScanBlock] + E.lp[pos] + step;
    E.hitb[at] = hit ? 1 : 0;
    E.where[at] = (uint32_t)j;
  }
}

// per ray (a lane each): the reference's stop rule over its contiguous hit bits -> new length, in place (nothing reads
// lengths here).  A ray that used to stop and does not any more within the steps it has marks for goes on in round 1.
__global__ void __launch_bounds__(256) k_eo2_stop0(EoBatch Bt) {
  const EoView& E = Bt.v[blockIdx.y];
  if (E.ctl->fail) return;
  const uint32_t n = E.C->n_rays;
  const int lim = E.F->max_collisions;
  for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < n; r += gridDim.x * 256u) {
    const uint32_t pos = E.ray_list[r];
    const uint32_t cv = E.cnt_a[pos], v0 = eo_visited(cv);
    const uint8_t* hb = E.hitb + (E.btp[pos / kScanBlock] + E.lp[pos]);
    int c = 0, stop = -1;
    for (uint32_t k = 0; k < v0; ++k) {
      c = hb[k] ? c + 1 : 0;
      if (c > lim) {
        stop = (int)k;
        break;
      }
    }
    // no stop: every step that has a mark is visited and updated; if the ray used to stop there it goes on in round 1
    const uint32_t now = stop >= 0 ? ((uint32_t)stop | kCntBroke) : v0;
    if (now != cv) {
      E.cnt_a[pos] = now;
      E.cnt_b[pos] = now;
    }
    if (stop < 0 && (cv & kCntBroke)) eo2_mark_dirty(E, pos, v0, E.list[1], &E.ctl->n_in[1]);
  }
}

// The validity bitmaps under the new lengths (a wavefront's ballot is a word), and the readers whose input changed: the
// owner of every valid mark whose predecessor in its slot is not valid any more.
__global__ void __launch_bounds__(256) k_eo2_bits(EoBatch Bt, uint32_t detect) {
  const EoView& E = Bt.v[blockIdx.y];
  const unsigned long long n = E.ctl->st.n_marks;
  if (E.ctl->fail) return;
  const unsigned long long n_pad = (n + 63ull) & ~63ull;
  for (unsigned long long j = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; j < n_pad; j += (unsigned long long)gridDim.x * 256ull) {
    bool v = false;
    uint64_t key = 0;
    if (j < n) {
      key = E.keys[j];
      v = ((uint32_t)key & 0x3fffffu) < eo_visited(E.cnt_a[(uint32_t)(key >> 22) & 0x3fffffu]);
    }
    const unsigned long long word = __ballot(v);
    if (lane_id() == 0) {
      E.bits_a[j >> 6] = word;
      E.bits_b[j >> 6] = word;
    }
    if (detect && v && j > 0) {
      bool vprev;
      uint64_t kprev;
      if (lane_id() != 0) {
        vprev = (word >> (lane_id() - 1u)) & 1ull;
        kprev = E.keys[j - 1];
      } else {
        kprev = E.keys[j - 1];
        vprev = ((uint32_t)kprev & 0x3fffffu) < eo_visited(E.cnt_a[(uint32_t)(kprev >> 22) & 0x3fffffu]);
      }
      if (!vprev && (uint32_t)(kprev >> 44) == (uint32_t)(key >> 44)) eo2_mark_dirty(E, (uint32_t)(key >> 22) & 0x3fffffu, (uint32_t)key & 0x3fffffu, E.list[1], &E.ctl->n_in[1]);
    }
  }
}

// index of the highest set bit below index j of a bitmap (-1: none) / of the lowest set bit at or above j (n_words * 64: none)
__device__ __forceinline__ long long eo2_prev_set(const unsigned long long* __restrict__ bits, uint32_t j) {
  if (j == 0u) return -1;
  uint32_t w = (j - 1u) >> 6;
  unsigned long long x = bits[w] & (~0ull >> (63u - ((j - 1u) & 63u)));
  for (;;) {
    if (x != 0ull) return (long long)w * 64 + (63 - __clzll((long long)x));
    if (w == 0u) return -1;
    --w;
    x = bits[w];
  }
}
// (the same, reading past the L1: for kernels that read bits other wavefronts — or they themselves — flip while they run.
// Another XCD's flips of a line this XCD's L2 holds may not be seen: the per-XCD L2s are not coherent with each other.  Reading
// through returning atomics instead — performed where the flips are — was measured in the filtered sweeps and changed nothing:
// the rays a confirming sweep still finds are not victims of stale lines.)
__device__ __forceinline__ long long eo2_prev_set_live(unsigned long long* bits, uint32_t j) {
  if (j == 0u) return -1;
  uint32_t w = (j - 1u) >> 6;
  unsigned long long x = eo2_ld64(&bits[w]) & (~0ull >> (63u - ((j - 1u) & 63u)));
  for (;;) {
    if (x != 0ull) return (long long)w * 64 + (63 - __clzll((long long)x));
    if (w == 0u) return -1;
    --w;
    x = eo2_ld64(&bits[w]);
  }
}
__device__ __forceinline__ unsigned long long eo2_next_set(const unsigned long long* __restrict__ bits, unsigned long long j, unsigned long long n) {
  const unsigned long long n_words = (n + 63ull) >> 6;
  unsigned long long w = j >> 6;
  if (w >= n_words) return n;
  unsigned long long x = bits[w] & (~0ull << (j & 63ull));
  for (;;) {
    if (x != 0ull) {
      const unsigned long long at = w * 64ull + (unsigned long long)(__ffsll((long long)x) - 1);
      return at < n ? at : n;
    }
    if (++w >= n_words) return n;
    x = bits[w];
  }
}

// What slot `slot` holds at time t = position << 22 | step, for the ray at `pos`, given the index j of the first mark of M at
// or after t (the ray's own mark of that step, if it has one in M).  The mark right before j counts if it is the ray's
// own (an earlier step of this walk: marks between two marks of one ray are that ray's); else the mark of the highest set
// bit of A below j, if it is of this slot; plus the slot's X marks (valid by their ray's current length, or the ray's own).
__device__ __forceinline__ bool eo2_content_at(const EoView& E, uint32_t j, uint32_t slot, uint64_t t, uint32_t pos, uint32_t& hash) {
  bool found = false;
  uint64_t best = 0;
  if (j > 0u) {
    const uint64_t kp = E.keys[j - 1u];
    long long i = -1;
    if ((uint32_t)(kp >> 44) == slot && ((uint32_t)(kp >> 22) & 0x3fffffu) == pos) i = (long long)j - 1;
    else i = eo2_prev_set(E.bits_a, j);
    if (i >= 0) {
      const uint64_t key = i == (long long)j - 1 ? kp : E.keys[i];
      if ((uint32_t)(key >> 44) == slot) {
        found = true;
        best = key & kEoLow44;
        hash = E.vals[i];
      }
    }
  }
  uint32_t xi = eo2_ld(&E.tab[slot].z);
  while (xi != 0u) {
    const unsigned long long k = eo2_ld64(&E.xnode[2u * xi]), hn = eo2_ld64(&E.xnode[2u * xi + 1u]);
    const uint32_t p = (uint32_t)(k >> 22), st = (uint32_t)k & 0x3fffffu;
    if (k < t && (p == pos || st < eo_visited(E.cnt_a[p])) && (!found || k > best)) {
      found = true;
      best = k;
      hash = (uint32_t)hn;
    }
    xi = (uint32_t)(hn >> 32);
  }
  return found;
}
// index of the first mark of M at or after time t in slot `slot` (binary search in the slot's range; for steps without a
// mark of their own in M)
__device__ __forceinline__ uint32_t eo2_lower_bound(const EoView& E, uint32_t slot, uint64_t t) {
  const uint4 e = E.tab[slot];
  uint32_t lo = e.x, hi = e.y;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((E.keys[mid] & kEoLow44) < t) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// ---- long rays (2 cm voxels, 10 m: stage B is "wide") ----------------------------------------------------------------------
// At that geometry the approximate set is overwhelmed (tens of marks per slot and frame), the seed is wrong on most rays,
// and a third of the final marks lie beyond the steps the seed walked: as X marks they would make every slot's chain tens
// of nodes long.  So the marks of M cover a ray's VIEW — here the whole ray (k_eo2_full; k_eo2_scan / k_eo2_emit) — emitted
// and sorted ONCE per frame; a mark counts while its step is below its ray's current length (bitmap A), and no step is
// ever without a mark.

// per position: the ray's length in steps and its view's pad = the whole ray (dead positions: an empty view)
__global__ void __launch_bounds__(256) k_eo2_full(EoBatch Bt) {
  const EoView& E = Bt.v[blockIdx.y];
  const FrameParams F = *E.F;
  for (uint32_t pos = blockIdx.x * 256u + threadIdx.x; pos < F.n; pos += gridDim.x * 256u) {
    uint4 ri = make_uint4(0u, 0u, 0u, 0u);
    if (E.live[pos]) {
      const RayDesc d = E.rays[ray_index(F, pos)];
      Dda dda{};
      dda.setup(F.T.t, {d.px, d.py, d.pz}, ((d.info >> 10) & 1u) != 0, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc, false);
      ri = make_uint4(0u, (uint32_t)dda.steps + 1u, (uint32_t)dda.steps + 1u, 0u);
    }
    E.rinfo[pos] = ri;
  }
}

// per sorted mark: where[] of its place in emission order = its index in M; the slots' ranges in M
__global__ void __launch_bounds__(256) k_eo2_where(EoBatch Bt) {
  const EoView& E = Bt.v[blockIdx.y];
  const unsigned long long n = E.ctl->st.n_marks;
  if (E.ctl->fail) return;
  for (unsigned long long j = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * 256ull) {
    const uint64_t key = E.keys[j];
    const uint32_t slot = (uint32_t)(key >> 44), pos = (uint32_t)(key >> 22) & 0x3fffffu, step = (uint32_t)key & 0x3fffffu;
    if (j == 0 || (uint32_t)(E.keys[j - 1] >> 44) != slot) E.tab[slot].x = (uint32_t)j;
    if (j + 1 == n || (uint32_t)(E.keys[j + 1] >> 44) != slot) E.tab[slot].y = (uint32_t)(j + 1);
    E.where[E.btp[pos / kScanBlock] + E.lp[pos] + step] = (uint32_t)j;
  }
}

// ---- long rays: SWEEPS ------------------------------------------------------------------------------------------------------
// Jacobi steps (k_eo2_hits_b + k_eo2_stopv, or the rounds) move a change ONE ray further per step, and at 2 cm / 10 m a
// change travels far: measured on full-size 1280x720 frames, 1e4 rays still change after 32 steps and the host-driven loop
// needs ~90.  The reference's loop is a single pass because every ray sees what the rays before it left.  A sweep is the
// parallel version of that: a wavefront per ray, rays taken in INTEGRATION ORDER (the hardware starts workgroups in order,
// and ~8000 wavefronts are resident of 6.5e5 rays), every visit looked up in the live bitmap A (through where[]: the
// marks of M cover the whole ray), and a changed length applied AT ONCE — so a ray that starts later sees it.  What a
// wavefront reads may be stale or half applied (plain loads, other wavefronts' atomics): that only costs another sweep,
// because a sweep in which NO length changed has read one unchanging state, the fixed point — and only that ends the
// iteration (k_eo2_sweep_done turns anything else into a failure: the host-driven loop takes the frame).
__device__ __forceinline__ unsigned long long eo2_next_set_live(unsigned long long* bits, unsigned long long j, unsigned long long n) {
  const unsigned long long n_words = (n + 63ull) >> 6;
  unsigned long long w = j >> 6;
  if (w >= n_words) return n;
  unsigned long long x = eo2_ld64(&bits[w]) & (~0ull << (j & 63ull));
  for (;;) {
    if (x != 0ull) {
      const unsigned long long at = w * 64ull + (unsigned long long)(__ffsll((long long)x) - 1);
      return at < n ? at : n;
    }
    if (++w >= n_words) return n;
    x = eo2_ld64(&bits[w]);
  }
}
// One ray of a sweep, by a whole wavefront; returns (wave-uniform) whether its length changed.
// cv: the ray's current length word (in: as the caller knows it; out: as this look leaves it)
__device__ __forceinline__ bool eo2_sweep_ray_once(const EoView& E, EoCtl* ctl, const FrameParams& F, uint32_t pos, uint32_t lane, uint32_t* cv_io,
                                                   bool* extended) {
  const int lim = F.max_collisions;
  const uint64_t offset = F.observed_offset;
  const uint32_t cv = *cv_io, vo = eo_visited(cv);
  const uint4 ri = E.rinfo[pos];
  const uint32_t view = ri.x, full = ri.z;
  const unsigned long long base = E.btp[pos / kScanBlock] + E.lp[pos];
  int c = 0, stop = -1;
  bool consulted = false;
  for (uint32_t k0 = 0; k0 < view && stop < 0; k0 += 64u) {
    const bool act = k0 + lane < view;
    bool hit = false;
    if (act) {
      const uint32_t j = E.where[base + k0 + lane], h = E.hseq[base + k0 + lane];
      const uint32_t slot = (uint32_t)(((uint64_t)h + offset) & kSetMask);
      long long i = -1;
      if (j > 0u) {
        const uint64_t kp = E.keys[j - 1u];
        if ((uint32_t)(kp >> 44) == slot) {
          // (the bitmap is read past the L1: a wavefront that walks a chain reads the bits it has just written itself)
          if (((uint32_t)(kp >> 22) & 0x3fffffu) == pos || ((eo2_ld64(&E.bits_a[(j - 1u) >> 6]) >> ((j - 1u) & 63u)) & 1ull)) i = (long long)j - 1;
          else {
            i = eo2_prev_set_live(E.bits_a, j);
            if (i >= 0 && (uint32_t)(E.keys[i] >> 44) != slot) i = -1;
          }
        }
      }
      if (i >= 0) hit = E.vals[i] == h;
      else {
        hit = h == 0u ? ctl->assumed0 == 0ull : E.plain[slot] == (uint64_t)h;
        consulted |= h == 0u;
      }
    }
    const int st_r = early_out_stop(__ballot(act && hit), __ballot(act), lim, c);
    if (st_r >= 0) stop = (int)k0 + st_r;
  }
  // (whole-ray views: a ray that does not stop ends with its view)
  const uint32_t now = stop >= 0 ? ((uint32_t)stop | kCntBroke) : (view >= full ? full : view);
  if (__ballot(consulted) != 0ull && lane == 0 && !(atomicOr(&E.ux[pos], 0x80000000u) >> 31)) E.consulted[atomicAdd(&ctl->n_consulted, 1u)] = pos;
  if (now == cv) return false;
  const uint32_t vn = eo_visited(now);
  *extended = vn > vo;
  *cv_io = now;
  const uint32_t lo = vo < vn ? vo : vn, hi = vo < vn ? vn : vo;
  for (uint32_t k = lo + lane; k < hi; k += 64u) {
    const uint32_t j = E.where[base + k];
    if (vn > vo) {
      atomicOr(&E.bits_a[j >> 6], 1ull << (j & 63u));
      atomicOr(&E.bits_b[j >> 6], 1ull << (j & 63u));
    } else {
      atomicAnd(&E.bits_a[j >> 6], ~(1ull << (j & 63u)));
      atomicAnd(&E.bits_b[j >> 6], ~(1ull << (j & 63u)));
    }
  }
  if (lane == 0) {
    __hip_atomic_store(&E.cnt_a[pos], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    E.cnt_b[pos] = now;
  }
  // HINTS for the filtered sweeps (k_eo2_sweep): the ray that reads each flipped mark next — the owner of the next mark of
  // the slot that counts — is flagged, once the flips are in place.  Only hints: a reader that starts to count while this
  // runs can be missed, which is why a filtered sweep that changes nothing is followed by a full one.
  KS_WAIT_VMEM();
  const unsigned long long n_m = ctl->st.n_marks;
  for (uint32_t k = lo + lane; k < hi; k += 64u) {
    const uint32_t j = E.where[base + k];
    const unsigned long long m = eo2_next_set_live(E.bits_a, (unsigned long long)j + 1ull, n_m);
    if (m < n_m) {
      const uint64_t key = E.keys[m];
      const uint32_t p = (uint32_t)(key >> 22) & 0x3fffffu;
      if ((uint32_t)(key >> 44) == (uint32_t)(E.keys[j] >> 44) && p != pos) __hip_atomic_store(&E.dirty[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return true;
}
// A ray that gets FURTHER has looked at its new steps before their marks counted: a neighbour that flipped a mark in between
// looked for the next reader of that mark and could not find this ray yet.  So such a ray looks again once its own flips are
// in place — from then on every flip finds it.
__device__ __forceinline__ bool eo2_sweep_ray(const EoView& E, EoCtl* ctl, const FrameParams& F, uint32_t pos, uint32_t lane) {
  bool changed = false;
  // (the length word travels in a register from one look to the next: the store of lane 0 is neither ordered against the other
  // lanes' next load nor certain to have left this CU's L1)
  uint32_t cv = eo2_ld(&E.cnt_a[pos]);
  for (int pass = 0; pass < 3; ++pass) {
    bool extended = false;
    if (!eo2_sweep_ray_once(E, ctl, F, pos, lane, &cv, &extended)) break;
    changed = true;
    if (!extended) break;
    KS_WAIT_VMEM();
  }
  return changed;
}

// order 0: rays in integration order, a wavefront per ray (grid-stride).  order 1: a wavefront per (chain, segment of
// kEoSweepSegment generations): it takes ITS rays one after the other, generation by generation, waiting for its own bit
// flips before the next one — the reference's loop along a chain of neighbouring pixels, which is the direction changes
// travel in (ks_k_march.h: the chains of the "mixed" order); chains and segments run side by side.
// ctl->sw_prev = rays the previous sweep changed (k_eo2_sweep_next rotates the counters between two sweeps).
constexpr uint32_t kEoSweepSegment = 128;
// FULL and FILTERED sweeps: a full sweep looks at every ray (the first one, and the one that CONFIRMS the fixed point: nothing
// else ends the iteration); a filtered sweep only at the rays flagged since they were last looked at (E.dirty: hints, see
// eo2_sweep_ray) — after the first few sweeps a few thousand of 6.5e5.  ctl->sw_full says which kind runs (k_eo2_sweep_next).
__device__ __forceinline__ bool eo2_sweep_wants(const EoView& E, uint32_t pos, bool full, uint32_t lane) {
  // (the flag goes down before the ray is looked at: a change that lands meanwhile flags it again)
  uint32_t d = 0u;
  if (lane == 0) d = atomicExch(&E.dirty[pos], 0u);
  d = (uint32_t)__shfl((int)d, 0);
  return full || d != 0u;
}
__global__ void __launch_bounds__(256) k_eo2_sweep(EoBatch Bt, uint32_t order) {
  const EoView& E = Bt.v[blockIdx.y];
  EoCtl* ctl = E.ctl;
  if (ctl->fail || ctl->sw_prev == 0u) return;   // a full sweep changed nothing: the fixed point
  const bool full = ctl->sw_full != 0u;
  const FrameParams& F = *E.F;
  const uint32_t lane = lane_id(), w0 = blockIdx.x * 4u + (threadIdx.x >> 6), nw = gridDim.x * 4u;
  uint32_t changed = 0;
  if (order == 0u) {
    const uint32_t n = E.C->n_rays;
    for (uint32_t r = w0; r < n; r += nw) {
      const uint32_t pos = E.ray_list[r];
      if (eo2_sweep_wants(E, pos, full, lane)) changed += eo2_sweep_ray(E, ctl, F, pos, lane) ? 1u : 0u;
    }
  } else {
    const uint32_t n_chains = F.chains, n_gen = (F.n + n_chains - 1u) / n_chains;
    const uint32_t n_seg = (n_gen + kEoSweepSegment - 1u) / kEoSweepSegment;
    for (uint32_t w = w0; w < n_chains * n_seg; w += nw) {
      // (segment-major: the workgroups that start first hold the chains' first segments)
      const uint32_t chain = w % n_chains, seg = w / n_chains;
      const uint32_t g1 = (seg + 1u) * kEoSweepSegment < n_gen ? (seg + 1u) * kEoSweepSegment : n_gen;
      for (uint32_t g0 = seg * kEoSweepSegment; g0 < g1; g0 += 64u) {
        const uint32_t g = g0 + lane;
        const uint64_t p = (uint64_t)g * n_chains + chain;
        // The flags of the segment's next 64 rays, all lanes at once (one atomic per ray, one after the other, is what a
        // filtered sweep with a handful of flagged rays would otherwise consist of); a flagged ray's flag goes down BEFORE the
        // rays are looked at, so a change that lands while the wavefront is on its way flags the ray again for the next sweep.
        // A ray that CHANGES is what flags the rays behind it — most of all its own chain's next ray: that one is looked at
        // whatever its flag says, and the flags of the rest of the group are read again (measured: reading them once per group
        // lets a change travel one group per sweep along its chain — 45 sweeps per frame instead of 36).
        const bool mine = g < g1 && p < F.n && E.live[p < F.n ? p : 0u] != 0;
        auto take_flags = [&](uint32_t from) -> unsigned long long {   // the flags of the group's rays of generation >= from
          bool want = false;
          if (mine && g >= from && __hip_atomic_load(&E.dirty[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            (void)atomicExch(&E.dirty[p], 0u);
            want = true;
          }
          const unsigned long long m = __ballot(want);
          if (m != 0ull) KS_WAIT_VMEM();
          return m;
        };
        unsigned long long todo = full ? __ballot(mine) : 0ull;
        todo |= take_flags(g0);
        bool prev_changed = false;
        for (unsigned long long rest = __ballot(mine); rest != 0ull; rest &= rest - 1ull) {
          const uint32_t bit = (uint32_t)(__ffsll((long long)rest) - 1), gi = g0 + bit;
          if (!((todo >> bit) & 1ull) && !prev_changed) continue;
          prev_changed = eo2_sweep_ray(E, ctl, F, gi * n_chains + chain, lane);
          if (prev_changed) {
            ++changed;
            KS_WAIT_VMEM();   // the bits of this ray's marks are in place before the chain's next ray looks
            todo |= take_flags(gi + 1u);
          }
        }
      }
    }
  }
  if (lane == 0 && changed) atomicAdd(&ctl->sw_cur, changed);
}
// between two sweeps: what the last one changed becomes "previous"
__global__ void __launch_bounds__(64) k_eo2_sweep_next(EoBatch Bt) {
  EoCtl* ctl = Bt.v[blockIdx.x].ctl;
  if (threadIdx.x != 0 || ctl->fail || ctl->sw_prev == 0u) return;
  const uint32_t i = ctl->rounds, chg = ctl->sw_cur;
  ctl->dense_chg[i & 31u] = chg | (ctl->sw_full ? 0x80000000u : 0u);   // (statistics: KS_EXACT_TRACE; bit 31: a full sweep)
  ctl->rounds = i + 1u;
  // a full sweep that changed nothing ends the iteration; a filtered one that changed nothing asks for the full one
  if (ctl->sw_full) {
    ctl->sw_prev = chg;
    ctl->sw_full = 0u;
  } else {
    ctl->sw_prev = 1u;
    ctl->sw_full = chg == 0u ? 1u : 0u;
  }
  ctl->sw_cur = 0u;
}
// after the last sweep the host is willing to enqueue: a frame that is still changing has not reached the fixed point
__global__ void __launch_bounds__(64) k_eo2_sweep_done(EoBatch Bt) {
  EoCtl* ctl = Bt.v[blockIdx.x].ctl;
  if (threadIdx.x == 0 && !ctl->fail && ctl->sw_prev != 0u) atomicOr(&ctl->fail, kEoFailRounds);
}

// LDS of a wavefront of the round kernels
struct EoWaveLds {
  unsigned long long keys[64];  // slot << 32 | hash of the 64 steps being looked at
  float e[3 * kES];             // scratch of the parallel caster
};

// keys of the steps [first, first + 64) of a ray (as far as it goes: full = its step count) -> W.keys; `ust` = wave-uniform
// caster state at `first` (advanced)
__device__ __forceinline__ void eo2_cast64(const FrameParams& F, Dda& ust, bool par, uint32_t first, uint32_t full, EoWaveLds& W, uint32_t lane) {
  const uint32_t n_r = full - first < 64u ? full - first : 64u;
  if (par) {
    dda_round64(ust, W.e, lane, [&](uint32_t r, int vx, int vy, int vz) {
      if (r < n_r) {
        const uint32_t h = index_hash(vx, vy, vz);
        const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
        W.keys[r] = ((unsigned long long)slot << 32) | h;
      }
    });
  } else {
    // (NaN / inf crossing times: every lane walks the same serial caster — the state stays wave-uniform — lane 0 writes)
    for (uint32_t i = 0; i < n_r; ++i) {
      if (lane == 0) {
        const uint32_t h = index_hash(ust.cx, ust.cy, ust.cz);
        const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
        W.keys[i] = ((unsigned long long)slot << 32) | h;
      }
      ust.advance(first + i + 1u < full);
    }
    for (uint32_t i = n_r; i < 64u; ++i) ust.advance(false);
  }
  KS_WAVE_LDS_ORDER();
}

// A ray's steps in batches of up to 64 consecutive ones, WITHOUT casting it for the steps the seed has marks for: their
// hashes are contiguous in hseq (the slot follows from the hash).  Past them the ray is cast on from its checkpoint
// (rounds of 64 steps on the checkpoint's grid).  All members wave-uniform.
struct EoWalk {
  unsigned long long base;  // the ray's place in emission order
  uint32_t u0, um, full, cs, pos;
  uint32_t k0;              // first step of the next batch's grid
  bool cast;                // past the seed's steps
  bool par;
  Dda ust;
  __device__ __forceinline__ void begin(const EoView& E, uint32_t p) {
    pos = p;
    const uint4 ri = E.rinfo[p];
    u0 = ri.x; um = ri.y; full = ri.z; cs = ri.w;
    base = E.btp[p / kScanBlock] + E.lp[p];
    k0 = 0;
    cast = false;
    par = false;
  }
  // next batch of steps below `limit`: W.keys[k - g] = slot << 32 | hash for the steps k in [a, b) (b - a <= 64, g <= a the
  // batch's grid origin: lane l holds step g + l); false when there is none
  __device__ __forceinline__ bool next(const EoView& E, const FrameParams& F, EoWaveLds& W, uint32_t lane, uint32_t limit, uint32_t& g, uint32_t& a,
                                       uint32_t& b) {
    if (limit > full) limit = full;
    for (;;) {
      if (!cast) {
        const uint32_t end = u0 < limit ? u0 : limit;
        if (k0 < end) {
          g = a = k0;
          b = k0 + 64u < end ? k0 + 64u : end;
          if (a + lane < b) {
            const uint32_t h = E.hseq[base + a + lane];
            W.keys[lane] = ((unsigned long long)(uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask) << 32) | h;
          }
          KS_WAVE_LDS_ORDER();
          k0 += 64u;
          return true;
        }
        if (u0 >= limit) return false;
        // the ray goes on past the steps the seed walked: its caster's state at step cs <= u0
        const uint4 c0 = E.ckpt[3u * pos], c1 = E.ckpt[3u * pos + 1u], c2 = E.ckpt[3u * pos + 2u];
        ust.cx = (int)c0.x; ust.cy = (int)c0.y; ust.cz = (int)c0.z;
        ust.sx = (int)(c0.w & 3u) - 1; ust.sy = (int)((c0.w >> 2) & 3u) - 1; ust.sz = (int)((c0.w >> 4) & 3u) - 1;
        ust.tx = __uint_as_float(c1.x); ust.ty = __uint_as_float(c1.y); ust.tz = __uint_as_float(c1.z);
        ust.dx = __uint_as_float(c1.w); ust.dy = __uint_as_float(c2.x); ust.dz = __uint_as_float(c2.y);
        ust.steps = (int)full - 1;
        ust.in_range = true;
        par = dda_parallel_ok(ust);
        cast = true;
        k0 = cs;
      } else {
        if (k0 >= limit) return false;
        eo2_cast64(F, ust, par, k0, full, W, lane);
        g = k0;
        a = k0 > u0 ? k0 : u0;
        b = k0 + 64u < limit ? k0 + 64u : limit;
        k0 += 64u;
        if (a < b) return true;
      }
    }
  }
};

// ONE ray evaluated by a whole wavefront (all operands wave-uniform): how far it gets against the marks valid under A.
// Appends X marks for the visited steps it has no mark for yet; a changed length goes to B and the ray to the change list.
__device__ __forceinline__ void eo2_eval_ray(const EoView& E, const FrameParams& F, uint32_t pos, EoWaveLds& W, uint32_t* n_chg) {
  const uint32_t lane = lane_id();
  // the earliest step whose input changed since the ray was last evaluated (eo2_mark_dirty); the flag goes down BEFORE
  // anything is read, so a change that lands during this evaluation queues the ray again
  uint32_t dword = 0u;
  if (lane == 0) dword = atomicExch(&E.dirty[pos], 0u);
  dword = (uint32_t)__shfl((int)dword, 0);
  uint32_t r_min = dword ? kEoStepMax - (dword & kEoStepMax) : 0u;
  EoWalk wk;
  wk.begin(E, pos);
  const uint32_t full = wk.full, u0 = wk.u0;
  const int lim = F.max_collisions;
  // (wave-uniform addresses whose content changes inside the one-workgroup finisher: read past the scalar cache)
  const uint32_t old = eo2_ld(&E.cnt_a[pos]);
  {
    // (the owners of a slot's later X marks are queued whether or not the ray still gets that far: a step past the ray's
    // last visit is no visit — what the ray's last visit saw is then what may have changed)
    const uint32_t vo = eo_visited(old);
    if (r_min + 1u > vo) r_min = vo ? vo - 1u : 0u;
  }
  // Nothing below r_min changed, and the ray did not stop there (it owns a visit at r_min): its walk is taken up 64 steps
  // before r_min — the consecutive-hit counter at r_min is the length of a run that ended inside those 64 steps, since a
  // longer run would have stopped the ray (lim < 32) — but never inside the part that has to be cast (the caster's
  // checkpoint is at or below u0).
  if (lim >= 0 && lim < 32 && r_min > 64u) {
    const uint32_t from = r_min - 64u, cap = u0 > 64u ? u0 - 64u : 0u;
    wk.k0 = from < cap ? from : cap;
  }
  const uint32_t ux_word = eo2_ld(&E.ux[pos]);          // [31] the ray is in the list of rays that consulted earlier frames' marks
  const uint32_t ux0 = ux_word & 0x7fffffffu;
  int c = 0, stop = -1;
  bool consulted = false;
  uint32_t visited = full;
  uint32_t g, a, b;
  while (wk.next(E, F, W, lane, full, g, a, b)) {
    const uint32_t k = g + lane;
    const bool act = k >= a && k < b;
    bool hit = false;
    unsigned long long key = 0ull;
    if (act) {
      key = W.keys[lane];
      const uint32_t slot = (uint32_t)(key >> 32), h = (uint32_t)key;
      const uint64_t t = ((uint64_t)pos << 22) | k;
      bool own = false;
      if (k >= ux0) {
        // steps of this batch that have no mark yet: the ray's own latest earlier visit of the slot among them, if any
        const uint32_t first = (ux0 > a ? ux0 : a) - g;
        for (uint32_t l2 = lane; l2 > first && !own;) {
          --l2;
          const unsigned long long k2 = W.keys[l2];
          if ((uint32_t)(k2 >> 32) == slot) {
            own = true;
            hit = (uint32_t)k2 == h;
          }
        }
      }
      if (!own) {
        uint32_t content = 0;
        const uint32_t j = k < u0 ? E.where[wk.base + k] : eo2_lower_bound(E, slot, t);
        const bool found = eo2_content_at(E, j, slot, t, pos, content);
        if (found) hit = content == h;
        else {
          hit = h == 0u ? eo2_ld64(&E.ctl->assumed0) == 0ull : E.plain[slot] == (uint64_t)h;
          consulted |= h == 0u;   // (the only hash an entry of an EARLIER offset can equal: the zero-initialised slot)
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t sh = a - g;
    const int st_r = early_out_stop(__ballot(act && hit) >> sh, __ballot(act) >> sh, lim, c);
    const uint32_t n_vis = st_r >= 0 ? (uint32_t)st_r + 1u : b - a;   // steps of this batch the ray visits
    // X marks for the visited steps without a mark
    const bool need = act && k - a < n_vis && k >= ux0;
    const unsigned long long nm = __ballot(need);
    if (nm != 0ull) {
      uint32_t xb = 0;
      if (lane == 0) xb = atomicAdd(&E.ctl->n_x, (uint32_t)__popcll(nm));
      xb = (uint32_t)__shfl((int)xb, 0);
      if (xb + (uint32_t)__popcll(nm) > E.cap_x) {
        if (lane == 0) atomicOr(&E.ctl->fail, kEoFailX);
      } else if (need) {
        const uint32_t xi = xb + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
        const uint32_t slot = (uint32_t)(key >> 32);
        __hip_atomic_store(&E.xnode[2u * xi], ((unsigned long long)pos << 22) | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t head = eo2_ld(&E.tab[slot].z);
        for (;;) {
          __hip_atomic_store(&E.xnode[2u * xi + 1u], (unsigned long long)(uint32_t)key | ((unsigned long long)head << 32), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          // the node's two words (agent-scope stores: they do not stay in a cache other CUs cannot see) have been performed
          // before the head names the node.  (Not a fence: an agent-scope release would write back the whole L2 — per push.)
          KS_WAIT_VMEM();
          const uint32_t seen = atomicCAS(&E.tab[slot].z, head, xi);
          if (seen == head) break;
          head = seen;
        }
      }
    }
    if (st_r >= 0) {
      stop = (int)(a + (uint32_t)st_r);
      visited = (uint32_t)stop + 1u;
      break;
    }
  }
  const bool any_consulted = __ballot(consulted) != 0ull;
  if (lane == 0) {
    const bool list_it = any_consulted && !(ux_word >> 31);
    if (visited > ux0 || list_it) E.ux[pos] = (visited > ux0 ? visited : ux0) | (any_consulted ? 0x80000000u : (ux_word & 0x80000000u));
    const uint32_t now = stop >= 0 ? ((uint32_t)stop | kCntBroke) : full;
    if (now != old) {
      E.cnt_b[pos] = now;
      E.chg[atomicAdd(n_chg, 1u)] = pos;
    }
    if (list_it) E.consulted[atomicAdd(&E.ctl->n_consulted, 1u)] = pos;
  }
  // the ray's marks between the old and the new length count / stop counting under the NEXT lengths
  {
    const uint32_t vo = eo_visited(old);
    const uint32_t lo = vo < visited ? vo : visited, hi0 = vo < visited ? visited : vo, hi = hi0 < u0 ? hi0 : u0;
    for (uint32_t k = lo + lane; k < hi; k += 64u) {
      const uint32_t j = E.where[wk.base + k];
      if (visited > vo) atomicOr(&E.bits_b[j >> 6], 1ull << (j & 63u));
      else atomicAnd(&E.bits_b[j >> 6], ~(1ull << (j & 63u)));
    }
  }
}

// Queue ray p for the next round: the input of its visit of step `step` changed.  dirty[p] = 0: not queued; else
// [31] | 0x3fffff - (the EARLIEST such step): atomicMax keeps the smallest step, and the ray is evaluated again from there
// (eo2_eval_ray) instead of from its first voxel.
__device__ __forceinline__ void eo2_mark_dirty(const EoView& E, uint32_t p, uint32_t step, uint32_t* out, uint32_t* n_out) {
  const uint32_t v = 0x80000000u | (kEoStepMax - (step < kEoStepMax ? step : kEoStepMax));
  if (atomicMax(&E.dirty[p], v) == 0u) out[atomicAdd(n_out, 1u)] = p;
}

// ONE changed ray, by a whole wavefront: its marks between the old and the new length toggled — dirty their readers;
// then the new length becomes current.
__device__ __forceinline__ void eo2_propagate_ray(const EoView& E, const FrameParams& F, uint32_t pos, EoWaveLds& W, uint32_t* out, uint32_t* n_out) {
  const uint32_t lane = lane_id();
  const uint32_t old = eo2_ld(&E.cnt_a[pos]), now = eo2_ld(&E.cnt_b[pos]);
  const uint32_t vo = eo_visited(old), vn = eo_visited(now);
  const uint32_t lo = vo < vn ? vo : vn, hi = vo < vn ? vn : vo;
  EoWalk wk;
  wk.begin(E, pos);
  // (the steps below lo are not looked at: the walk starts at the batch that holds lo — a multiple of 64 below the seed's steps)
  wk.k0 = lo < wk.u0 ? (lo & ~63u) : wk.u0;
  const unsigned long long n_m = E.ctl->st.n_marks;
  uint32_t g, a, b;
  while (wk.next(E, F, W, lane, hi, g, a, b)) {
    const uint32_t k = g + lane;
    if (k >= a && k < b && k >= lo) {
      const uint32_t slot = (uint32_t)(W.keys[lane] >> 32);
      const uint64_t t = ((uint64_t)pos << 22) | k;
      // the next mark of the slot that counts under the new lengths: its owner reads something else now
      const unsigned long long j = k < wk.u0 ? (unsigned long long)E.where[wk.base + k] + 1ull : (unsigned long long)eo2_lower_bound(E, slot, t);
      const unsigned long long m = eo2_next_set(E.bits_b, j, n_m);
      if (m < n_m) {
        const uint64_t key = E.keys[m];
        const uint32_t p = (uint32_t)(key >> 22) & 0x3fffffu;
        if ((uint32_t)(key >> 44) == slot && p != pos) eo2_mark_dirty(E, p, (uint32_t)key & 0x3fffffu, out, n_out);
      }
      for (uint32_t xi = eo2_ld(&E.tab[slot].z); xi != 0u;) {
        const unsigned long long kx = eo2_ld64(&E.xnode[2u * xi]), hn = eo2_ld64(&E.xnode[2u * xi + 1u]);
        const uint32_t p = (uint32_t)(kx >> 22);
        if (kx > t && p != pos) eo2_mark_dirty(E, p, (uint32_t)kx & 0x3fffffu, out, n_out);
        xi = (uint32_t)(hn >> 32);
      }
      if (k < wk.u0) {   // the mark's bit under the current lengths follows
        const uint32_t jm = E.where[wk.base + k];
        if (vn > vo) atomicOr(&E.bits_a[jm >> 6], 1ull << (jm & 63u));
        else atomicAnd(&E.bits_a[jm >> 6], ~(1ull << (jm & 63u)));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) E.cnt_a[pos] = now;
}

// One bulk round = two launches (grid-stride over the list, one wavefront per ray):
//   k_eo2_eval      : rays of list[r & 1] (round 0: every ray of the frame) -> change list
//   k_eo2_propagate : change list -> list[(r + 1) & 1]
__global__ void __launch_bounds__(256) k_eo2_eval(EoBatch Bt, uint32_t round) {
  __shared__ EoWaveLds s_w[4];
  const EoView& E = Bt.v[blockIdx.y];
  EoCtl* ctl = E.ctl;
  if (ctl->fail) return;
  const uint32_t n = ctl->n_in[round];
  const uint32_t* list = E.list[round & 1u];
  const uint32_t wave = threadIdx.x >> 6, w0 = blockIdx.x * 4u + wave, nw = gridDim.x * 4u;
  if (w0 >= n) return;
  const FrameParams F = *E.F;
  for (uint32_t i = w0; i < n; i += nw) eo2_eval_ray(E, F, list[i], s_w[wave], &ctl->n_chg[round]);
}
__global__ void __launch_bounds__(256) k_eo2_propagate(EoBatch Bt, uint32_t round) {
  __shared__ EoWaveLds s_w[4];
  const EoView& E = Bt.v[blockIdx.y];
  EoCtl* ctl = E.ctl;
  if (ctl->fail) return;
  const uint32_t n = ctl->n_chg[round];
  const uint32_t wave = threadIdx.x >> 6, w0 = blockIdx.x * 4u + wave, nw = gridDim.x * 4u;
  if (blockIdx.x == 0 && threadIdx.x == 0 && n) ctl->rounds = round + 1u;
  if (w0 >= n) return;
  const FrameParams F = *E.F;
  for (uint32_t i = w0; i < n; i += nw) eo2_propagate_ray(E, F, E.chg[i], s_w[wave], E.list[(round + 1u) & 1u], &ctl->n_in[round + 1u]);
}

// The remaining rounds, by ONE workgroup (barriers instead of launches), until nothing is dirty.  Enqueued after the
// PREVIOUS frame's marks have entered `plain`: rays whose result depends on what earlier frames left there go first.
constexpr uint32_t kEoFinishThreads = 1024;
__global__ void __launch_bounds__(kEoFinishThreads) k_eo2_finish(EoView E, uint32_t first_round, uint32_t chained) {
  __shared__ EoWaveLds s_w[kEoFinishThreads / 64];
  __shared__ uint32_t s_n, s_stop;
  EoCtl* ctl = E.ctl;
  const uint32_t wave = threadIdx.x >> 6, nw = kEoFinishThreads / 64;
  const FrameParams F = *E.F;
  uint32_t cur = first_round & 1u;
  if (threadIdx.x == 0) {
    // (a predecessor that fell back to the host-driven loop has not entered its marks yet: this frame follows it there)
    if (chained && eo2_ld(E.committed) != F.eo_frame) atomicOr(&ctl->fail, kEoFailChain);
    ctl->fin_in[cur] = ctl->n_in[first_round];
    ctl->fin_in[cur ^ 1u] = 0u;
    ctl->fin_chg = 0u;
    s_stop = ctl->fail;
  }
  __syncthreads();
  if (s_stop) {
    if (threadIdx.x == 0) atomicOr(&E.C->err, kErrExact);
    return;
  }
  if (chained) {
    // every frame before this one has entered its marks: was the assumption about the slot of hash 0 right?
    __shared__ uint32_t s_redo;
    if (threadIdx.x == 0) {
      const unsigned long long actual = E.plain[(uint32_t)(F.observed_offset & kSetMask)];
      s_redo = actual != ctl->assumed0 ? 1u : 0u;
      ctl->assumed0 = actual;
    }
    __syncthreads();
    if (s_redo) {
      const uint32_t nc = eo2_ld(&ctl->n_consulted);
      for (uint32_t i = threadIdx.x; i < nc; i += kEoFinishThreads) eo2_mark_dirty(E, eo2_ld(&E.consulted[i]), 0u, E.list[cur], &ctl->fin_in[cur]);
    }
    __syncthreads();
  }
  for (uint32_t it = 0;; ++it) {
    if (threadIdx.x == 0) {
      s_n = eo2_ld(&ctl->fin_in[cur]);
      s_stop = eo2_ld(&ctl->fail) | ((it >= kEoFinishRounds || s_n > kEoFinishRays) ? kEoFailRounds : 0u);
      if (s_n && !s_stop) ctl->rounds += 1u;
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (n == 0u || s_stop) break;
    for (uint32_t i = wave; i < n; i += nw) eo2_eval_ray(E, F, eo2_ld(&E.list[cur][i]), s_w[wave], &ctl->fin_chg);
    __syncthreads();
    if (threadIdx.x == 0) {
      s_n = eo2_ld(&ctl->fin_chg);
      ctl->fin_in[cur] = 0u;
    }
    __syncthreads();
    const uint32_t nc = s_n;
    for (uint32_t i = wave; i < nc; i += nw) eo2_propagate_ray(E, F, eo2_ld(&E.chg[i]), s_w[wave], E.list[cur ^ 1u], &ctl->fin_in[cur ^ 1u]);
    __syncthreads();
    if (threadIdx.x == 0) ctl->fin_chg = 0u;
    cur ^= 1u;
  }
  if (threadIdx.x == 0 && s_stop) {
    atomicOr(&ctl->fail, s_stop);
    atomicOr(&E.C->err, kErrExact);
  }
}

// The frame's marks enter the reference's table: per slot, the hash of the last valid mark in time order; the slot's
// entry of the per-frame table is cleared for the frame slot's next frame.  After a failure only the clearing happens.
__global__ void __launch_bounds__(256) k_eo2_commit(EoView E) {
  const bool ok = E.ctl->fail == 0u;
  for (uint32_t slot = blockIdx.x * 256u + threadIdx.x; slot < (1u << kSetBits); slot += gridDim.x * 256u) {
    const uint4 e = E.tab[slot];
    if (e.x == e.y && e.z == 0u) continue;
    E.tab[slot] = make_uint4(0u, 0u, 0u, 0u);
    if (!ok) continue;
    bool found = false;
    uint64_t best = 0;
    uint32_t hash = 0;
    if (e.y > e.x) {
      const long long j = eo2_prev_set(E.bits_a, e.y);
      if (j >= (long long)e.x) {
        found = true;
        best = E.keys[j] & kEoLow44;
        hash = E.vals[j];
      }
    }
    for (uint32_t xi = e.z; xi != 0u;) {
      const unsigned long long k = E.xnode[2u * xi], hn = E.xnode[2u * xi + 1u];
      if (((uint32_t)k & 0x3fffffu) < eo_visited(E.cnt_a[(uint32_t)(k >> 22)]) && (!found || k > best)) {
        found = true;
        best = k;
        hash = (uint32_t)hn;
      }
      xi = (uint32_t)(hn >> 32);
    }
    if (found) E.plain[slot] = (uint64_t)hash;
  }
  if (ok && blockIdx.x == 0 && threadIdx.x == 0) *E.committed = E.F->eo_frame + 1u;
}

// start of a frame's fix point: counters (n_x = 1: node 0 is the end of a chain)
__global__ void __launch_bounds__(64) k_eo2_begin(EoBatch Bt) {
  const EoView& E = Bt.v[blockIdx.x];   // one workgroup per frame of the batch
  EoCtl* ctl = E.ctl;
  uint32_t* w = (uint32_t*)ctl;
  for (uint32_t i = threadIdx.x; i < sizeof(EoCtl) / 4u; i += 64) w[i] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    ctl->n_x = 1u;
    ctl->sw_prev = 1u;   // (long rays: "not at the fixed point yet" — the first sweep runs, and looks at every ray)
    ctl->sw_full = 1u;
    ctl->assumed0 = E.plain[(uint32_t)(E.F->observed_offset & kSetMask)];
  }
}

}  // namespace ksk
