// ks_k_apply_xl.h — the runs of more than kXLongRun updates WITHOUT their serial chain of 21 x n dependent additions.
//
// The voxel that holds the sensor is updated by every ray of a frame (2.4e5 bundles at 1280x720 / 2 cm): one run whose 21
// class sums p_c <- fl(p_c + x_i) are, taken literally, 21 chains of n dependent f32 additions — 3.2 ms in k_apply_xlong,
// the floor of the whole update stage (profiles/r06_sq_c4_merged_before.txt).  They need not be evaluated as chains:
//
//   While p stays inside ONE BINADE [2^E, 2^(E+1)) every representable value there is a multiple of u = 2^(E-23), p = -M u
//   with an integer M in [2^23, 2^24), and the correctly rounded sum of p and an increment x <= 0 is
//       fl(p + x) = -(M + R) u,   R = |x| / u rounded to the nearest integer,
//   INDEPENDENT of M unless |x| / u lies exactly half-way between two integers (then the tie goes to the even M + R).  Over a
//   chunk of updates without such a tie the chain collapses into the exact integer sum M + R_1 + ... + R_k, and the R_i of
//   different updates are independent of each other: they are evaluated a lane per update, reduced over the wavefront, and
//   the chunks of a run — thousands — are evaluated side by side.  What remains serial is one integer addition per chunk
//   and class (k_xl_walk), and the few chunks where the shortcut does not hold: a tie, an increment that is not <= 0, or the
//   sum leaving the binade (|p| doubles at most a couple of times per run) — those are replayed update by update, exactly
//   as k_apply_xlong does.  The binade of p when a chunk begins is not known to the wavefront that evaluates the chunk (the
//   voxel that has just become the sensor's sees ten times the updates it saw as a neighbour: its sums grow by factors of
//   tens within one run); it evaluates the run's starting binade E0_c (per class) and the kXlBinades - 1 after it, and the
//   walk takes whichever p is in (beyond them: replay).
//
//   The TSDF half of such a voxel: its distance sits at the clamp, +truncation, and stays there while every update is
//   "saturating" (k_apply_xlong's own shortcut, evaluated per update against the largest weight the voxel can have); its
//   WEIGHT w <- min(max_weight, fl(w + uw_i)) is a 22nd chain of the same kind — positive, growing — until it reaches the
//   clamp, where it stays.  k_xl_measure checks the state the run starts from, k_xl_chunks the updates; a run that fails
//   either, a colour-blending context, or a run where more than an eighth of the chunks turn out to need a replay (a voxel
//   in its first frame next to the sensor: its sums start at -0.6 and cross twenty binades) goes to k_apply_xlong as before
//   (the fall-back list), so the worst case costs what it always did.
//
// Same arithmetic as the reference's sequential loop ([K:src/semantic_integrator_base.cpp:283-380] through k_apply_xlong):
// the records are identical bit for bit — tests/test_apply_runs_gpu.py, tests/test_emu_parity.py, and the full-size frames
// against the real sources.
#pragma once
#include "ks_k_apply.h"

namespace ksk {

constexpr uint32_t kXlMaxRuns = 4096;        // runs on the parallel path per frame (more: the fall-back list)
constexpr uint32_t kXlSub = 4;               // updates per lane and chunk
constexpr uint32_t kXlChunk = 64 * kXlSub;   // updates per chunk: one wavefront, kXlSub updates per lane (the 176 reductions per chunk cost more than the rounding)
constexpr int kXlBinades = 8;                // binades evaluated per chunk and chain: the starting one and the next seven
constexpr int kXlChains = kNumLabels + 1;    // the 21 class sums (negative, falling) and the weight (positive, rising)
enum : uint32_t { kXlOk = 1u, kXlBadTsdf = 2u };

struct XlRun {
  unsigned long long start;   // index of the run's first pair
  uint32_t len;               // updates
  uint32_t first_chunk;       // index of its first chunk summary
  uint32_t vox;
  uint32_t flags;             // kXlOk: on the parallel path (k_xl_number); kXlBadTsdf: some update may move the distance off the clamp (k_xl_chunks)
  int32_t e0[kXlChains];      // biased exponent of every chain when the run begins
  uint32_t pad[4];
};
static_assert(sizeof(XlRun) == 128, "XlRun");
struct XlChunk {              // one chunk of 64 updates, evaluated in binade E0_c + k of every chain c, k < kXlBinades
  int32_t s[kXlBinades][kXlChains];   // sum of the rounded increments, in units of that binade's spacing
  uint32_t tie[kXlBinades];   // bit c: some increment of chain c is a rounding tie in that binade, or out of range
  unsigned long long active[kXlSub];  // which of the updates carry a semantic update at all (update 64 j + lane: bit lane of word j)
  uint32_t cnt, pad;
};
struct XlHeader {
  uint32_t n_listed;          // runs k_xl_measure looked at (<= kXlMaxRuns)
  uint32_t n_runs;            // of them, on the parallel path: xl_idx[0 .. n_runs)
  uint32_t n_chunks;          // their chunks
  uint32_t n_fallback;        // runs k_apply_xlong takes (xl_fb_list)
  // since the context was created (ks_update_stats): runs walked, runs handed to k_apply_xlong, chunks, chunks replayed
  unsigned long long tot_walked, tot_fallback, tot_chunks, tot_replayed;
};

// Sum over the wavefront, the same value in every lane.  Six DPP adds (no LDS round trips: the 176 sums of a chunk are what
// k_xl_chunks spends its time on) — quad swaps, row rotations, then the row broadcasts of gfx9; the total lands in lane 63.
#ifndef KS_WAVE_SUM_I32
#define KS_WAVE_SUM_I32(v) ksk::wave_sum_i32_dpp(v)
__device__ __forceinline__ int wave_sum_i32_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);    // quad_perm:[1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);    // quad_perm:[2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);   // row_ror:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
#endif
__device__ __forceinline__ int wave_sum_i32(int v) { return KS_WAVE_SUM_I32(v); }

// The state-independent half of an update of voxel `v_voxel_origin` by the ray of `d` (k_apply_xlong's first_half).
__device__ __forceinline__ void xl_tsdf_operands(const FrameParams& F, const RayDesc& d, const f3& v_voxel_origin, float& sdf, float& uw) {
  const TsdfParams& Pm = F.tsdf;
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
__device__ __forceinline__ f3 xl_voxel_origin(const FrameParams& F, const TileTable& T, uint32_t vox) {
  const VoxelRef v = voxel_ref(T, vox);
  const float vs = F.tsdf.voxel_size;
  const f3 c = {((float)v.vx + 0.5f) * vs, ((float)v.vy + 0.5f) * vs, ((float)v.vz + 0.5f) * vs};
  return sub3(c, F.T.t);
}

// Thread r: the length of listed run r (the list holds only where it starts) and the state its voxel starts from.
template <int COLOR_MODE>
__global__ void __launch_bounds__(256) k_xl_measure(FrameParams F, unsigned long long n_pairs, const uint64_t* __restrict__ pairs,
                                                    Pool P, const unsigned long long* __restrict__ xlong_list, const Counters* C,
                                                    XlRun* __restrict__ runs) {
  const uint32_t n_x = C->n_xlong, r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_x || r >= kXlMaxRuns) return;
  const unsigned long long start = xlong_list[r];
  const uint32_t vox = (uint32_t)(pairs[start] >> F.seq_bits);
  // the run's end: gallop, then bisect (the pairs are sorted by voxel: "same voxel" is a prefix property)
  unsigned long long lo = kXLongRun, hi;   // pairs[start + lo] is known to be this voxel's (k_find_long)
  for (unsigned long long step = kXLongRun;; step <<= 1) {
    hi = lo + step;
    if (start + hi >= n_pairs) { hi = n_pairs - start; break; }
    if ((uint32_t)(pairs[start + hi] >> F.seq_bits) != vox) break;
    lo = hi;
  }
  // invariant: element lo belongs to the run, element hi does not (or is the end of the list)
  while (hi - lo > 1ull) {
    const unsigned long long mid = lo + ((hi - lo) >> 1);
    if ((uint32_t)(pairs[start + mid] >> F.seq_bits) == vox) lo = mid;
    else hi = mid;
  }
  const uint32_t* rec = (const uint32_t*)(P.vox + (size_t)vox * 8);
  const float w = __uint_as_float(rec[1]);
  // the distance at the clamp; a weight the integer form can carry (and that no update can leave below kEps: w + uw >= w)
  bool ok = (COLOR_MODE != KS_COLOR_MODE_COLOR) && __uint_as_float(rec[0]) == F.tsdf.trunc && w >= kEps && w <= F.tsdf.max_weight &&
            F.tsdf.max_weight < 1e30f && hi < 0xffffffffull;
  XlRun R;
  for (int c = 0; c < kNumLabels; ++c) {
    const uint32_t b = rec[4 + c];
    const uint32_t e = (b >> 23) & 0xffu;
    // a class sum the shortcut can carry: negative, normal, finite (anything else: the serial kernel)
    ok = ok && (b >> 31) == 1u && e >= 1u && e <= 253u;
    R.e0[c] = (int32_t)e;
  }
  R.e0[kNumLabels] = (int32_t)((rec[1] >> 23) & 0xffu);
  R.start = start;
  R.len = (uint32_t)hi;
  R.first_chunk = 0u;
  R.vox = vox;
  R.flags = ok ? kXlOk : 0u;
  R.pad[0] = R.pad[1] = R.pad[2] = R.pad[3] = 0u;
  runs[r] = R;
}

// ONE workgroup: the listed runs are numbered — parallel path (xl_idx, with their chunk offsets) / fall-back list.
__global__ void __launch_bounds__(1024) k_xl_number(const unsigned long long* __restrict__ xlong_list, const Counters* C,
                                                    XlRun* __restrict__ runs, uint32_t* __restrict__ xl_idx,
                                                    unsigned long long* __restrict__ fb_list, XlHeader* __restrict__ hdr, uint32_t cap_chunks) {
  constexpr uint32_t PER = kXlMaxRuns / 1024;
  __shared__ uint32_t s_runs[1024], s_chunks[1024];
  const uint32_t n_x = C->n_xlong, n = n_x < kXlMaxRuns ? n_x : kXlMaxRuns, t = threadIdx.x;
  uint32_t my_runs = 0u, my_chunks = 0u;
  for (uint32_t k = 0; k < PER; ++k) {
    const uint32_t r = t * PER + k;
    if (r < n && (runs[r].flags & kXlOk)) {
      ++my_runs;
      my_chunks += (runs[r].len + kXlChunk - 1u) / kXlChunk;
    }
  }
  s_runs[t] = my_runs;
  s_chunks[t] = my_chunks;
  __syncthreads();
  // inclusive scan over the 1024 threads (Hillis-Steele; twice 10 steps)
  for (uint32_t o = 1; o < 1024u; o <<= 1) {
    const uint32_t a = t >= o ? s_runs[t - o] : 0u, b = t >= o ? s_chunks[t - o] : 0u;
    __syncthreads();
    s_runs[t] += a;
    s_chunks[t] += b;
    __syncthreads();
  }
  uint32_t run_no = s_runs[t] - my_runs, chunk_no = s_chunks[t] - my_chunks;
  __shared__ uint32_t s_n_fb, s_acc_runs[1024], s_acc_chunks[1024];
  if (t == 0) s_n_fb = 0u;
  __syncthreads();
  uint32_t acc_runs = 0u, acc_chunks = 0u;   // what this thread's accepted runs reach up to
  for (uint32_t k = 0; k < PER; ++k) {
    const uint32_t r = t * PER + k;
    if (r >= n) break;
    bool par = (runs[r].flags & kXlOk) != 0u;
    if (par) {
      const uint32_t ch = (runs[r].len + kXlChunk - 1u) / kXlChunk;
      if (chunk_no + ch <= cap_chunks) {
        runs[r].first_chunk = chunk_no;
        xl_idx[run_no] = r;
        acc_runs = run_no + 1u;
        acc_chunks = chunk_no + ch;
      } else {
        runs[r].flags = 0u;   // no room for its chunk summaries: the serial kernel's
        par = false;
      }
      ++run_no;
      chunk_no += ch;
    }
    if (!par) fb_list[atomicAdd(&s_n_fb, 1u)] = runs[r].start;
  }
  // (the accepted runs are a prefix of the numbering: the largest of the per-thread marks is the count)
  s_acc_runs[t] = acc_runs;
  s_acc_chunks[t] = acc_chunks;
  __syncthreads();
  for (uint32_t o = 512u; o >= 1u; o >>= 1) {
    if (t < o) {
      s_acc_runs[t] = s_acc_runs[t] > s_acc_runs[t + o] ? s_acc_runs[t] : s_acc_runs[t + o];
      s_acc_chunks[t] = s_acc_chunks[t] > s_acc_chunks[t + o] ? s_acc_chunks[t] : s_acc_chunks[t + o];
    }
    __syncthreads();
  }
  // (more runs listed than this path numbers: the rest go to the serial kernel as they are)
  for (uint32_t r = kXlMaxRuns + t; r < n_x; r += 1024u) fb_list[atomicAdd(&s_n_fb, 1u)] = xlong_list[r];
  __syncthreads();
  if (t == 0) {
    hdr->n_listed = n;
    hdr->n_runs = s_acc_runs[0];
    hdr->n_chunks = s_acc_chunks[0];
    hdr->n_fallback = s_n_fb;
    hdr->tot_fallback += s_n_fb;
  }
}

// A wavefront per chunk (grid-stride over all chunks of all runs): a lane takes updates lane, 64 + lane, ... of the chunk,
// rounds their increments in every binade, and adds its own up before the wavefront's sums are taken.
__global__ void __launch_bounds__(256) k_xl_chunks(FrameParams F, const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                   const float* __restrict__ deltas, TileTable T, XlRun* __restrict__ runs,
                                                   const uint32_t* __restrict__ xl_idx, const XlHeader* __restrict__ hdr,
                                                   XlChunk* __restrict__ chunks) {
  const uint32_t n_runs = hdr->n_runs, n_chunks = hdr->n_chunks;
  const uint32_t lane = lane_id();
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t ch = wave; ch < n_chunks; ch += n_waves) {
    // the run of this chunk: the last one whose first chunk is <= ch
    uint32_t lo = 0u, hi = n_runs;
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (runs[xl_idx[mid]].first_chunk <= ch) lo = mid;
      else hi = mid;
    }
    const uint32_t ri = xl_idx[lo];
    const XlRun& R = runs[ri];
    const uint32_t b = ch - R.first_chunk;
    const TsdfParams& Pm = F.tsdf;
    const f3 vo = xl_voxel_origin(F, T, R.vox);
    // the updates of this lane: all their loads are requested before the first is used
    bool in[kXlSub];
    RayDesc d[kXlSub];
    uint32_t rp[kXlSub];
#pragma unroll
    for (uint32_t j = 0; j < kXlSub; ++j) {
      const uint32_t off = b * kXlChunk + j * 64u + lane;
      in[j] = off < R.len;
      const uint64_t key = pairs[R.start + (in[j] ? off : 0u)];
      rp[j] = (uint32_t)key & F.point_mask;
    }
#pragma unroll
    for (uint32_t j = 0; j < kXlSub; ++j) d[j] = rays[ray_index(F, rp[j])];
    float uw[kXlSub];
    uint32_t kind[kXlSub];
    bool bad = false;
    XlChunk& O = chunks[ch];
#pragma unroll
    for (uint32_t j = 0; j < kXlSub; ++j) {
      float sdf;
      xl_tsdf_operands(F, d[j], vo, sdf, uw[j]);
      // the distance stays at +truncation if every update is saturating (k_apply_xlong's shortcut), judged against the
      // largest weight the voxel can have when the update arrives: the right-hand side grows with it
      const float nw_most = Pm.max_weight + uw[j];
      const bool sat = (sdf - Pm.trunc) * uw[j] >= 1e-6f * Pm.trunc * nw_most;
      bad = bad || (in[j] && !((uw[j] >= 0.0f) && sat));
      kind[j] = in[j] ? ((d[j].info >> 8) & 3u) : 0u;
      const unsigned long long act = __ballot(kind[j] != 0u);
      if (lane == 0) O.active[j] = act;
    }
    const bool tsdf_bad = __ballot(bad) != 0ull;
    uint32_t tie[kXlBinades];
#pragma unroll
    for (int k = 0; k < kXlBinades; ++k) tie[k] = 0u;
#pragma unroll
    for (int l = 0; l < kXlChains; ++l) {
      const int E0 = R.e0[l] - 127;
      int32_t acc[kXlBinades];
#pragma unroll
      for (int k = 0; k < kXlBinades; ++k) acc[k] = 0;
      uint32_t my_tie = 0u;   // bit k: one of this lane's increments is a tie in binade k (all bits: out of range)
#pragma unroll
      for (uint32_t j = 0; j < kXlSub; ++j) {
        // the magnitude of this update's increment of chain l, and whether it has one at all
        float x = 0.0f;
        bool has = false;
        if (l < kNumLabels) {
          has = kind[j] != 0u;
          if (kind[j] == 1u) x = -(((uint32_t)l == (d[j].info & 0xffu)) ? d[j].d_match : d[j].d_non);
          else if (kind[j] == 2u) x = -deltas[(size_t)rp[j] * kNumLabels + l];
        } else {
          has = in[j];
          x = uw[j];
        }
        // in units of the spacing 2^(E - 23) of binade E = E0 + k; a tie, or an increment the integer sum cannot carry
        // (NaN, of the wrong sign, 2^22 spacings and more): replay
        float t = ldexpf(x, 23 - E0);
        const bool out = has && !(x >= 0.0f && t < 4194304.0f);
        if (out) my_tie = ~0u;
#pragma unroll
        for (int k = 0; k < kXlBinades; ++k) {
          const float r = rintf(t);
          if (has && fabsf(t - r) == 0.5f) my_tie |= 1u << k;
          acc[k] += (has && !out) ? (int)r : 0;
          t *= 0.5f;   // (exact; a result that underflows rounds to 0 spacings and is no tie, as it should)
        }
      }
      const unsigned long long any_tie = __ballot(my_tie != 0u);
      int32_t keep = 0;   // lane k < kXlBinades keeps the sum of binade k
#pragma unroll
      for (int k = 0; k < kXlBinades; ++k) {
        if (any_tie && __ballot((my_tie >> k) & 1u)) tie[k] |= 1u << l;
        const int sk = wave_sum_i32(acc[k]);
        if (lane == (uint32_t)k) keep = sk;
      }
      if (lane < (uint32_t)kXlBinades) O.s[lane][l] = keep;
    }
    if (lane < (uint32_t)kXlBinades) {
      uint32_t tk = 0u;
#pragma unroll
      for (int k = 0; k < kXlBinades; ++k)
        if (lane == (uint32_t)k) tk = tie[k];
      O.tie[lane] = tk;
    }
    if (lane == 0) {
      const uint32_t left = R.len - b * kXlChunk;
      O.cnt = left < kXlChunk ? left : kXlChunk;
      if (tsdf_bad) atomicOr(&runs[ri].flags, kXlBadTsdf);
    }
  }
}

// A wavefront per run: the integer walk over its chunk summaries, lanes 0..20 = the class sums, lane 21 = the weight; a
// chunk the shortcut cannot carry is replayed update by update (lanes = its 64 updates for the gather, then the chains).
template <int COLOR_MODE>
__global__ void __launch_bounds__(64) k_xl_walk(FrameParams F, const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                const float* __restrict__ deltas, TileTable T, Pool P, const uint32_t* __restrict__ label_lut,
                                                XlRun* __restrict__ runs, const uint32_t* __restrict__ xl_idx, XlHeader* __restrict__ hdr,
                                                const XlChunk* __restrict__ chunks, unsigned long long* __restrict__ fb_list) {
  __shared__ float s_inc[64][kNumLabels];   // (a replay takes 64 updates of the chunk at a time)
  __shared__ float s_uw[64];
  const uint32_t n_runs = hdr->n_runs;
  const int lane = (int)lane_id();
  const int cls = lane < kXlChains ? lane : 0;
  const bool is_w = lane == kNumLabels;
  const float max_w = F.tsdf.max_weight;
  for (uint32_t rk = blockIdx.x; rk < n_runs; rk += gridDim.x) {
    const XlRun& R = runs[xl_idx[rk]];
    const uint32_t n_ch = (R.len + kXlChunk - 1u) / kXlChunk;
    if (R.flags & kXlBadTsdf) {
      // some update may move the distance off the clamp: k_apply_xlong's, the record untouched so far
      if (lane == 0) {
        fb_list[atomicAdd(&hdr->n_fallback, 1u)] = R.start;
        atomicAdd(&hdr->tot_fallback, 1ull);
      }
      continue;
    }
    uint32_t n_replayed = 0u;
    const uint32_t replay_budget = n_ch / 8u + 8u;
    bool gave_up = false;
    uint32_t* rec = (uint32_t*)(P.vox + (size_t)R.vox * 8);
    float p = __uint_as_float(is_w ? rec[1] : rec[4 + (cls < kNumLabels ? cls : 0)]);   // the chain's value: a class sum (< 0) or the weight (> 0)
    const int e0 = R.e0[cls];
    const XlChunk* ch = chunks + R.first_chunk;
    // The binade index k = (exponent of p) - e0 only ever grows.  The summaries of the next chunks are requested ahead for
    // the binade p is in now and the one after it; a chunk that finds p further on reads its own (rare: a crossing).
    constexpr int PF = 4;
    int kt[PF];                 // binade index slot i of the queue was loaded for (per lane = per chain)
    int32_t qa[PF], qb[PF];     // sums in binade kt[i] / kt[i] + 1 of the chunk the slot holds
    uint32_t ta[PF], tb[PF];
    auto load_q = [&](int i, uint32_t b, int k) {
      const XlChunk& X = ch[b < n_ch ? b : n_ch - 1u];
      const int ka = k < kXlBinades ? k : kXlBinades - 1, kb = k + 1 < kXlBinades ? k + 1 : kXlBinades - 1;
      kt[i] = ka;
      qa[i] = X.s[ka][cls];
      qb[i] = X.s[kb][cls];
      ta[i] = X.tie[ka];
      tb[i] = X.tie[kb];
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_q(i, (uint32_t)i, 0);   // (e0 was read from this record: every chain starts in its binade 0)
    for (uint32_t b0 = 0; b0 < n_ch && !gave_up; b0 += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const uint32_t b = b0 + (uint32_t)i;
        if (b >= n_ch) break;
        const uint32_t bits = __float_as_uint(p);
        const int e = (int)((bits >> 23) & 0xffu);
        const int k = e - e0;
        int32_t S = 0;
        uint32_t tie = 1u;   // (a binade outside the evaluated ones: replay)
        if (k == kt[i]) { S = qa[i]; tie = (ta[i] >> cls) & 1u; }
        else if (k == kt[i] + 1 && k < kXlBinades) { S = qb[i]; tie = (tb[i] >> cls) & 1u; }
        else if (k >= 0 && k < kXlBinades) { S = ch[b].s[k][cls]; tie = (ch[b].tie[k] >> cls) & 1u; }
        // the slot just used is refilled for chunk b + PF, in the binade p is in NOW
        load_q(i, b + PF, k < 0 ? 0 : k);
        const uint32_t M = (bits & 0x7fffffu) | 0x800000u;
        // (of the chain's sign, finite, normal: what the integer form stands for; the sum stays strictly inside the binade)
        bool ok = !tie && (bits >> 31) == (is_w ? 0u : 1u) && e >= 1 && e <= 253 && (M + (uint32_t)S <= 0xfffffeu);
        // M + S < 2^24 is exact in f32, and so is the scaling
        float np = ldexpf((float)(M + (uint32_t)S), e - 150);
        if (is_w) {
          // the weight: at the clamp it stays (every update weight >= 0: k_xl_chunks); below it the sum must end below it
          // (then no update of the chunk met the clamp: the chain only rises)
          if (p == max_w) { ok = true; np = p; }
          else ok = ok && np < max_w;
        } else {
          np = -np;
        }
        if (__ballot(lane < kXlChains && !ok) == 0ull) {
          p = np;
        } else {
          if (++n_replayed > replay_budget) {
            // more chunks to replay than the serial kernel would be slower for: hand the run over, record untouched
            gave_up = true;
            break;
          }
          // replay the chunk, 64 updates at a time: gather (lane = update), then the 22 chains over them, update by update
          const XlChunk& X = ch[b];
          const int cnt_all = (int)X.cnt;
          for (uint32_t j = 0; j < kXlSub && (int)(j * 64u) < cnt_all; ++j) {
            const uint32_t off = b * kXlChunk + j * 64u + (uint32_t)lane;
            const bool in = off < R.len;
            const uint64_t key = pairs[R.start + (in ? off : 0u)];
            const uint32_t rp = (uint32_t)key & F.point_mask;
            const RayDesc d = rays[ray_index(F, rp)];
            const uint32_t kind = in ? ((d.info >> 8) & 3u) : 0u, lab = d.info & 0xffu;
            float sdf, uw;
            xl_tsdf_operands(F, d, xl_voxel_origin(F, T, R.vox), sdf, uw);
            s_uw[lane] = uw;
            if (kind == 2u) {
              const float* dl = deltas + (size_t)rp * kNumLabels;
#pragma unroll
              for (int l = 0; l < kNumLabels; ++l) s_inc[lane][l] = dl[l];
            } else {
              const float a = (kind == 1u) ? d.d_match : 0.0f, bn = (kind == 1u) ? d.d_non : 0.0f;
#pragma unroll
              for (int l = 0; l < kNumLabels; ++l) s_inc[lane][l] = ((uint32_t)l == lab) ? a : bn;
            }
            KS_WAVE_LDS_ORDER();
            const unsigned long long active = X.active[j];
            const int cnt = cnt_all - (int)(j * 64u) < 64 ? cnt_all - (int)(j * 64u) : 64;
            for (int kk = 0; kk < cnt; ++kk) {
              if (is_w) {
                const float nw = p + s_uw[kk];
                if (!(nw < kEps)) p = std_min(max_w, nw);
              } else if ((active >> kk) & 1ull) {
                p += s_inc[kk][cls < kNumLabels ? cls : 0];
              }
            }
            KS_WAVE_LDS_ORDER();
          }
        }
      }
    }
    if (lane == 0) {
      atomicAdd(&hdr->tot_chunks, (unsigned long long)n_ch);
      atomicAdd(&hdr->tot_replayed, (unsigned long long)n_replayed);
    }
    if (gave_up) {
      if (lane == 0) {
        fb_list[atomicAdd(&hdr->n_fallback, 1u)] = R.start;
        atomicAdd(&hdr->tot_fallback, 1ull);
      }
      continue;
    }
    if (lane == 0) atomicAdd(&hdr->tot_walked, 1ull);
    // argmax over lanes 0..20, first strict maximum
    int best = 0;
    float m = bcast_f(p, 0);
#pragma unroll
    for (int l = 1; l < kNumLabels; ++l) {
      const float x = bcast_f(p, l);
      if (x > m) { m = x; best = l; }
    }
    if (lane < kNumLabels) rec[4 + lane] = __float_as_uint(p);
    if (is_w) rec[1] = __float_as_uint(p);
    uint32_t color = rec[2];
    if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = label_lut[best];
    else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) color = rainbow_color_map((double)(float)exp((double)m));
    if (lane == 0) {
      // (the distance stays where it is: the clamp)
      rec[2] = color;
      rec[3] = (uint32_t)best;
      rec[25] = 1u;  // updated since the last voxel-level host sync
    }
  }
}

}  // namespace ksk
