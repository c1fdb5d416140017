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
//   The TSDF half of such a voxel is at its fixed point — distance clamped at +truncation, weight at max_weight — whenever
//   k_apply_xlong's own shortcuts apply (weight == max_weight and every update weight >= 0; distance == truncation and every
//   update saturating): k_xl_plan checks the state, k_xl_chunks the updates; a run that fails either, a colour-blending
//   context, or a run where more than an eighth of the chunks turn out to need a replay goes to k_apply_xlong as before
//   (the fall-back list), so the worst case costs what it always did.
//
// Same arithmetic as the reference's sequential loop ([K:src/semantic_integrator_base.cpp:283-380] through k_apply_xlong):
// the records are identical bit for bit — tests/test_apply_runs_gpu.py, tests/test_emu_parity.py, and the full-size frames
// against the real sources.
#pragma once
#include "ks_k_apply.h"

namespace ksk {

constexpr uint32_t kXlMaxRuns = 256;         // runs on the parallel path per frame (more: the fall-back list)
constexpr uint32_t kXlChunk = 64;            // updates per chunk = one wavefront
constexpr int kXlBinades = 8;                // binades evaluated per chunk and class: the starting one and the next seven
enum : uint32_t { kXlBadTsdf = 1u, kXlFallback = 2u };

struct XlRun {
  unsigned long long start;   // index of the run's first pair
  uint32_t len;               // updates
  uint32_t first_chunk;       // index of its first chunk summary
  uint32_t vox;
  uint32_t flags;             // kXlBadTsdf: some update does not leave the TSDF fixed point (set by k_xl_chunks)
  uint32_t pad[2];
  int32_t e0[24];             // biased exponent of every class sum when the run begins
};
struct XlChunk {              // one chunk of 64 updates, evaluated in binade E0_c + k of every class c, k < kXlBinades
  int32_t s[kXlBinades][kNumLabels];   // sum of the rounded increments, in units of that binade's spacing
  uint32_t tie[kXlBinades];   // bit c: some increment of class c is a rounding tie in that binade, or out of range
  uint32_t active_lo, active_hi;  // which of the 64 updates carry a semantic update at all
  uint32_t cnt, pad;
};
struct XlHeader {
  uint32_t n_runs;            // runs on the parallel path
  uint32_t n_chunks;          // their chunks
  uint32_t n_fallback;        // runs k_apply_xlong takes (the front of the xlong list is rewritten with them)
  uint32_t pad;
  // since the context was created (ks_update_stats): runs walked, runs handed to k_apply_xlong, chunks, chunks replayed
  unsigned long long tot_walked, tot_fallback, tot_chunks, tot_replayed;
};

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

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

// ONE workgroup.  Thread r < n_xlong: the length of run r (the list holds only where it starts), the state of its voxel;
// then the runs are numbered — parallel path / fall-back — and their chunks counted.
template <int COLOR_MODE>
__global__ void __launch_bounds__(256) k_xl_plan(FrameParams F, unsigned long long n_pairs, const uint64_t* __restrict__ pairs,
                                                 Pool P, unsigned long long* __restrict__ xlong_list, const Counters* C,
                                                 XlRun* __restrict__ runs, XlHeader* __restrict__ hdr, uint32_t cap_chunks) {
  __shared__ unsigned long long s_start[256];
  __shared__ uint32_t s_len[256], s_ok[256];
  const uint32_t n_x = C->n_xlong, r = threadIdx.x;
  // (the list was written with one atomic cursor per workgroup of k_find_long: any order; sorted here so that the numbering
  // — and with it nothing that is observable, but every trace and statistic — does not depend on timing)
  unsigned long long start = ~0ull;
  uint32_t len = 0u, ok = 0u;
  if (r < n_x && r < 256u) {
    start = xlong_list[r];
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
    len = (uint32_t)hi;
    const uint32_t* rec = (const uint32_t*)(P.vox + (size_t)vox * 8);
    ok = (COLOR_MODE != KS_COLOR_MODE_COLOR) && __uint_as_float(rec[1]) == F.tsdf.max_weight && __uint_as_float(rec[0]) == F.tsdf.trunc &&
         hi < 0xffffffffull;
    for (int c = 0; c < kNumLabels; ++c) {
      const uint32_t b = rec[4 + c];
      const uint32_t e = (b >> 23) & 0xffu;
      // a class sum the shortcut can carry: negative, normal, finite (anything else: the serial kernel)
      ok = ok && (b >> 31) == 1u && e >= 1u && e <= 253u;
    }
  }
  s_start[r] = start;
  s_len[r] = len;
  s_ok[r] = ok;
  __syncthreads();
  if (r == 0) {
    const uint32_t n = n_x < 256u ? n_x : 256u;
    // insertion sort by start (n is a handful)
    for (uint32_t i = 1; i < n; ++i) {
      const unsigned long long ks = s_start[i];
      const uint32_t kl = s_len[i], ko = s_ok[i];
      uint32_t j = i;
      for (; j > 0 && s_start[j - 1] > ks; --j) {
        s_start[j] = s_start[j - 1];
        s_len[j] = s_len[j - 1];
        s_ok[j] = s_ok[j - 1];
      }
      s_start[j] = ks;
      s_len[j] = kl;
      s_ok[j] = ko;
    }
    uint32_t n_runs = 0u, n_chunks = 0u, n_fb = 0u;
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t ch = (s_len[i] + kXlChunk - 1u) / kXlChunk;
      if (s_ok[i] && n_runs < kXlMaxRuns && n_chunks + ch <= cap_chunks) {
        XlRun& R = runs[n_runs++];
        R.start = s_start[i];
        R.len = s_len[i];
        R.first_chunk = n_chunks;
        R.vox = (uint32_t)(pairs[s_start[i]] >> F.seq_bits);
        R.flags = 0u;
        const uint32_t* rec = (const uint32_t*)(P.vox + (size_t)R.vox * 8);
        for (int c = 0; c < kNumLabels; ++c) R.e0[c] = (int32_t)((rec[4 + c] >> 23) & 0xffu);
        n_chunks += ch;
      } else {
        xlong_list[n_fb++] = s_start[i];
      }
    }
    // (more than 256 such runs in one frame: the rest keep their places behind the fall-back ones)
    for (uint32_t i = 256u; i < n_x; ++i) xlong_list[n_fb++] = xlong_list[i];
    hdr->n_runs = n_runs;
    hdr->n_chunks = n_chunks;
    hdr->n_fallback = n_fb;
    hdr->tot_fallback += n_fb;
  }
}

// A wavefront per chunk (grid-stride over all chunks of all runs): lane = update.
__global__ void __launch_bounds__(256) k_xl_chunks(FrameParams F, const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                   const float* __restrict__ deltas, TileTable T, XlRun* __restrict__ runs,
                                                   const XlHeader* __restrict__ hdr, XlChunk* __restrict__ chunks) {
  const uint32_t n_runs = hdr->n_runs, n_chunks = hdr->n_chunks;
  const uint32_t lane = lane_id();
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t ch = wave; ch < n_chunks; ch += n_waves) {
    // the run of this chunk: the last one whose first chunk is <= ch (a handful of runs: linear)
    uint32_t ri = 0u;
    for (uint32_t k = 1; k < n_runs; ++k)
      if (runs[k].first_chunk <= ch) ri = k;
    const XlRun& R = runs[ri];
    const uint32_t b = ch - R.first_chunk;
    const uint32_t off = b * kXlChunk + lane;
    const bool in = off < R.len;
    const uint64_t key = pairs[R.start + (in ? off : 0u)];
    const uint32_t rp = (uint32_t)key & F.point_mask;
    const RayDesc d = rays[ray_index(F, rp)];
    const VoxelRef v = voxel_ref(T, R.vox);
    const TsdfParams& Pm = F.tsdf;
    const f3 c = {((float)v.vx + 0.5f) * Pm.voxel_size, ((float)v.vy + 0.5f) * Pm.voxel_size, ((float)v.vz + 0.5f) * Pm.voxel_size};
    float sdf, uw;
    xl_tsdf_operands(F, d, sub3(c, F.T.t), sdf, uw);
    // k_apply_xlong's two shortcuts, with the voxel at (weight, distance) = (max_weight, truncation):
    const float my_nw = Pm.max_weight + uw;
    const bool my_sat = (my_nw < kEps) || ((sdf - Pm.trunc) * uw >= 1e-6f * Pm.trunc * my_nw);
    const bool tsdf_bad = __ballot(in && !((uw >= 0.0f) && my_sat)) != 0ull;
    const uint32_t kind = in ? ((d.info >> 8) & 3u) : 0u, lab = d.info & 0xffu;
    const unsigned long long active = __ballot(kind != 0u);
    const float a = d.d_match, bnon = d.d_non;
    const float* dl = deltas + (size_t)rp * kNumLabels;
    XlChunk& O = chunks[ch];
    uint32_t tie[kXlBinades];
#pragma unroll
    for (int k = 0; k < kXlBinades; ++k) tie[k] = 0u;
#pragma unroll
    for (int l = 0; l < kNumLabels; ++l) {
      float x = 0.0f;
      if (kind == 1u) x = ((uint32_t)l == lab) ? a : bnon;
      else if (kind == 2u) x = dl[l];
      const int E0 = R.e0[l] - 127;
      // |x| in units of the spacing 2^(E - 23) of binade E = E0 + k; a tie, or an increment the integer sum cannot carry
      // (NaN, positive, 2^22 spacings and more): replay
      const float t0 = ldexpf(-x, 23 - E0);
      const bool out = kind != 0u && !(x <= 0.0f && t0 < 4194304.0f);
      const unsigned long long any_out = __ballot(out);
      int32_t keep = 0;   // lane k < kXlBinades keeps the sum of binade k
#pragma unroll
      for (int k = 0; k < kXlBinades; ++k) {
        const float t = ldexpf(-x, 23 - E0 - k);
        const float r = rintf(t);
        if (any_out || __ballot(kind != 0u && fabsf(t - r) == 0.5f)) tie[k] |= 1u << l;
        const int sk = wave_sum_i32((kind != 0u && !out) ? (int)r : 0);
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
      O.active_lo = (uint32_t)active;
      O.active_hi = (uint32_t)(active >> 32);
      const uint32_t left = R.len - b * kXlChunk;
      O.cnt = left < kXlChunk ? left : kXlChunk;
      if (tsdf_bad) atomicOr(&runs[ri].flags, kXlBadTsdf);
    }
  }
}

// A wavefront per run: the integer walk over its chunk summaries, lanes 0..20 = the classes; a chunk the shortcut cannot
// carry is replayed update by update (lanes = its 64 updates for the gather, then the classes again).
template <int COLOR_MODE>
__global__ void __launch_bounds__(64) k_xl_walk(FrameParams F, const uint64_t* __restrict__ pairs, const RayDesc* __restrict__ rays,
                                                const float* __restrict__ deltas, Pool P, const uint32_t* __restrict__ label_lut,
                                                XlRun* __restrict__ runs, XlHeader* __restrict__ hdr,
                                                const XlChunk* __restrict__ chunks, unsigned long long* __restrict__ xlong_list) {
  __shared__ float s_inc[kXlChunk][kNumLabels];
  const uint32_t n_runs = hdr->n_runs;
  const int lane = (int)lane_id();
  const int cls = lane < kNumLabels ? lane : 0;
  for (uint32_t ri = blockIdx.x; ri < n_runs; ri += gridDim.x) {
    const XlRun& R = runs[ri];
    const uint32_t n_ch = (R.len + kXlChunk - 1u) / kXlChunk;
    if (R.flags & kXlBadTsdf) {
      // not at the TSDF fixed point all the way through: k_apply_xlong's, untouched so far
      if (lane == 0) {
        xlong_list[atomicAdd(&hdr->n_fallback, 1u)] = R.start;
        atomicAdd(&hdr->tot_fallback, 1ull);
      }
      continue;
    }
    uint32_t n_replayed = 0u;
    const uint32_t replay_budget = n_ch / 8u + 64u;
    bool gave_up = false;
    uint32_t* rec = (uint32_t*)(P.vox + (size_t)R.vox * 8);
    float p = __uint_as_float(rec[4 + cls]);
    const int e0 = R.e0[cls];
    const XlChunk* ch = chunks + R.first_chunk;
    // The binade index k = (exponent of p) - e0 only ever grows.  The summaries of the next chunks are requested ahead for
    // the binade p is in now and the one after it; a chunk that finds p further on reads its own (rare: a crossing).
    constexpr int PF = 4;
    int kt[PF];                 // binade index slot i of the queue was loaded for (per lane = per class)
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
    for (int i = 0; i < PF; ++i) load_q(i, (uint32_t)i, 0);   // (e0 was read from this record: p starts in binade 0)
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
        // (negative, finite, normal: what the integer form stands for)
        const bool ok = !tie && (bits >> 31) == 1u && e >= 1 && e <= 253 && (M + (uint32_t)S <= 0xfffffeu);
        if (__ballot(lane < kNumLabels && !ok) == 0ull) {
          // p = -(M + S) 2^(E - 23): M + S < 2^24 is exact in f32, and so is the scaling
          p = -ldexpf((float)(M + (uint32_t)S), e - 150);
        } else {
          if (++n_replayed > replay_budget) {
            // more chunks to replay than the serial kernel would be slower for: hand the run over, record untouched
            gave_up = true;
            break;
          }
          // replay the chunk: gather (lane = update), then the 21 chains over it
          const XlChunk& X = ch[b];
          const uint32_t off = b * kXlChunk + (uint32_t)lane;
          const bool in = off < R.len;
          const uint64_t key = pairs[R.start + (in ? off : 0u)];
          const uint32_t rp = (uint32_t)key & F.point_mask;
          const RayDesc d = rays[ray_index(F, rp)];
          const uint32_t kind = in ? ((d.info >> 8) & 3u) : 0u, lab = d.info & 0xffu;
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
          const unsigned long long active = (unsigned long long)X.active_lo | ((unsigned long long)X.active_hi << 32);
          const int cnt = (int)X.cnt;
          for (int kk = 0; kk < cnt; ++kk)
            if ((active >> kk) & 1ull) p += s_inc[kk][cls];
          KS_WAVE_LDS_ORDER();
        }
      }
    }
    if (lane == 0) {
      atomicAdd(&hdr->tot_chunks, (unsigned long long)n_ch);
      atomicAdd(&hdr->tot_replayed, (unsigned long long)n_replayed);
    }
    if (gave_up) {
      if (lane == 0) {
        xlong_list[atomicAdd(&hdr->n_fallback, 1u)] = R.start;
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
    uint32_t color = rec[2];
    if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = label_lut[best];
    else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) color = rainbow_color_map((double)(float)exp((double)m));
    if (lane == 0) {
      // (distance and weight stay where they are: the fixed point)
      rec[2] = color;
      rec[3] = (uint32_t)best;
      rec[25] = 1u;  // updated since the last voxel-level host sync
    }
  }
}

}  // namespace ksk
