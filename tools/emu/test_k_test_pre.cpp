// tools/emu/test_k_test_pre.cpp — functional check of the early-out kernels WITHOUT a GPU (tools/emu/README.md):
// runs ksk::k_test and ksk::k_test_pre (kimera_semantics_amd/csrc/ks_k_march.h, compiled for the host against the
// stand-in <hip/hip_runtime.h> of this directory) over a synthetic frame, phase by phase, and compares three things
// after every phase — the per-ray update counts and the newest entry of every slot of the shared early-out set:
//   (1) every phase through k_test                       (the kernel the GPU tests pin against the oracle)
//   (2) k_prewalk, then the leading phases of one sub-run per chain through k_test_pre, the others through k_test
//       (what ks_hip.hip launches with KS_TEST_PRE=1)
//   (3) a plain serial restatement of the ordered-phase schedule written here
// Usage: test_k_test_pre [n_points] [max_collisions] [seed] [voxel_size_m]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "ks_k_march.h"

namespace ksk {
thread_local unsigned long long s_test[64 * 1024 / 8 * 4];   // the kernels' dynamic LDS
thread_local unsigned long long s_bt[1 << 16];
}  // namespace ksk

using namespace ksk;

// shared-set atomics of the k_test_pre side (obs_atomic_max): how many, and how many on the most contended address
#include <unordered_map>
static std::unordered_map<const void*, uint32_t> g_atomics;
static void count_atomic(const void* p) { ++g_atomics[p]; }

static std::vector<uint32_t> phase_bounds(uint32_t n_gen, int growth) {   // = ks_hip.hip
  std::vector<uint32_t> b{0};
  for (;;) {
    const uint64_t inc = std::max<uint64_t>(1, (uint64_t)b.back() * (uint64_t)(growth - 16) / 16);
    if (b.back() + inc >= n_gen) break;
    b.push_back((uint32_t)(b.back() + inc));
  }
  return b;
}

struct Frame {
  FrameParams F{};
  std::vector<RayDesc> rays;
  std::vector<uint8_t> live;
  uint32_t steps_cap = 0;
};

static Frame make_frame(uint32_t n, int lim, uint32_t seed, uint32_t tag, uint32_t tag_lo, uint64_t offset, float voxel) {
  Frame fr;
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  FrameParams& F = fr.F;
  F.T.w = 1.f;
  F.T.v = {0.f, 0.f, 0.f};
  F.T.t = {0.13f, -0.21f, 0.37f};
  F.voxel_size_inv = 1.0f / voxel;
  F.min_ray = 0.1f;
  F.max_ray = 5.0f;
  F.trunc = 3.0f * voxel;
  F.observed_offset = offset;
  F.obs_tag = tag;
  F.obs_tag_lo = tag_lo;
  F.max_collisions = lim;
  F.n = n;
  F.per_group = n / 1024u;
  F.carving = 1;
  F.method = KS_METHOD_FAST;
  F.sorted_order = 0;
  F.early_out = 1;
  fr.rays.resize(n);
  fr.live.resize(n);
  // a depth image of a room corner seen from the sensor: neighbouring pixels -> nearly the same rays (the early-out has
  // something to do), plus clearing rays, axis-parallel rays (the serial caster path) and a few degenerate ones
  const uint32_t w = 160;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t px = i % w, py = i / w;
    const float ax = ((float)px - 80.f) / 110.f, ay = ((float)py - 60.f) / 110.f;
    float depth = 2.2f + 1.5f * sinf(0.02f * px) + 0.8f * cosf(0.05f * py) + 0.02f * U(rng);
    const bool clearing = U(rng) < 0.05f;
    if (clearing) depth = 4.7f;   // (a clearing ray: cast to max_ray at most)
    f3 dir = {ax, ay, 1.f};
    const float inv = 1.0f / sqrtf(ax * ax + ay * ay + 1.f);
    f3 pt = {F.T.t.x + dir.x * inv * depth, F.T.t.y + dir.y * inv * depth, F.T.t.z + dir.z * inv * depth};
    const float r = U(rng);
    if (r < 0.01f) pt = {F.T.t.x, F.T.t.y, F.T.t.z + depth};            // axis-parallel
    else if (r < 0.015f) pt = {F.T.t.x + depth, F.T.t.y, F.T.t.z};
    else if (r < 0.017f) pt = F.T.t;                                       // zero-length ray
    RayDesc d{};
    d.px = pt.x; d.py = pt.y; d.pz = pt.z;
    d.weight = 1.f;
    d.info = 1u | (1u << 8) | (clearing ? (1u << 10) : 0u);
    fr.rays[i] = d;
    fr.live[i] = U(rng) < 0.7f ? 1 : 0;   // indexed by integration position
  }
  const double max_len = 5.0 + 2.0 * (double)F.trunc;
  const size_t steps_max = (size_t)std::ceil(1.7321 * max_len * (double)F.voxel_size_inv) + 8;
  fr.steps_cap = (uint32_t)((steps_max + 3) & ~(size_t)3);
  return fr;
}

struct State {
  std::vector<unsigned long long> observed;   // [2^20][2]
  std::vector<uint32_t> cnt;
  std::vector<uint32_t> pre_hash;
  std::vector<int> pre_steps;
  Counters C{};
  FrameParams F;
};

static BatchView view_of(const Frame& fr, State& st) {
  st.F = fr.F;
  st.F.observed = (uint64_t*)st.observed.data();
  BatchView V{};
  SlotView& sv = V.s[0];
  sv.F = &st.F;
  sv.live = fr.live.data();
  sv.rays = fr.rays.data();
  sv.cnt = st.cnt.data();
  sv.C = &st.C;
  sv.pre_hash = st.pre_hash.data();
  sv.pre_steps = st.pre_steps.data();
  return V;
}

static const bool g_by_generation = getenv("EMU_SUB_RUN_GENERATIONS") != nullptr;
static const bool g_no_overlap = getenv("EMU_TEST_NO_OVERLAP") != nullptr;
static void run_prewalk(const Frame& fr, State& st, uint32_t G, uint32_t Gpad, uint32_t cap) {
  st.pre_hash.assign((size_t)kChains * Gpad * cap, 0xdeadbeefu);
  st.pre_steps.assign((size_t)kChains * Gpad, -7);
  BatchView V = view_of(fr, st);
  emu::launch(dim3(kChains * (Gpad / kSubRun), 1), dim3(64), [&] { k_prewalk(V, G, Gpad, cap); });
}

static void run_phase_kernel(const Frame& fr, State& st, uint32_t g0, uint32_t g1, bool pre, uint32_t Gpad, uint32_t cap) {
  BatchView V = view_of(fr, st);
  const uint32_t n_sub = (g1 - g0 + kSubRun - 1) / kSubRun;
  if (pre) {
    // = the choice of ks_hip.hip (KS_TEST_PRE=1); EMU_PRE_NO_DEDUP=1: one shared-set mark per visited voxel
    static const bool no_dedup = getenv("EMU_PRE_NO_DEDUP") != nullptr;
#define LAUNCH_PRE(WW, DD) emu::launch(dim3(kChains, 1), dim3(64), [&] { k_test_pre<WW, DD>(V, g0, g1, Gpad, cap); })
    if (g1 - g0 <= 4) { if (no_dedup) LAUNCH_PRE(8, false); else LAUNCH_PRE(8, true); }
    else if (g1 - g0 <= 8) { if (no_dedup) LAUNCH_PRE(16, false); else LAUNCH_PRE(16, true); }
    else { if (no_dedup) LAUNCH_PRE(32, false); else LAUNCH_PRE(32, true); }
#undef LAUNCH_PRE
  } else {
    // one wavefront per block (the kernel derives its (chain, sub-run) from blockIdx and blockDim)
    emu::launch(dim3(kChains * n_sub, 1), dim3(64), [&] { if (g_no_overlap) k_test<false>(V, g0, g1, fr.steps_cap, g_by_generation ? 1u : 0u, kSubRun); else k_test<true>(V, g0, g1, fr.steps_cap, g_by_generation ? 1u : 0u, kSubRun); });
  }
}

// (3) the schedule, serially
static void run_phase_serial(const Frame& fr, State& st, uint32_t g0, uint32_t g1) {
  const FrameParams& F = fr.F;
  const uint32_t n_gen = (F.n + kChains - 1u) / kChains;
  if (g1 > n_gen) g1 = n_gen;
  const std::vector<unsigned long long> snap = st.observed;   // the set as it stood when the phase began
  std::vector<unsigned long long> priv(kPrivSlots);
  std::vector<unsigned long long> keys;
  for (uint32_t chain = 0; chain < kChains; ++chain) {
    uint32_t live_seen = 0;
    {
      for (uint32_t g = g0; g < g1; ++g) {
        const uint64_t p = (uint64_t)g * kChains + chain;
        // a sub-run = 16 live rays of the chain (EMU_SUB_RUN_GENERATIONS=1: 16 generations, the schedule until round 3)
        if (g_by_generation ? (g - g0) % kSubRun == 0 : (p < F.n && fr.live[p] && live_seen % kSubRun == 0)) std::fill(priv.begin(), priv.end(), 0ull);
        if (p >= F.n || !fr.live[p]) continue;
        ++live_seen;
        const RayDesc d = fr.rays[ray_index(F, (uint32_t)p)];
        Dda dda{};
        dda.setup(F.T.t, {d.px, d.py, d.pz}, ((d.info >> 10) & 1u) != 0, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc, false);
        keys.clear();
        int c = 0, stop = -1;
        for (int s = 0; s <= dda.steps; ++s) {
          const uint32_t h = index_hash(dda.cx, dda.cy, dda.cz);
          const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
          keys.push_back(((unsigned long long)slot << 32) | h);
          bool hit = false;
          if (!priv_lookup(priv.data(), slot, h, hit)) hit = obs_match(snap[2u * slot], h, F.obs_tag_lo, F.obs_tag);
          c = hit ? c + 1 : 0;
          if (c > F.max_collisions) {
            stop = s;
            break;
          }
          if (s < dda.steps) dda.advance();
        }
        const uint32_t updates = stop >= 0 ? (uint32_t)stop : (uint32_t)dda.steps + 1u;
        const uint32_t visited = stop >= 0 ? updates + 1u : updates;
        for (uint32_t s = 0; s < visited; ++s) {
          const uint32_t slot = (uint32_t)(keys[s] >> 32), h = (uint32_t)keys[s];
          unsigned long long& pe = priv[slot & (kPrivSlots - 1u)];
          pe = std::max(pe, priv_key(g, s, slot, h));
          unsigned long long& e = st.observed[2u * slot];
          e = std::max(e, (unsigned long long)obs_entry(F.obs_tag, (uint32_t)p, h));
          ++g_atomics[&e];   // one atomicMax per visited voxel: what k_test issues
        }
        st.cnt[p] = updates | (stop >= 0 ? kCntBroke : 0u);
      }
    }
  }
}

static bool same(const State& a, const State& b, const char* what, uint32_t g0, uint32_t g1) {
  size_t bad_cnt = 0, bad_obs = 0, first = ~(size_t)0;
  for (size_t i = 0; i < a.cnt.size(); ++i)
    if (a.cnt[i] != b.cnt[i]) {
      if (!bad_cnt) first = i;
      ++bad_cnt;
    }
  for (size_t s = 0; s < a.observed.size(); s += 2) bad_obs += a.observed[s] != b.observed[s];
  if (bad_cnt || bad_obs) {
    printf("  MISMATCH %s in phase [%u,%u): %zu counts (first position %zu: %08x vs %08x), %zu slots\n", what, g0, g1, bad_cnt, first,
           first != ~(size_t)0 ? a.cnt[first] : 0u, first != ~(size_t)0 ? b.cnt[first] : 0u, bad_obs);
    return false;
  }
  return true;
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 19956u;
  const int lim = argc > 2 ? atoi(argv[2]) : 2;
  const uint32_t seed = argc > 3 ? (uint32_t)atoi(argv[3]) : 1u;
  const float voxel = argc > 4 ? (float)atof(argv[4]) : 0.05f;
  bool ok = true;
  State A, B, S;
  for (State* s : {&A, &B, &S}) {
    s->observed.assign(2u << kSetBits, 0ull);
    s->cnt.assign(n, 0u);
  }
  // two frames of one offset generation: the second one also meets marks of the first (tags tag_lo .. tag)
  for (uint32_t frame = 0; frame < 2 && ok; ++frame) {
    const Frame fr = make_frame(n, lim, seed + 17u * frame, 5u + frame, 5u, 0x9e3779b97f4a7c15ull, voxel);
    if (frame == 0) printf("n %u, voxel %.3f m, steps_cap %u, k_test_pre cap %u, LDS %u B\n", n, voxel, fr.steps_cap, test_pre_cap(fr.steps_cap), test_pre_lds_bytes(test_pre_cap(fr.steps_cap)));
    A.pre_hash.assign(4, 0u); A.pre_steps.assign(4, 0);
    for (State* s : {&A, &B, &S}) std::fill(s->cnt.begin(), s->cnt.end(), 0u);
    const uint32_t n_gen = (n + kChains - 1) / kChains;
    const std::vector<uint32_t> PB = phase_bounds(n_gen, 32);
    // = pre_plan() of ks_hip.hip: the leading phases of at most one sub-run per chain
    uint32_t G = 0;
    for (size_t j = 0; j < PB.size(); ++j) {
      const uint32_t g0 = PB[j], g1 = j + 1 < PB.size() ? PB[j + 1] : n_gen;
      if (g1 - g0 > kSubRun) break;
      G = g1;
    }
    const uint32_t Gpad = (G + kSubRun - 1) / kSubRun * kSubRun, cap = test_pre_cap(fr.steps_cap);
    if (test_pre_lds_bytes(cap) > 64 * 1024 || (cap + 63) / 64 * 16 > kPreMaxChunks) { fprintf(stderr, "k_test_pre does not apply (LDS)\n"); return 2; }
    run_prewalk(fr, B, G, Gpad, cap);
    unsigned long long updates = 0, broke = 0, rays = 0;
    for (size_t j = 0; j < PB.size() && ok; ++j) {
      const uint32_t g0 = PB[j], g1 = j + 1 < PB.size() ? PB[j + 1] : n_gen;
      const bool pre = g1 <= G;
      run_phase_kernel(fr, A, g0, g1, false, Gpad, cap);
      g_atomics.clear();
      emu::global_atomic_hook = pre ? count_atomic : nullptr;
      run_phase_kernel(fr, B, g0, g1, pre, Gpad, cap);
      emu::global_atomic_hook = nullptr;
      if (pre) {
        size_t total = 0, worst = 0;
        for (const auto& kv : g_atomics) { total += kv.second; worst = std::max<size_t>(worst, kv.second); }
        printf("  k_test_pre: %zu shared-set atomics on %zu addresses, %zu on the most contended one\n", total, g_atomics.size(), worst);
      }
      g_atomics.clear();
      run_phase_serial(fr, S, g0, g1);
      {
        size_t total = 0, worst = 0;
        for (const auto& kv : g_atomics) { total += kv.second; worst = std::max<size_t>(worst, kv.second); }
        printf("  one mark per visited voxel (k_test): %zu shared-set atomics on %zu addresses, %zu on the most contended one\n", total,
               g_atomics.size(), worst);
      }
      const bool ok1 = same(A, S, "k_test vs serial restatement", g0, g1);
      const bool ok2 = same(B, S, pre ? "k_test_pre vs serial restatement" : "k_test (after k_test_pre phases) vs serial restatement", g0, g1);
      ok = ok1 && ok2;
      printf("frame %u phase [%u,%u) %s: %s\n", frame, g0, g1, pre ? "k_test_pre" : "k_test    ", ok ? "identical" : "DIFFERENT");
      fflush(stdout);
    }
    for (uint32_t p = 0; p < n; ++p)
      if (fr.live[p]) {
        ++rays;
        updates += S.cnt[p] & ~kCntBroke;
        broke += (S.cnt[p] & kCntBroke) != 0;
      }
    printf("frame %u: %llu rays, %llu updates, %llu rays stopped early (max_collisions %d)\n", frame, rays, updates, broke, lim);
  }
  printf(ok ? "OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
