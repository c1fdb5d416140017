// ks_k_march.h — stage B kernels: which voxels each ray updates (ordered-phase early-out), where its
// updates go in the pair list (scan), the pair emission with tile allocation, the counter snapshot,
// and the initialisation of freshly allocated tiles.
//
// Stage B never walks a ray voxel by voxel with a dependent memory access per step.  A ray (or bundle)
// is handled by a GROUP OF LANES of one wavefront: every lane runs the ray caster forward to its own
// step (the DDA state is a handful of registers; its float accumulation is replayed exactly), so one
// memory round trip serves 16 or 64 consecutive voxels of the ray.
//   k_test   (fast, early-out on) decides how far every ray gets and enters its marks:  ORDERED-PHASE schedule, below
//   k_scan_local + k_emit: exclusive scan of the per-ray update counts in integration order, then
//            every ray writes its (voxel, ray) keys at its own offset — the pair list comes out in
//            integration order, so the sort that follows only has to group by voxel (stable)
// [K:src/semantic_tsdf_integrator_fast.cpp:94-141], [K:src/semantic_tsdf_integrator_merged.cpp:288-328]
#pragma once
#include "ks_types.h"

namespace ksk {

// ------------------------------------------------------------------------------------------
// Shared approximate set of the fast integrator's early-out (voxel_observed_approx_set_,
// [K:include/kimera_semantics/semantic_tsdf_integrator_fast.h:102-130]): 2^20 slots, slot =
// (hash + offset) & mask as in the reference; an entry holds the LAST mark in integration order:
//   [63:54] frame tag   [53:32] integration position + 1   [31:0] the voxel hash
// so that marks can be entered with one atomicMax each, in any order, with the reference's
// "last writer wins" outcome.  Entries of an older offset generation can never match in the
// reference (the slot depends on the offset); here they are recognised by their tag.
// 0 = never written since the last full reset; position field 0 with a non-zero hash field = poison /
// retired entry.  Neither matches anything.
// A slot is TWO such entries, 16 bytes: {newest, older}.  k_test enters a ray's marks itself, with atomicMax on
// `newest`, while other wavefronts of the same launch are still testing — and a test must see the set AS IT STOOD
// WHEN THE PHASE BEGAN.  So a test that finds a current-phase mark in `newest` (same frame tag, position inside the
// phase) reads `older` instead, which holds the newest mark from BEFORE the phase: every test that loads an older-
// phase `newest` saves it there first (atomicMax; idempotent), and a wavefront waits for its saves to complete
// (s_waitcnt vmcnt(0)) before it enters marks of its own.  Both entries share one 16-byte line: one load per test.  (The reference's zero-initialised slots "contain" hash 0 —
// the voxel whose hash is 0 looks already observed until something overwrites its slot; the ordered-
// phase schedule does not reproduce that one-voxel artefact, which lets every frame in flight use a
// table of its own.)
// ------------------------------------------------------------------------------------------
constexpr uint64_t kObsPoison = 0x00000000ffffffffull;   // ApproxHashSet's SIZE_MAX at slot[offset = 0]
constexpr uint64_t kObsRetired = 0x00000000fffffffeull;  // written by k_obs_retag when the frame tag wraps
constexpr uint32_t kObsMaxTag = 1023;
constexpr uint32_t kObsMaxPoints = (1u << 22) - 2;        // position + 1 must fit 22 bits

__device__ __forceinline__ uint64_t obs_entry(uint32_t tag, uint32_t pos, uint32_t hash) {
  return ((uint64_t)tag << 54) | ((uint64_t)(pos + 1u) << 32) | (uint64_t)hash;
}
// does the slot content match hash h?  (tag_lo..tag = the frames of the current offset generation)
__device__ __forceinline__ bool obs_match(uint64_t e, uint32_t h, uint32_t tag_lo, uint32_t tag) {
  const uint32_t posf = (uint32_t)(e >> 32) & 0x3fffffu, t = (uint32_t)(e >> 54);
  return posf != 0u && t >= tag_lo && t <= tag && (uint32_t)e == h;
}

__global__ void __launch_bounds__(256) k_obs_retag(uint64_t* __restrict__ set) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (2u << kSetBits)) {  // both entries of every slot
    const uint64_t e = set[i];
    if (e != 0ull) set[i] = kObsRetired;
  }
}

// A group of LPR lanes of one wavefront cooperating on a ray.
template <int LPR>
struct LaneGroup {
  static_assert(LPR == 16 || LPR == 64, "lanes per ray");
  uint32_t l;      // lane within the group
  uint32_t shift;  // first wave lane of the group
  __device__ __forceinline__ LaneGroup() {
    const uint32_t lane = lane_id();
    l = lane & (LPR - 1);
    shift = lane & ~(uint32_t)(LPR - 1);
  }
  __device__ __forceinline__ uint64_t bits(unsigned long long ballot) const {
    return LPR == 64 ? ballot : ((ballot >> shift) & ((1ull << (LPR & 63)) - 1ull));
  }
  template <typename T>
  __device__ __forceinline__ T from(T x, uint32_t src) const { return __shfl(x, (int)(shift + src)); }
};

// Runs the caster from the round's first step to this lane's step (l advances).  The loop bound is
// uniform; the float accumulation is exactly the serial caster's.
template <int LPR>
__device__ __forceinline__ Dda dda_at_lane(const Dda& round_start, uint32_t l) {
  Dda d = round_start;
#pragma unroll 4
  for (uint32_t i = 0; i + 1 < (uint32_t)LPR; ++i) d.advance(i < l);
  return d;
}
// state of the next round's first step: the last lane's state advanced once more
template <int LPR>
__device__ __forceinline__ void dda_next_round(const LaneGroup<LPR>& G, Dda& round_start, Dda mine) {
  mine.advance();
  round_start.cx = G.from(mine.cx, LPR - 1);
  round_start.cy = G.from(mine.cy, LPR - 1);
  round_start.cz = G.from(mine.cz, LPR - 1);
  round_start.tx = G.from(mine.tx, LPR - 1);
  round_start.ty = G.from(mine.ty, LPR - 1);
  round_start.tz = G.from(mine.tz, LPR - 1);
}

// ------------------------------------------------------------------------------------------
// EXACT PARALLEL RAY CASTER: 64 consecutive steps of ONE ray by one wavefront.  The caster's state is three
// independent float accumulations (t_to_next[a] += t_step[a]); which axis steps next is a 3-way merge of
// the three increasing sequences of crossing times, ties going to the lower axis (Eigen's minCoeff = first
// strict minimum).  Lanes 0..2 replay the three accumulations exactly (65 dependent adds instead of 64
// full caster steps), every lane then ranks crossing number `lane` of each axis in the merged order by
// binary search (strict / non-strict comparison = the tie rule), and the crossing of rank r yields the voxel
// of step r.  Valid while every ray component is non-zero and all times are finite (else the three
// sequences are not increasing: NaN / inf cases of an axis-parallel ray go through the serial caster).
// Checked against the serial caster on 3.5e5 random rays incl. 1.2e7 exact ties (tools/pdda_model.cpp).
// ------------------------------------------------------------------------------------------
constexpr int kES = 66;  // floats per axis of the LDS scratch (65 used)
__device__ __forceinline__ bool dda_parallel_ok(const Dda& d) {
  return d.sx != 0 && d.sy != 0 && d.sz != 0 && isfinite(d.tx) && isfinite(d.ty) && isfinite(d.tz) && isfinite(d.dx) &&
         isfinite(d.dy) && isfinite(d.dz);
}
template <bool LE>
__device__ __forceinline__ int sorted_count(const float* e, float v) {  // #{i < 64 : e[i] < v}  (LE: <=)
  int base = 0;
#pragma unroll
  for (int h = 32; h >= 1; h >>= 1) {
    const float x = e[base + h - 1];
    base += (LE ? (x <= v) : (x < v)) ? h : 0;
  }
  const float x = e[base];
  return base + ((LE ? (x <= v) : (x < v)) ? 1 : 0);
}
// st: wave-uniform state at the round's first step (advanced by 64 steps on return); E: 3 * kES floats of LDS
// private to the wavefront; visit(r, vx, vy, vz) is called once for every step r in [0, 64), by the lane
// that owns the crossing which follows it.
template <typename Visit>
__device__ __forceinline__ void dda_round64(Dda& st, float* E, uint32_t lane, Visit&& visit) {
  float tx0 = st.tx, ty0 = st.ty, tz0 = st.tz, dx0 = st.dx, dy0 = st.dy, dz0 = st.dz;
  KS_VALUE_BARRIER(tx0); KS_VALUE_BARRIER(ty0); KS_VALUE_BARRIER(tz0);
  KS_VALUE_BARRIER(dx0); KS_VALUE_BARRIER(dy0); KS_VALUE_BARRIER(dz0);
  if (lane < 3) {
    float t = lane == 0 ? tx0 : lane == 1 ? ty0 : tz0;
    const float d = lane == 0 ? dx0 : lane == 1 ? dy0 : dz0;
    float* e = E + lane * kES;
    for (int i = 0; i < 65; ++i) {
      e[i] = t;
      t = t + d;
    }
  }
  KS_WAVE_LDS_ORDER();
  const float *Ex = E, *Ey = E + kES, *Ez = E + 2 * kES;
  const float ex = Ex[lane], ey = Ey[lane], ez = Ez[lane];
  const int cyx = sorted_count<false>(Ey, ex), czx = sorted_count<false>(Ez, ex);
  const int cxy = sorted_count<true>(Ex, ey), czy = sorted_count<false>(Ez, ey);
  const int cxz = sorted_count<true>(Ex, ez), cyz = sorted_count<true>(Ey, ez);
  const int rx = (int)lane + cyx + czx, ry = (int)lane + cxy + czy, rz = (int)lane + cxz + cyz;
  if (rx < 64) visit((uint32_t)rx, st.cx + st.sx * (int)lane, st.cy + st.sy * cyx, st.cz + st.sz * czx);
  if (ry < 64) visit((uint32_t)ry, st.cx + st.sx * cxy, st.cy + st.sy * (int)lane, st.cz + st.sz * czy);
  if (rz < 64) visit((uint32_t)rz, st.cx + st.sx * cxz, st.cy + st.sy * cyz, st.cz + st.sz * (int)lane);
  const int nx = (int)__popcll(__ballot(rx < 64)), ny = (int)__popcll(__ballot(ry < 64)), nz = (int)__popcll(__ballot(rz < 64));
  const float ntx = Ex[nx], nty = Ey[ny], ntz = Ez[nz];
  st.cx += st.sx * nx;
  st.cy += st.sy * ny;
  st.cz += st.sz * nz;
  st.tx = ntx;
  st.ty = nty;
  st.tz = ntz;
}
__device__ __forceinline__ Dda dda_bcast(const Dda& d, int src) {
  Dda r;
  r.cx = __shfl(d.cx, src); r.cy = __shfl(d.cy, src); r.cz = __shfl(d.cz, src);
  r.sx = __shfl(d.sx, src); r.sy = __shfl(d.sy, src); r.sz = __shfl(d.sz, src);
  r.tx = __shfl(d.tx, src); r.ty = __shfl(d.ty, src); r.tz = __shfl(d.tz, src);
  r.dx = __shfl(d.dx, src); r.dy = __shfl(d.dy, src); r.dz = __shfl(d.dz, src);
  r.steps = __shfl(d.steps, src);
  r.in_range = true;
  return r;
}

// What stage B reads and writes of ONE frame (slot), and a batch of up to kBatchMax frames: every stage-B kernel is
// launched once for the whole batch (blockIdx.y = frame of the batch).  The chains of small dependent launches that
// decide the early-out are latency bound — a launch costs little more for four frames than for one (eight cost 1.9x four:
// measured) — and the hardware runs at most a couple of such chains side by side when they sit on different streams.
constexpr int kBatchMax = 8;
struct SlotView {
  const FrameParams* F;          // the frame's parameters in device memory
  const uint8_t* live;           // fast: position holds a ray that survived the start-voxel dedup
  const RayDesc* rays;
  uint32_t* cnt;                 // updates per integration position (| kCntBroke)
  uint32_t* lp;                  // block-local exclusive prefix of cnt
  unsigned long long* bt;        // block totals of that scan
  const uint32_t* ray_list;
  uint64_t* pairs;               // (voxel, position) keys, integration order
  unsigned long long pairs_cap;
  Counters* C;
  uint32_t* host_snap;           // pinned snapshot (k_publish)
  const uint32_t* eo_stats;      // exact early-out, event-driven: {X marks + 1, failure bits, rounds} of the frame (else nullptr)
};
struct BatchView {
  SlotView s[kBatchMax];
};

constexpr uint32_t kOrderStep = 1024;   // vxb::MixedThreadSafeIndex::step_size_
constexpr uint32_t kPrivSlots = 1024;   // chain-private direct-mapped set (8 KiB of LDS per chain)
constexpr uint32_t kCntBroke = 1u << 31;  // cnt[] flag: the ray stopped on a voxel it visited but did not update

// ------------------------------------------------------------------------------------------
// k_test — ORDERED-PHASE early-out (the CPU checker under oracle/ restates it as integrate_fast_phased):
// integration position s -> chain s % chains, generation s / chains (chains = the groups of the "mixed" order,
// FrameParams::chains: N / 1024 in the upstream form, 1024 in the other one); one launch per phase of generations
// [g0, g1).  Inside a phase a chain's LIVE rays (those that survived the start-voxel dedup: nearly all of the
// first generations, one in five to one in twenty of the late ones) are cut, in generation order, into SUB-RUNS
// of 16; ONE WAVEFRONT owns a (chain, sub-run) and resolves its rays in generation order.  The launch covers the worst
// case (every ray live: one wavefront per 16 generations); a wavefront finds its rays by ranking the chain's live
// flags of the phase (64 generations per ballot) and ends at once if the chain has fewer than 16 * sub + 1 of them.
// A ray tests its voxels against (a) the
// marks previous rays of its sub-run made (8 KiB of LDS, newest (generation, step) wins a slot) and
// (b) the shared set as it stood when the phase began (read-only during the launch).
//   A  lanes 0..15, one ray each: descriptor, caster set-up, the first 16 voxels walked serially (no lane
//      replays another lane's steps); (slot | hash) of every voxel goes to LDS
//   B  all 64 lanes: the shared-set entries of those <= 256 voxels, four per lane, in flight together
//   C  the rays one after the other, 16 lanes per ray, LDS only: private-set lookup, the reference's
//      consecutive-collision rule on the 16-bit hit mask, the ray's marks.  A ray not decided within 16
//      voxels (the first through its corridor) is cast on 64 voxels at a time by the whole wavefront (exact parallel
//      caster) and tested by all 64 lanes; OVERLAP: the next 64 voxels are cast while the shared-set entries of
//      the current 64 are in flight (KS_TEST_OVERLAP=0 = one after the other, the code measured until round 3).
//   Every ray's marks (all visited voxels: the highest (position, hash) stays in a slot = the reference's last writer
//   in serial order) enter the shared set as soon as the ray is decided, by the same wavefront, from the keys it holds
//   in LDS — there is no second walk and no second launch per phase.
// Output: cnt[s] = number of voxels the ray updates (| kCntBroke).
// The reference's loop: [K:src/semantic_tsdf_integrator_fast.cpp:110-122].
// ------------------------------------------------------------------------------------------
// The shared set is read and written through pointers that SAY global memory: the table's address comes out of a struct in
// memory (FrameParams), so the compiler would otherwise emit FLAT operations, and a flat load counts as an LDS operation
// too — every wait for the LDS (s_waitcnt lgkmcnt) would then also wait for the set's entries in flight.
typedef unsigned long long obs_u64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned long long obs_global_u64;
typedef __attribute__((address_space(1))) obs_u64x2 obs_global_u64x2;
__device__ __forceinline__ void obs_atomic_max(obs_global_u64* p, unsigned long long v) {   // = atomicMax, result unused
  (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ ulonglong2 obs_load2(const obs_global_u64* observed, uint32_t slot) {   // {newest, older} of a slot
  const obs_u64x2 e = ((const obs_global_u64x2*)observed)[slot];
  return make_ulonglong2(e.x, e.y);
}
constexpr int kTestThreads = 256;    // 4 wavefronts per block
constexpr uint32_t kSubRun = 16;     // rays per (chain, sub-run) wavefront
#ifdef KS_STATS
__device__ unsigned long long g_test_stats[16];  // diagnostics build only (tools/test_stats.py)
#define KS_STAT_ADD(i, v) do { if (lane_id() == 0) atomicAdd(&g_test_stats[i], (unsigned long long)(v)); } while (0)
#define KS_STAT_MAX(i, v) do { if (lane_id() == 0) atomicMax(&g_test_stats[i], (unsigned long long)(v)); } while (0)
#else
#define KS_STAT_ADD(i, v)
#define KS_STAT_MAX(i, v)
#endif

// the reference's consecutive-collision rule over one round: returns the index of the step the ray
// breaks on (it is visited, not updated) or -1; c = running counter (in/out).  Wave-uniform operands.
__device__ __forceinline__ int early_out_stop(uint64_t hits, uint64_t valids, int lim, int& c) {
  // valids is a prefix of ones (steps past the ray's end are invalid); hits is a subset of it
  const int nv = valids == ~0ull ? 64 : (int)__ffsll((long long)~valids) - 1;
  if (lim >= 0 && lim <= 6 && c <= lim && nv + c <= 64) {
    // the ray breaks at the first step that completes a run of lim + 1 hits; the c hits carried in
    // from the previous round are prepended as ones
    const uint64_t hh = (c ? ((hits << c) | ((1ull << c) - 1ull)) : hits);
    uint64_t run = hh;
    for (int k = 1; k <= lim; ++k) run &= hh >> k;  // bit i set <=> bits i .. i+lim of hh all set
    if (run != 0ull) {
      const int stop = (int)__ffsll((long long)run) - 1 + lim - c;
      if (stop < nv) return stop;  // c is irrelevant after a break
    }
    // no break: the new counter is the length of the trailing run of hits among the nv valid steps
    if (nv == 0) return -1;
    const uint64_t top = nv + c == 64 ? hh : (hh | (~0ull << (nv + c)));
    const uint64_t inv = ~top;  // zeros of hh below bit nv + c
    const int last_zero = inv == 0ull ? -1 : 63 - (int)__clzll((long long)inv);
    c = nv + c - 1 - last_zero;
    return -1;
  }
  int stop = -1;
  for (int i = 0; i < nv; ++i) {
    c = ((hits >> i) & 1ull) ? c + 1 : 0;
    if (c > lim) {
      stop = i;
      break;
    }
  }
  return stop;
}
__device__ __forceinline__ unsigned long long priv_key(uint32_t gen, uint32_t s, uint32_t slot, uint32_t h) {
  return ((unsigned long long)gen << 52) | ((unsigned long long)(s < 1023u ? s : 1023u) << 42) |
         ((unsigned long long)(slot >> 10) << 32) | (unsigned long long)h;
}
// private-set lookup: does an entry for this slot exist, and does it hold hash h?
__device__ __forceinline__ bool priv_lookup(const unsigned long long* priv, uint32_t slot, uint32_t h, bool& hit) {
  const unsigned long long pe = priv[slot & (kPrivSlots - 1u)];
  if (pe != 0ull && (uint32_t)((pe >> 32) & 1023ull) == (slot >> 10)) {
    hit = (uint32_t)pe == h;
    return true;
  }
  return false;
}

// LDS per wavefront: private set | keys of the first 16 voxels of 16 rays | keys of one long ray | per-ray words
__host__ __device__ inline uint32_t test_lds_words64(uint32_t steps_cap) { return kPrivSlots + 256u + steps_cap + 16u + (3u * kES + 1u) / 2u; }

template <bool OVERLAP>
__global__ void __launch_bounds__(kTestThreads) k_test(BatchView V, uint32_t g0, uint32_t g1, uint32_t steps_cap) {
  // stage B kernels read the frame's parameters from device memory: the launch sequence of a frame slot is
  // then identical from frame to frame and is replayed as a captured graph
  const SlotView& sv = V.s[blockIdx.y];
  const uint8_t* __restrict__ live = sv.live;
  const RayDesc* __restrict__ rays = sv.rays;
  uint32_t* __restrict__ cnt = sv.cnt;
  const Counters* C = sv.C;
  const FrameParams F = *sv.F;  // a COPY: through the pointer every loop iteration would re-load the fields it uses (they may alias the stores)
  obs_global_u64* observed = (obs_global_u64*)(unsigned long long*)F.observed;   // [slot] = {newest, older}
  const uint32_t n_chains = F.chains;
  const uint32_t phase_pos0 = g0 * n_chains;  // marks at positions >= this one belong to the phase being run
  // slot content as it stood when the phase began (see the set's description above)
  bool my_save = false;  // this lane has issued a save since the wavefront last waited for its saves
  // (in two steps, so that a batch of entries can be looked at before the first save goes out: a wait for the next
  // entry of the batch would otherwise also wait for the save issued in between — memory operations complete in order)
  auto snapshot_look = [&](const ulonglong2 e, uint32_t h, bool& save) -> bool {
    unsigned long long content = e.x;
    const bool current = (uint32_t)(e.x >> 54) == F.obs_tag && ((uint32_t)(e.x >> 32) & 0x3fffffu) > phase_pos0;
    if (current) content = e.y;
    save = !current && e.x != 0ull && e.x != e.y;
    return obs_match(content, h, F.obs_tag_lo, F.obs_tag);
  };
  auto snapshot_save = [&](const ulonglong2 e, uint32_t slot) {
    obs_atomic_max(&observed[2u * slot + 1u], e.x);
    my_save = true;
  };
  auto snapshot_decide = [&](const ulonglong2 e, uint32_t slot, uint32_t h) -> bool {
    bool save;
    const bool hit = snapshot_look(e, h, save);
    if (save) snapshot_save(e, slot);
    return hit;
  };
  auto snapshot_hit = [&](uint32_t slot, uint32_t h) -> bool { return snapshot_decide(obs_load2(observed, slot), slot, h); };
  extern __shared__ unsigned long long s_test[];
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  unsigned long long* priv = s_test + (size_t)wave * test_lds_words64(steps_cap);
  unsigned long long* keys = priv + kPrivSlots;        // [16 rays][16 steps]  slot << 32 | hash
  unsigned long long* lkeys = keys + 256;              // [steps_cap]          the long ray's voxels from step 16 on
  uint32_t* rinfo = (uint32_t*)(lkeys + steps_cap);    // [16] steps of the ray | [16] shared-set hit mask
  float* escr = (float*)(rinfo + 32);                  // [3 * kES] scratch of the parallel caster
  // (looked at below: this load and the live flags are in flight together; an atomic load stays where it is written)
  const uint32_t frame_err = __hip_atomic_load(&C->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t w = blockIdx.x * (blockDim.x >> 6) + wave;
  const uint32_t chain = w % n_chains, sub = w / n_chains;
  {  // the launch covers the slot's capacity: the frame's last generation ends the last phase
    const uint32_t n_gen = (F.n + n_chains - 1u) / n_chains;
    if (g1 > n_gen) g1 = n_gen;
  }
  // ---- which rays: my_gen = generation of the ray lane l < 16 owns (~0u: none) ----
  uint32_t my_gen = ~0u;
  {
    // rank of every live ray of the chain within the phase; ranks [16 sub, 16 sub + 16) are this wavefront's.  Four
    // ballots' worth of flags are in flight at a time (a late phase of a 640x480 frame is 128 generations long).
    const uint32_t r_lo = sub * kSubRun;
    uint32_t seen = 0;
    for (uint32_t gb = g0; gb < g1 && seen < r_lo + kSubRun; gb += 256u) {
      uint8_t fl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (unconditional loads from a clamped address: a load under a branch would be waited for before the next is issued)
        const uint32_t g = gb + 64u * (uint32_t)q + lane;
        const bool in = g < g1 && (uint64_t)g * n_chains + chain < F.n;
        fl[q] = live[in ? (uint64_t)g * n_chains + chain : (uint64_t)0];
        if (!in) fl[q] = 0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned long long m = __ballot(fl[q] != 0);
        const uint32_t rank = seen + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (fl[q] != 0 && rank >= r_lo && rank < r_lo + kSubRun) rinfo[rank - r_lo] = gb + 64u * (uint32_t)q + lane;
        seen += (uint32_t)__popcll(m);
      }
    }
    if (seen <= r_lo) return;  // the chain has no 16 * sub + 1 live rays in this phase
    KS_WAVE_LDS_ORDER();
    if (lane < kSubRun && lane < seen - r_lo) my_gen = rinfo[lane];
    KS_WAVE_LDS_ORDER();       // (rinfo is reused below)
  }
  if (__ballot(my_gen != ~0u) == 0ull || (frame_err & (kErrLabel | kErrIndex))) return;
  const int lim = F.max_collisions;
#ifdef KS_STATS
  const unsigned long long t_begin = __builtin_readcyclecounter();
  uint32_t st_long = 0, st_rounds = 0;
#endif

  // ---- A: lane l < 16 owns a ray ----
  Dda dda{};
  int my_steps = -1;
  {
    const uint64_t p = (uint64_t)my_gen * n_chains + chain;
    const bool is_live = my_gen != ~0u;
    if (is_live) {
      const RayDesc d = rays[ray_index(F, (uint32_t)p)];
      dda.setup(F.T.t, {d.px, d.py, d.pz}, ((d.info >> 10) & 1u) != 0, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc,
                /*cast_from_origin=*/false);
      my_steps = dda.steps;
    }
    for (int s = 0; s < 16; ++s) {  // (uniform trip count; lanes without a ray idle through it)
      if (s <= my_steps) {
        const uint32_t h = index_hash(dda.cx, dda.cy, dda.cz);
        const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
        keys[lane * 16 + s] = ((unsigned long long)slot << 32) | h;
      }
      dda.advance(s < my_steps);  // the state after min(16, steps) advances stays in the owner lane
    }
    if (lane < 16) rinfo[lane] = (uint32_t)my_steps;
  }
  const uint32_t live_mask = (uint32_t)(__ballot(my_steps >= 0) & 0xffffull);
  if (live_mask == 0u) return;
  // ---- B: shared-set entries of the first 16 voxels of every ray: 4 rays per pass, all in flight ----
  {
    const uint32_t grp = lane >> 4, l = lane & 15u;
    bool hit[4];
    unsigned long long kk[4];
    ulonglong2 ee[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {  // the four loads first (the saves below may not be reordered with loads of the set)
      const uint32_t r = (uint32_t)b * 4u + grp;
      const bool on = ((live_mask >> r) & 1u) && (int)l <= (int)rinfo[r];
      kk[b] = on ? keys[r * 16 + l] : ~0ull;
      ee[b] = on ? obs_load2(observed, (uint32_t)(kk[b] >> 32)) : make_ulonglong2(0ull, 0ull);
    }
    // the private set is cleared while those loads are in flight (wave-private: no block barrier needed; first used in C)
    for (uint32_t i = lane; i < kPrivSlots; i += 64) priv[i] = 0ull;
    KS_WAIT_LOADS();
    bool sv4[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      sv4[b] = false;
      hit[b] = kk[b] != ~0ull && snapshot_look(ee[b], (uint32_t)kk[b], sv4[b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (kk[b] != ~0ull && sv4[b]) snapshot_save(ee[b], (uint32_t)(kk[b] >> 32));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const unsigned long long m = __ballot(hit[b]);
      if (lane < 4) rinfo[16 + b * 4 + lane] = (uint32_t)((m >> (16 * lane)) & 0xffffull);
    }
  }
  // ---- C: the rays in generation order against the private set ----
  for (uint32_t todo = live_mask; todo != 0u; todo &= todo - 1u) {
    KS_WAVE_LDS_ORDER();   // B's hit masks / the previous ray's marks in the private set
    const uint32_t j = (uint32_t)__ffs((int)todo) - 1u;
    const int steps_j = (int)rinfo[j];
    const uint32_t hs = rinfo[16 + j];
    const uint32_t gen_j = (uint32_t)__shfl((int)my_gen, (int)j), pos_j = gen_j * n_chains + chain;
    const bool valid = lane < 16 && (int)lane <= steps_j;
    unsigned long long k = 0ull;
    bool hit = false;
    if (valid) {
      k = keys[j * 16 + lane];
      if (!priv_lookup(priv, (uint32_t)(k >> 32), (uint32_t)k, hit)) hit = (hs >> lane) & 1u;
    }
    int c = 0;
    int stop = early_out_stop(__ballot(valid && hit), __ballot(valid), lim, c);
    uint32_t updates, visited;
    if (stop >= 0 || steps_j < 16) {
      updates = stop >= 0 ? (uint32_t)stop : (uint32_t)steps_j + 1u;
      visited = stop >= 0 ? updates + 1u : updates;
    } else {
      // long ray: its owner lane walks on, 64 voxels at a time; all lanes test them
#ifdef KS_STATS
      ++st_long;
#endif
      uint32_t s0 = 16;
      Dda ust = dda_bcast(dda, (int)j);          // the ray's state at step 16, wave-uniform
      const bool par = dda_parallel_ok(ust);
      // keys of the steps [first, first + 64) of the ray (as far as it goes) -> lkeys
      auto cast_round = [&](uint32_t first) {
        const uint32_t n_r = (uint32_t)steps_j + 1u - first < 64u ? (uint32_t)steps_j + 1u - first : 64u;
        if (par) {  // all 64 lanes: the exact parallel caster
          dda_round64(ust, escr, lane, [&](uint32_t r, int vx, int vy, int vz) {
            if (r < n_r) {
              const uint32_t h = index_hash(vx, vy, vz);
              const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
              lkeys[first - 16u + r] = ((unsigned long long)slot << 32) | h;
            }
          });
        } else if (lane == j) {  // axis-parallel ray (NaN / inf crossing times): its owner lane walks it
          for (uint32_t i = 0; i < n_r; ++i) {
            const uint32_t h = index_hash(dda.cx, dda.cy, dda.cz);
            const uint32_t slot = (uint32_t)(((uint64_t)h + F.observed_offset) & kSetMask);
            lkeys[first - 16u + i] = ((unsigned long long)slot << 32) | h;
            dda.advance(first + i < (uint32_t)steps_j);
          }
        }
      };
      if (OVERLAP) cast_round(s0);
      for (;;) {
#ifdef KS_STATS
        ++st_rounds;
#endif
        const uint32_t n_round = (uint32_t)steps_j + 1u - s0 < 64u ? (uint32_t)steps_j + 1u - s0 : 64u;
        if (!OVERLAP) cast_round(s0);
        KS_WAVE_LDS_ORDER();
        const bool v64 = lane < n_round;
        bool hit64 = false;
        if (!OVERLAP) {
          if (v64) {
            const unsigned long long k64 = lkeys[s0 - 16u + lane];
            if (!priv_lookup(priv, (uint32_t)(k64 >> 32), (uint32_t)k64, hit64)) hit64 = snapshot_hit((uint32_t)(k64 >> 32), (uint32_t)k64);
          }
        } else {
          // the shared-set entries of this round are requested, THEN the caster runs the next round (it touches LDS
          // only), then the entries are looked at: with one such wavefront per SIMD nothing else would fill the round
          // trip.  A round cast in vain (the ray stops in this one) costs nothing the wait would not have cost.
          unsigned long long k64 = 0ull;
          bool need = false;
          if (v64) {
            k64 = lkeys[s0 - 16u + lane];
            need = !priv_lookup(priv, (uint32_t)(k64 >> 32), (uint32_t)k64, hit64);
          }
          ulonglong2 e64 = make_ulonglong2(0ull, 0ull);
          if (need) e64 = obs_load2(observed, (uint32_t)(k64 >> 32));
          if (s0 + 64u <= (uint32_t)steps_j) cast_round(s0 + 64u);
          KS_WAIT_LOADS();   // (here, for all lanes: the save below then leaves without anything waiting for it)
          if (need) hit64 = snapshot_decide(e64, (uint32_t)(k64 >> 32), (uint32_t)k64);
        }
        stop = early_out_stop(__ballot(v64 && hit64), __ballot(v64), lim, c);
        if (stop >= 0) {
          updates = s0 + (uint32_t)stop;
          visited = updates + 1u;
          break;
        }
        if (s0 + 64u > (uint32_t)steps_j) {
          updates = (uint32_t)steps_j + 1u;
          visited = updates;
          break;
        }
        s0 += 64u;
      }
    }
    // the older-phase marks this wavefront's tests saved have been performed before any mark of its own goes out
    if (__ballot(my_save) != 0ull) {
      KS_WAIT_VMEM();
      my_save = false;
    }
    // marks of the voxels past the first 16 (kept in LDS: no second walk): the chain's private set, and the shared set
    for (uint32_t m0 = 16; m0 < visited; m0 += 64) {
      const uint32_t s = m0 + lane;
      if (s < visited) {
        const unsigned long long k64 = lkeys[s - 16u];
        atomicMax(&priv[(uint32_t)(k64 >> 32) & (kPrivSlots - 1u)], priv_key(gen_j, s, (uint32_t)(k64 >> 32), (uint32_t)k64));
        obs_atomic_max(&observed[2u * (uint32_t)(k64 >> 32)], (unsigned long long)obs_entry(F.obs_tag, pos_j, (uint32_t)k64));
      }
    }
    if (lane == 0) cnt[pos_j] = updates | (stop >= 0 ? kCntBroke : 0u);
    if (valid && lane < visited) {
      atomicMax(&priv[(uint32_t)(k >> 32) & (kPrivSlots - 1u)], priv_key(gen_j, lane, (uint32_t)(k >> 32), (uint32_t)k));
      obs_atomic_max(&observed[2u * (uint32_t)(k >> 32)], (unsigned long long)obs_entry(F.obs_tag, pos_j, (uint32_t)k));
    }
  }
#ifdef KS_STATS
  {
    const unsigned long long dt = __builtin_readcyclecounter() - t_begin;
    KS_STAT_ADD(0, dt);
    KS_STAT_MAX(1, dt);
    KS_STAT_ADD(2, 1);
    KS_STAT_ADD(4, __popc(live_mask));
    KS_STAT_ADD(5, st_long);
    KS_STAT_ADD(6, st_rounds);
    KS_STAT_MAX(7, st_rounds);
  }
#endif
}

constexpr uint32_t kLaneWalk = 32;
// Work split of the kernels that walk whole rays (k_emit_lane, k_eo_emit): RPW rays per wavefront, owned by its
// first RPW lanes.  RPW = 64 when most rays are a few voxels long (fast with the early-out), 8 when rays are
// long (merged bundles, 2 cm voxels): the long part of a ray costs whole-wavefront rounds, so fewer rays per
// wavefront means more wavefronts sharing that work.  Per wavefront, the tail beyond kLaneWalk voxels is
// either walked by the owner lanes (all at once, as long as the longest) or taken ray by ray with the
// parallel caster, whichever the step counts say is shorter.
__device__ __forceinline__ bool tails_by_wavefront(unsigned long long long_mask, uint32_t my_len) {
  if (long_mask == 0ull) return false;
  uint32_t rounds = 0, longest = 0;
  for (unsigned long long m = long_mask; m != 0ull; m &= m - 1ull) {
    const uint32_t v = __shfl(my_len, __ffsll((long long)m) - 1);
    rounds += (v - kLaneWalk + 63u) / 64u;
    longest = v > longest ? v : longest;
  }
  return rounds * 8u < longest - kLaneWalk;   // a 64-voxel round ~ 8 owner-lane steps
}
// entries of the per-position update counts: merged keeps the clearing bundles' counts at +n
__device__ __forceinline__ uint32_t scan_length(const FrameParams& F) { return (F.method == KS_METHOD_MERGED ? 2u : 1u) * F.n; }

// ------------------------------------------------------------------------------------------
// Scan of the per-ray update counts in integration order.  cnt[] is indexed by integration position
// (merged: first-point position of the bundle, clearing bundles offset by n: they integrate after all
// normal bundles, [K:src/semantic_tsdf_integrator_merged.cpp:126-144]); dead positions hold 0.
// k_scan_local: 4096 entries per block -> local exclusive prefix + block total; k_emit adds the
// block offsets (a few hundred values, scanned again by every block in LDS).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kScanBlock = 4096;
__global__ void __launch_bounds__(1024) k_scan_local(BatchView V) {
  __shared__ uint32_t s_wave[16];
  const SlotView& sv = V.s[blockIdx.y];
  const uint32_t* __restrict__ cnt = sv.cnt;
  uint32_t* __restrict__ lp = sv.lp;
  unsigned long long* __restrict__ bt = sv.bt;
  const uint32_t n_scan = scan_length(*sv.F);  // the grid covers the slot's capacity
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const uint32_t i0 = blockIdx.x * kScanBlock + threadIdx.x * 4u;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n_scan) ? (cnt[i0 + k] & ~kCntBroke) : 0u;
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
    if (i0 + k < n_scan) lp[i0 + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 1023) bt[blockIdx.x] = (unsigned long long)(wbase + x);
}

// ------------------------------------------------------------------------------------------
// k_emit — every live ray writes its (voxel, ray) keys at its offset of the scan; allocates voxel
// tiles on first touch (CAS + pool bump; waits for a slot another lane is publishing only after the
// wave has reconverged).  Key = [63:56] label | kind | clearing, [..] voxel id << seq_bits | ray sequence.
// The list is in integration order; total = Counters::n_pairs.
// ------------------------------------------------------------------------------------------
#define KS_SLOT_ARGS(V)                                       \
  const SlotView& sv = (V).s[blockIdx.y];                     \
  const uint32_t* __restrict__ ray_list = sv.ray_list;        \
  const RayDesc* __restrict__ rays = sv.rays;                 \
  const uint32_t* __restrict__ cnt = sv.cnt;                  \
  const uint32_t* __restrict__ lp = sv.lp;                    \
  const unsigned long long* __restrict__ bt = sv.bt;          \
  uint64_t* __restrict__ pairs = sv.pairs;                    \
  const unsigned long long pairs_cap = sv.pairs_cap;          \
  Counters* C = sv.C;                                         \
  const FrameParams F = *sv.F; /* a COPY: through the pointer every loop iteration would re-load the fields it uses */

template <int LPR>
__global__ void __launch_bounds__(256) k_emit(BatchView V, TileTable T, Pool P) {
  KS_SLOT_ARGS(V)
  // launched over an upper bound of rays: blocks past the live ray count leave at once (block 0 stays:
  // it publishes the total)
  if (blockIdx.x != 0 && blockIdx.x * (256u / LPR) >= C->n_rays) return;
  // exclusive prefix of the block totals (every block redundantly; <= a few thousand values)
  extern __shared__ unsigned long long s_bt[];
  __shared__ unsigned long long s_carry;
  const uint32_t nb = (scan_length(F) + kScanBlock - 1) / kScanBlock;
  if (threadIdx.x == 0) s_carry = 0ull;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
    const uint32_t b = b0 + threadIdx.x;
    const unsigned long long v = b < nb ? bt[b] : 0ull;
    // block-wide exclusive scan of 256 values
    __shared__ unsigned long long s_w[4];
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
  const unsigned long long total = s_carry;
  if (blockIdx.x == 0 && threadIdx.x == 0) C->n_pairs = total;
  if (total > pairs_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&C->err, kErrPairs);
    return;
  }
  if (C->err & (kErrLabel | kErrIndex | kErrExact)) return;
  const LaneGroup<LPR> G;
  const uint32_t n_rays = C->n_rays;
  const uint32_t r = (blockIdx.x * 256u + threadIdx.x) / LPR;
  const bool have = r < n_rays;
  uint32_t p = 0, count = 0;
  unsigned long long base = 0;
  uint64_t key_lo = 0;  // info byte | sequence
  bool clearing = false;
  uint64_t own_key = 0;
  Dda round{};
  if (have) {
    p = ray_list[r];
    const RayDesc d = rays[ray_index(F, p)];
    clearing = ((d.info >> 10) & 1u) != 0;
    const uint32_t si = p + ((F.method == KS_METHOD_MERGED && clearing) ? F.n : 0u);
    count = cnt[si] & ~kCntBroke;
    base = s_bt[si / kScanBlock] + lp[si];
    key_lo = ((uint64_t)((d.info & 0x1fu) | (((d.info >> 8) & 3u) << 5) | (((d.info >> 10) & 1u) << 7)) << 56) |
             (uint64_t)((F.method == KS_METHOD_MERGED && clearing) ? (p | F.clear_bit) : p);
    own_key = F.ray_keys ? F.ray_keys[p] : 0ull;
    round.setup(F.T.t, {d.px, d.py, d.pz}, clearing, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc,
                /*cast_from_origin=*/F.method == KS_METHOD_MERGED);
  }
  // With anti-grazing (merged, off by default) some steps of the walk emit nothing: `count` counts the
  // emitting steps (k_count_grazing), the walk covers all steps of the ray.
  const uint32_t walk = have ? (F.grazing_keys ? (uint32_t)round.steps + 1u : count) : 0u;
  uint32_t emitted = 0;
  uint64_t last_tile = kEmpty64;
  uint32_t slot = 0;
  for (uint32_t s0 = 0; __ballot(s0 < walk) != 0ull; s0 += LPR) {
    const bool in = s0 < walk;
    const Dda mine = dda_at_lane<LPR>(round, G.l);
    const bool step_on = in && (s0 + G.l < walk);
    const bool emit = step_on && !grazing_skip(F, mine.cx, mine.cy, mine.cz, clearing, own_key);
    uint32_t hpos = 0, got = 0;
    bool need_tile = false;
    if (emit) {
      const uint64_t tk = pack_tile(mine.cx >> 3, mine.cy >> 3, mine.cz >> 3);
      if (tk != last_tile) {
        need_tile = true;
        last_tile = tk;
        got = tile_slot_nowait(T, C, tk, &hpos);
      }
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
      if (slot < T.max_tiles) { P.updated[slot] = 1; P.dirty[slot] = 1; }
      else atomicOr(&C->err, kErrPool);  // pool exhausted (now or in an earlier frame): this frame is not applied
    }
    const uint64_t em = G.bits(__ballot(emit));
    if (emit) {
      const uint32_t local = (uint32_t)(mine.cx & 7) + 8u * ((uint32_t)(mine.cy & 7) + 8u * (uint32_t)(mine.cz & 7));
      const uint32_t k = emitted + (uint32_t)__popcll(em & ((1ull << G.l) - 1ull));
      pairs[base + k] = ((uint64_t)(slot * (uint32_t)kTileVoxels + local) << F.seq_bits) | key_lo;
    }
    emitted += (uint32_t)__popcll(em);
    if (in && s0 + LPR < walk) dda_next_round<LPR>(G, round, mine);
  }
}

// k_emit_lane — the same emission without anti-grazing (every step of a ray emits): ONE LANE PER RAY walks the
// first 32 voxels serially (consecutive voxels share their tile: one table lookup per tile crossing), the few
// rays that go further are then taken one at a time by the whole wavefront (exact parallel caster).
template <int RPW>
__global__ void __launch_bounds__(256) k_emit_lane(BatchView V, TileTable T, Pool P) {
  KS_SLOT_ARGS(V)
  if (blockIdx.x != 0 && blockIdx.x * 4u * (uint32_t)RPW >= C->n_rays) return;
  extern __shared__ unsigned long long s_bt[];
  __shared__ unsigned long long s_carry;
  __shared__ float s_e[4][3 * kES];
  const uint32_t nb = (scan_length(F) + kScanBlock - 1) / kScanBlock;
  if (threadIdx.x == 0) s_carry = 0ull;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
    const uint32_t b = b0 + threadIdx.x;
    const unsigned long long v = b < nb ? bt[b] : 0ull;
    __shared__ unsigned long long s_w[4];
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
  const unsigned long long total = s_carry;
  if (blockIdx.x == 0 && threadIdx.x == 0) C->n_pairs = total;
  if (total > pairs_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&C->err, kErrPairs);
    return;
  }
  if (C->err & (kErrLabel | kErrIndex | kErrExact)) return;
  const uint32_t lane = lane_id();
  const uint32_t n_rays = C->n_rays;
  const uint32_t r = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (uint32_t)RPW + lane;
  uint32_t count = 0;
  unsigned long long base = 0;
  uint64_t key_lo = 0;
  Dda dda{};
  if (lane < (uint32_t)RPW && r < n_rays) {
    const uint32_t p = ray_list[r];
    const RayDesc d = rays[ray_index(F, p)];
    const bool clearing = ((d.info >> 10) & 1u) != 0;
    const uint32_t si = p + ((F.method == KS_METHOD_MERGED && clearing) ? F.n : 0u);
    count = cnt[si] & ~kCntBroke;
    base = s_bt[si / kScanBlock] + lp[si];
    key_lo = ((uint64_t)((d.info & 0x1fu) | (((d.info >> 8) & 3u) << 5) | (((d.info >> 10) & 1u) << 7)) << 56) |
             (uint64_t)((F.method == KS_METHOD_MERGED && clearing) ? (p | F.clear_bit) : p);
    dda.setup(F.T.t, {d.px, d.py, d.pz}, clearing, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc,
              /*cast_from_origin=*/F.method == KS_METHOD_MERGED);
  }
  auto wait_slot = [&](uint32_t got, uint32_t hpos) -> uint32_t {
    // the allocating lane has stored the slot before it left tile_slot_nowait: this terminates
    uint32_t spins = 0;
    while (got == kSlotPending) {
      got = __hip_atomic_load(&T.ent[hpos].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (++spins > (1u << 22)) {
        atomicOr(&C->err, kErrTable);
        got = kSlotBad;
      }
    }
    if (got < T.max_tiles) { P.updated[got] = 1; P.dirty[got] = 1; }
    else atomicOr(&C->err, kErrPool);  // pool exhausted (now or in an earlier frame): this frame is not applied
    return got;
  };
  const unsigned long long long_mask = __ballot(count > kLaneWalk);
  const bool by_wave = tails_by_wavefront(long_mask, count);
  const uint32_t own = (by_wave && count > kLaneWalk) ? kLaneWalk : count;
  uint64_t last_tile = kEmpty64;
  uint32_t slot = 0;
  for (uint32_t s = 0; __ballot(s < own) != 0ull; ++s) {
    const bool on = s < own;
    uint32_t hpos = 0, got = 0;
    bool need_tile = false;
    if (on) {
      const uint64_t tk = pack_tile(dda.cx >> 3, dda.cy >> 3, dda.cz >> 3);
      if (tk != last_tile) {
        need_tile = true;
        last_tile = tk;
        got = tile_slot_nowait(T, C, tk, &hpos);
      }
    }
    if (need_tile) slot = wait_slot(got, hpos);  // (the wave has reconverged)
    if (on) {
      const uint32_t local = (uint32_t)(dda.cx & 7) + 8u * ((uint32_t)(dda.cy & 7) + 8u * (uint32_t)(dda.cz & 7));
      const uint64_t key = ((uint64_t)(slot * (uint32_t)kTileVoxels + local) << F.seq_bits) | key_lo;
      pairs[base + s] = key;
    }
    dda.advance(on);
  }
  float* escr = s_e[threadIdx.x >> 6];
  for (unsigned long long todo = by_wave ? long_mask : 0ull; todo != 0ull; todo &= todo - 1ull) {
    const int j = __ffsll((long long)todo) - 1;
    Dda ust = dda_bcast(dda, j);  // state at step kLaneWalk
    const uint32_t c_j = __shfl(count, j);
    const unsigned long long base_j = __shfl(base, j);
    const uint64_t key_j = __shfl(key_lo, j);
    if (dda_parallel_ok(ust)) {
      for (uint32_t s0 = kLaneWalk; s0 < c_j; s0 += 64) {
        dda_round64(ust, escr, lane, [&](uint32_t rr, int vx, int vy, int vz) {
          if (s0 + rr < c_j) {
            uint32_t hpos = 0;
            const uint32_t got = tile_slot_nowait(T, C, pack_tile(vx >> 3, vy >> 3, vz >> 3), &hpos);
            const uint32_t sl = wait_slot(got, hpos);
            const uint32_t local = (uint32_t)(vx & 7) + 8u * ((uint32_t)(vy & 7) + 8u * (uint32_t)(vz & 7));
            pairs[base_j + s0 + rr] = ((uint64_t)(sl * (uint32_t)kTileVoxels + local) << F.seq_bits) | key_j;
          }
        });
      }
    } else if ((int)lane == j) {
      for (uint32_t s = kLaneWalk; s < count; ++s) {
        uint32_t hpos = 0;
        const uint32_t got = tile_slot_nowait(T, C, pack_tile(dda.cx >> 3, dda.cy >> 3, dda.cz >> 3), &hpos);
        const uint32_t sl = wait_slot(got, hpos);
        const uint32_t local = (uint32_t)(dda.cx & 7) + 8u * ((uint32_t)(dda.cy & 7) + 8u * (uint32_t)(dda.cz & 7));
        pairs[base + s] = ((uint64_t)(sl * (uint32_t)kTileVoxels + local) << F.seq_bits) | key_lo;
        dda.advance();
      }
    }
  }
}

// merged + anti-grazing: the number of steps of each bundle's ray that emit an update
template <int LPR>
__global__ void __launch_bounds__(256) k_count_grazing(BatchView V) {
  const SlotView& sv = V.s[blockIdx.y];
  const uint32_t* __restrict__ ray_list = sv.ray_list;
  const RayDesc* __restrict__ rays = sv.rays;
  uint32_t* __restrict__ cnt = sv.cnt;
  const Counters* C = sv.C;
  const FrameParams F = *sv.F;  // a COPY: through the pointer every loop iteration would re-load the fields it uses (they may alias the stores)
  const LaneGroup<LPR> G;
  const uint32_t r = (blockIdx.x * 256u + threadIdx.x) / LPR;
  if (r >= C->n_rays) return;
  const uint32_t p = ray_list[r];
  const RayDesc d = rays[ray_index(F, p)];
  const bool clearing = ((d.info >> 10) & 1u) != 0;
  const uint64_t own_key = F.ray_keys ? F.ray_keys[p] : 0ull;
  Dda round;
  round.setup(F.T.t, {d.px, d.py, d.pz}, clearing, F.carving != 0, F.max_ray, F.voxel_size_inv, F.trunc, true);
  const uint32_t walk = (uint32_t)round.steps + 1u;
  uint32_t total = 0;
  for (uint32_t s0 = 0; s0 < walk; s0 += LPR) {
    const Dda mine = dda_at_lane<LPR>(round, G.l);
    const bool emit = (s0 + G.l < walk) && !grazing_skip(F, mine.cx, mine.cy, mine.cz, clearing, own_key);
    total += (uint32_t)__popcll(G.bits(__ballot(emit)));
    if (s0 + LPR < walk) dda_next_round<LPR>(G, round, mine);
  }
  if (G.l == 0) cnt[p + (clearing ? F.n : 0u)] = total;
}

// The frame's parameters reach device memory through a kernel argument (a pageable-memory memcpy would go
// through the runtime's staging path and can block the enqueueing thread).
__global__ void __launch_bounds__(64) k_set_params(FrameParams F, FrameParams* __restrict__ out) {
  if (threadIdx.x == 0) *out = F;
}
// ... of all the frames of a batch in one launch (the kernel arguments hold them: 8 x sizeof(FrameParams) < 4 KB)
struct ParamsBatch {
  FrameParams F[kBatchMax];
  FrameParams* out[kBatchMax];
};
static_assert(sizeof(ParamsBatch) <= 3840, "kernel arguments");
__global__ void __launch_bounds__(64) k_set_params_batch(ParamsBatch P) {
  const uint32_t* src = (const uint32_t*)&P.F[blockIdx.x];
  uint32_t* dst = (uint32_t*)P.out[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < sizeof(FrameParams) / 4u; i += 64u) dst[i] = src[i];
}
static_assert(sizeof(FrameParams) % 4 == 0, "copied as words");

// End of stage B: the frame's counters and the persistent tile count go to pinned host memory,
// and the counters are cleared for the slot's next frame (the tail only uses n_long, which it
// expects to be zero): no memset launch per frame.
__global__ void __launch_bounds__(64) k_publish(BatchView V, const uint32_t* __restrict__ n_tiles) {
  static_assert(sizeof(Counters) == 32, "snapshot layout");
  Counters* __restrict__ C = V.s[blockIdx.x].C;      // one workgroup per frame of the batch
  uint32_t* __restrict__ host_snap = V.s[blockIdx.x].host_snap;
  if (threadIdx.x < 8) {
    host_snap[threadIdx.x] = ((const uint32_t*)C)[threadIdx.x];
    ((uint32_t*)C)[threadIdx.x] = 0u;
  }
  if (threadIdx.x == 8) host_snap[8] = *n_tiles;
  const uint32_t* eo_stats = V.s[blockIdx.x].eo_stats;
  if (eo_stats && threadIdx.x >= 9 && threadIdx.x < 12) host_snap[threadIdx.x] = eo_stats[threadIdx.x - 9];
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
