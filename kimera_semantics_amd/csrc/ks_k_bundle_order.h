// ks_k_bundle_order.h — `merged`: the integration order of the bundles, as the reference has it.
//
// The reference integrates the ray bundles in the ITERATION ORDER OF A libstdc++ std::unordered_map
// (voxel_map / clear_map, filled by vxb::MergedTsdfIntegrator::bundleRays, walked by integrateVoxels,
// [K:src/semantic_tsdf_integrator_merged.cpp:108-124, 200-232]).  Per voxel the update is an order-
// dependent f32 recurrence, so the labels are only bit-identical if the bundles are applied in that order.
// The order is a pure function of
//   (a) the distinct end-voxel keys in first-insertion order (element id = insertion index),
//   (b) their hash codes (vxb::LongIndexHash: 32 bits), and
//   (c) the container's rehash schedule: bucket count b_e is taken over when element t_e is inserted
//       (probed from the host's own libstdc++ at ks_create: 13, 29, 59, 127, ... ),
// because _M_insert_bucket_begin puts a node at the front of its bucket's segment and a bucket that
// becomes non-empty at the front of the whole list, and _M_rehash_aux re-inserts the list, in list
// order, with the same rule.  Hence, for epoch e (bucket count b_e, elements [0, min(t_{e+1}, B)) ):
//     position(x) = rank of x in the previous epoch's list      for x < t_e   (re-inserted)
//                 = x                                             for x >= t_e  (inserted afterwards)
//     list L_e    = elements ordered by ( min position in x's bucket  DESC,  position(x) DESC )
// (tools/umap_order_model.cpp checks this closed form against the real container on 3.6e6 keys.)
//
// The kernels evaluate the recurrence epoch by epoch, every element in parallel, with no sort: with
// j = total - 1 - position ("reverse position"), an epoch's list is the buckets in ascending order of their
// LARGEST j, elements of a bucket in ascending j:
//   k_bo_link(e) : every element computes its bucket, pushes j on the bucket's chain (atomicExch)
//   k_bo_walk(e) : every element walks its bucket's chain (load factor <= 1: a couple of nodes) for the
//                  bucket's largest j, its size and the number of smaller j; the element with the largest j
//                  contributes the bucket's size to a scan over j
//   k_bo_link(e+1): rank = scan[largest j of my bucket] + smaller j's in my bucket — the position of the
//                  next epoch, or the final integration rank.
// Both maps (normal bundles, clearing bundles) are handled by the same launches (blockIdx.y).
// The number of epochs launched depends only on the slot capacity; epochs a frame does not reach exit at once.
#pragma once
#include "ks_types.h"

namespace ksk {

constexpr int kBoMaxEpochs = 32;
constexpr uint32_t kBoBlock = 1024;          // elements per workgroup of the scans below
constexpr uint32_t kBoEmpty = 0xffffffffu;

struct BoSchedule {
  uint32_t n_epochs;
  uint32_t t[kBoMaxEpochs + 1];     // epoch e covers insertion indices [t[e], t[e+1]); t[n_epochs] = UINT32_MAX
  uint32_t b[kBoMaxEpochs];         // bucket count of epoch e
  uint32_t head_off[kBoMaxEpochs];  // offset of epoch e's bucket heads in BoMap::head
};

struct BoMap {
  uint32_t* H;        // [cap] hash code of element id
  uint32_t* rank;     // [cap] integration rank of element id (the result)
  uint32_t* next[2];  // [cap] by j: next node of the bucket chain        (two sets: epoch parity)
  uint32_t* idj[2];   // [cap] by j: element id
  uint32_t* kj[2];    // [cap] by j: bucket
  uint32_t* lp;       // [cap] by j: exclusive prefix of the segment sizes inside the workgroup's kBoBlock
  uint32_t* gm;       // [cap] by j: largest j of the bucket
  uint32_t* cj;       // [cap] by j: elements of the bucket with a smaller j
  uint32_t* bt;       // [cap / kBoBlock + 1] workgroup totals of that scan
  uint32_t* head;     // [sum of b_e] chain heads, kBoEmpty between frames
};

struct BoCtx {
  BoMap m[2];                  // 0: voxel_map (normal bundles), 1: clear_map
  uint32_t* B;                 // [2] number of bundles per map (device side)
  const BoSchedule* sched;
  uint32_t* flag;              // [2 cap] by position (+ n for clearing): a bundle's first point sits here
  uint32_t* flag_lp;           // [2 cap] exclusive prefix of flag inside the workgroup's kBoBlock
  uint32_t* flag_bt;           // [2 cap / kBoBlock + 1] workgroup totals
  uint32_t* t_of_head;         // [cap] by sorted index i of a bundle's head: its element id
};

// exclusive scan of `mine` over the workgroup (1024 threads); returns the exclusive prefix, *total = the sum
__device__ __forceinline__ uint32_t bo_block_scan(uint32_t mine, uint32_t* total) {
  __shared__ uint32_t s_wave[16];
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  uint32_t x = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += y;
  }
  if (lane == 63) s_wave[wave] = x;
  __syncthreads();
  uint32_t wbase = 0, all = 0;
  for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
    const uint32_t v = s_wave[w];
    if (w < wave) wbase += v;
    all += v;
  }
  __syncthreads();
  *total = all;
  return wbase + x - mine;
}

// s_out[i] = sum of bt[0..i) for i in [0, nb] (LDS, nb + 1 entries; every workgroup redundantly: nb is a few
// hundred).  All threads of the workgroup must call.
__device__ __forceinline__ void bo_prefix_totals(const uint32_t* __restrict__ bt, uint32_t nb, uint32_t* s_out) {
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0u;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nb; b0 += blockDim.x) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = b < nb ? bt[b] : 0u;
    uint32_t tot;
    const uint32_t ex = bo_block_scan(v, &tot);
    const uint32_t carry = s_carry;
    if (b < nb) s_out[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) s_out[nb] = s_carry;
  __syncthreads();
}

// ---- insertion indices: scan of the first-point flags in integration-position order -------------------------
__global__ void __launch_bounds__(kBoBlock) k_bo_scan_flags(uint32_t n, BoCtx X) {
  const uint32_t n2 = 2u * n;
  const uint32_t i = blockIdx.x * kBoBlock + threadIdx.x;
  if (blockIdx.x * kBoBlock >= n2) return;
  const uint32_t v = i < n2 ? X.flag[i] : 0u;
  uint32_t tot;
  const uint32_t ex = bo_block_scan(v, &tot);
  if (i < n2) X.flag_lp[i] = ex;
  if (threadIdx.x == 0) X.flag_bt[blockIdx.x] = tot;
}

// element id and hash code of every bundle (thread per sorted point; heads only); bundle counts
__global__ void __launch_bounds__(kBoBlock) k_bo_init(uint32_t n, const uint64_t* __restrict__ skeys,
                                                      const uint32_t* __restrict__ svals, BoCtx X) {
  extern __shared__ uint32_t s_tot[];  // prefix of the workgroup totals of the flag scan
  if (blockIdx.x * kBoBlock >= n && blockIdx.x != 0) return;
  const uint32_t nb = (2u * n + kBoBlock - 1u) / kBoBlock;
  bo_prefix_totals(X.flag_bt, nb, s_tot);
  const uint32_t n_normal = s_tot[n / kBoBlock] + X.flag_lp[n];  // flags below index n (n < 2n: the entry exists for n > 0)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    X.B[0] = n_normal;
    X.B[1] = s_tot[nb] - n_normal;
  }
  const uint32_t i = blockIdx.x * kBoBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = skeys[i];
  if (key == kEmpty64 || (i != 0 && skeys[i - 1] == key)) return;
  const uint32_t clearing = (uint32_t)(key >> 63);
  const uint32_t si = svals[i] + clearing * n;
  const uint32_t t = s_tot[si / kBoBlock] + X.flag_lp[si] - clearing * n_normal;
  const int vx = (int)((key >> 42) & 0x1fffffu) - kCoordBias, vy = (int)((key >> 21) & 0x1fffffu) - kCoordBias,
            vz = (int)(key & 0x1fffffu) - kCoordBias;
  X.m[clearing].H[t] = index_hash(vx, vy, vz);  // vxb::LongIndexHash
  X.t_of_head[i] = t;
}

// one epoch transition: ranks of epoch e-1 (-> final result, or positions of epoch e) and the chains of epoch e
// prev_small: epochs [0, e) were evaluated by k_bo_small, which left every element's rank in M.rank
__device__ __forceinline__ void bo_link_body(const BoCtx& X, int e, int prev_small, uint32_t map, uint32_t block, uint32_t* s_tot) {
  const BoSchedule& S = *X.sched;
  const BoMap& M = X.m[map];
  const uint32_t Bm = X.B[map];
  const bool prev_active = e >= 1 && Bm > S.t[e - 1];
  const bool cur_active = e < (int)S.n_epochs && Bm > S.t[e];
  if (!prev_active && !cur_active) return;
  if (prev_small && !cur_active) return;  // the ranks k_bo_small left are this map's final ranks
  const uint32_t prev_total = prev_active ? (S.t[e] < Bm ? S.t[e] : Bm) : 0u;
  const uint32_t cur_total = cur_active ? (S.t[e + 1] < Bm ? S.t[e + 1] : Bm) : 0u;
  const uint32_t i0 = block * kBoBlock;
  const uint32_t n_new = cur_active ? cur_total - S.t[e] : 0u;
  if (i0 >= prev_total && i0 >= n_new) return;
  const int cur = e & 1, prv = cur ^ 1;
  const uint32_t i = i0 + threadIdx.x;
  auto insert = [&](uint32_t id, uint32_t pos) {
    const uint32_t j = cur_total - 1u - pos;
    const uint32_t k = M.H[id] % S.b[e];
    const uint32_t nx = atomicExch(&M.head[S.head_off[e] + k], j);
    M.next[cur][j] = nx;
    M.idj[cur][j] = id;
    M.kj[cur][j] = k;
  };
  if (prev_active && prev_small) {
    if (i < prev_total) insert(i, M.rank[i]);
  } else if (prev_active) {
    bo_prefix_totals(M.bt, (prev_total + kBoBlock - 1u) / kBoBlock, s_tot);
    if (i < prev_total) {
      const uint32_t g = M.gm[i];
      const uint32_t out = s_tot[g / kBoBlock] + M.lp[g] + M.cj[i];
      const uint32_t id = M.idj[prv][i];
      M.head[S.head_off[e - 1] + M.kj[prv][i]] = kBoEmpty;  // leave the chains empty for the next frame
      if (cur_active) insert(id, out);
      else M.rank[id] = out;
    }
  }
  if (i < n_new) insert(S.t[e] + i, S.t[e] + i);
}
__global__ void __launch_bounds__(kBoBlock) k_bo_link(BoCtx X, int e, int prev_small) {
  extern __shared__ uint32_t s_tot[];
  bo_link_body(X, e, prev_small, blockIdx.y, blockIdx.x, s_tot);
}

__device__ __forceinline__ void bo_walk_body(const BoCtx& X, int e, uint32_t map, uint32_t block) {
  const BoSchedule& S = *X.sched;
  const BoMap& M = X.m[map];
  const uint32_t Bm = X.B[map];
  if (!(Bm > S.t[e])) return;
  const uint32_t total = S.t[e + 1] < Bm ? S.t[e + 1] : Bm;
  if (block * kBoBlock >= total) return;
  const int cur = e & 1;
  const uint32_t j = block * kBoBlock + threadIdx.x;
  uint32_t w = 0;
  if (j < total) {
    const uint32_t* nxt = M.next[cur];
    uint32_t cnt = 0, mx = 0, smaller = 0;
    for (uint32_t x = M.head[S.head_off[e] + M.kj[cur][j]]; x != kBoEmpty; x = nxt[x]) {
      ++cnt;
      mx = x > mx ? x : mx;
      smaller += x < j ? 1u : 0u;
    }
    M.gm[j] = mx;
    M.cj[j] = smaller;
    w = (mx == j) ? cnt : 0u;
  }
  uint32_t tot;
  const uint32_t ex = bo_block_scan(w, &tot);
  if (j < total) M.lp[j] = ex;
  if (threadIdx.x == 0) M.bt[block] = tot;
}
__global__ void __launch_bounds__(kBoBlock) k_bo_walk(BoCtx X, int e) { bo_walk_body(X, e, blockIdx.y, blockIdx.x); }

// The host launches the epochs a frame of `hint` bundles can reach — the bundle counts of the frames before it, with a margin —
// instead of those a frame of n POINTS could (a 640x480 / 5 cm frame has 1.5e4 bundles of 3e5 points: four of its seven launch
// pairs did nothing).  This kernel, launched last, is the rest of the recurrence for a map that has MORE bundles than the hint: one
// workgroup per map walks the remaining epochs, workgroup-sized block after block (slow — a frame that jumps past the margin
// pays for it once — and exact: the same bodies in the same order).  e_last = the epoch whose k_bo_link was launched without
// its k_bo_walk.
__global__ void __launch_bounds__(kBoBlock) k_bo_rest(BoCtx X, int e_last) {
  extern __shared__ uint32_t s_tot[];
  const BoSchedule& S = *X.sched;
  const uint32_t map = blockIdx.x;
  const uint32_t Bm = X.B[map];
  if (e_last >= (int)S.n_epochs || !(Bm > S.t[e_last])) return;   // (the common case: the launched epochs were all of them)
  const uint32_t nblk = (Bm + kBoBlock - 1u) / kBoBlock;
  for (int e = e_last;;) {
    for (uint32_t b = 0; b < nblk; ++b) bo_walk_body(X, e, map, b);
    __threadfence();
    __syncthreads();
    ++e;
    for (uint32_t b = 0; b < nblk; ++b) bo_link_body(X, e, 0, map, b, s_tot);
    __threadfence();
    __syncthreads();
    if (!(e < (int)S.n_epochs && Bm > S.t[e])) break;
  }
}

// The epochs that fit one workgroup's LDS — bucket counts up to kBoSmallBuckets, i.e. the first kBoSmallBuckets
// elements — in ONE launch (a 640x480 / 5 cm frame has ~1e4 bundles: two launch pairs remain of fifteen).  Same
// recurrence as k_bo_link / k_bo_walk, the chains, the scan and the ranks in LDS; one workgroup per map.
constexpr uint32_t kBoSmallBuckets = 5087;   // libstdc++'s 10th bucket count (13, 29, ..., 2357, 5087)
constexpr uint32_t kBoSmallItems = (kBoSmallBuckets + kBoBlock - 1) / kBoBlock;  // elements per thread
__global__ void __launch_bounds__(kBoBlock) k_bo_small(BoCtx X, int e_end) {
  __shared__ uint32_t s_head[kBoSmallBuckets];  // chain heads (reverse positions)
  __shared__ uint16_t s_next[kBoSmallBuckets], s_idj[kBoSmallBuckets], s_gm[kBoSmallBuckets], s_rank[kBoSmallBuckets];
  __shared__ uint16_t s_w[kBoSmallBuckets], s_cj[kBoSmallBuckets];
  __shared__ uint32_t s_H[kBoSmallBuckets];     // hash codes of the elements the small epochs cover
  const BoSchedule& S = *X.sched;
  const BoMap& M = X.m[blockIdx.x];
  const uint32_t Bm = X.B[blockIdx.x];
  const uint32_t tid = threadIdx.x;
  constexpr uint16_t kNil = 0xffffu;
  {
    const uint32_t cover = S.t[e_end] < Bm ? S.t[e_end] : Bm;
    for (uint32_t id = tid; id < cover; id += kBoBlock) s_H[id] = M.H[id];
  }
  __syncthreads();
  for (int e = 0; e < e_end; ++e) {
    if (!(Bm > S.t[e])) break;  // uniform
    const uint32_t total = S.t[e + 1] < Bm ? S.t[e + 1] : Bm;
    const uint32_t b = S.b[e], t_e = S.t[e];
    for (uint32_t k = tid; k < b; k += kBoBlock) s_head[k] = kBoEmpty;
    __syncthreads();
    // link: reverse position j = total - 1 - position
    for (uint32_t id = tid; id < total; id += kBoBlock) {
      const uint32_t pos = id < t_e ? (uint32_t)s_rank[id] : id;
      const uint32_t j = total - 1u - pos;
      const uint32_t nx = atomicExch(&s_head[s_H[id] % b], j);
      s_next[j] = nx == kBoEmpty ? kNil : (uint16_t)nx;
      s_idj[j] = (uint16_t)id;
    }
    __syncthreads();
    // walk: the bucket's largest j, its size, the smaller j's in it; the largest j carries the size into the scan
    for (uint32_t j = tid; j < total; j += kBoBlock) {
      uint32_t cnt = 0, mx = 0, smaller = 0;
      uint32_t x = s_head[s_H[s_idj[j]] % b];
      while (x != kBoEmpty) {
        ++cnt;
        mx = x > mx ? x : mx;
        smaller += x < j ? 1u : 0u;
        const uint16_t n = s_next[x];
        x = n == kNil ? kBoEmpty : (uint32_t)n;
      }
      s_gm[j] = (uint16_t)mx;
      s_cj[j] = (uint16_t)smaller;
      s_w[j] = (uint16_t)(mx == j ? cnt : 0u);
    }
    __syncthreads();
    // exclusive scan of s_w over j: thread t owns j in [t * kBoSmallItems, (t + 1) * kBoSmallItems)
    {
      uint32_t mine = 0;
      const uint32_t j0 = tid * kBoSmallItems;
      for (uint32_t q = 0; q < kBoSmallItems; ++q)
        if (j0 + q < total) mine += s_w[j0 + q];
      uint32_t tot;
      uint32_t run = bo_block_scan(mine, &tot);
      (void)tot;
      for (uint32_t q = 0; q < kBoSmallItems; ++q)
        if (j0 + q < total) {
          const uint32_t v = s_w[j0 + q];
          s_w[j0 + q] = (uint16_t)run;   // exclusive prefix
          run += v;
        }
    }
    __syncthreads();
    for (uint32_t j = tid; j < total; j += kBoBlock) s_rank[s_idj[j]] = (uint16_t)((uint32_t)s_w[s_gm[j]] + (uint32_t)s_cj[j]);
    __syncthreads();
  }
  // ranks of the elements the small epochs covered: final if the map ended there, positions of the next epoch otherwise
  const uint32_t done = S.t[e_end] < Bm ? S.t[e_end] : Bm;
  for (uint32_t id = tid; id < done; id += kBoBlock) M.rank[id] = s_rank[id];
}

// integration id q of the bundle whose head sits at sorted index i: canonical order = the position of its
// first point; reference order = its rank in the container's iteration order (clearing bundles after the
// normal ones: ids stay unique, and cnt[] keeps its two halves)
__device__ __forceinline__ uint32_t bundle_id(const BoCtx& X, bool use_rank, const uint32_t* __restrict__ svals, uint32_t i,
                                              bool clearing) {
  if (!use_rank) return svals[i];
  const uint32_t t = X.t_of_head[i];
  return clearing ? X.B[0] + X.m[1].rank[t] : X.m[0].rank[t];
}

}  // namespace ksk
