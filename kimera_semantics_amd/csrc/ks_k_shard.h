// ks_k_shard.h — the EXACT frame-sharded integration of `fast` (ks_integrate_round_exact, include/ks_hip.h).
//
// A batch of frames integrated in parallel on several GPUs cannot be merged map by map into what ONE GPU integrating the
// frames in order holds: the voxel update clamps per update (distance at +-truncation, weight at max_weight) and its f32
// sums are order dependent — the tile merge of ks_reduce is a different arithmetic (labels agree to > 99 %, distances to
// ~1e-4: tests/test_parallel_gpu.py).  What CAN be split exactly is the work before the update: with
// clear_checks_every_n_frames = 1 a frame's two approximate sets never see another frame's entries, so a frame's ray
// casting, its early-out and its list of (voxel, integration position) updates do not depend on the map at all.  So:
//
//   rank f mod G   marches frame f (stage A, the early-out, the emission — with the set offsets of global frame f) against
//                  a slot numbering of its own (no voxel data behind it), evaluates the state-independent half of every
//                  update (computeDistance, weight drop-off) and ships each update to the rank that owns the voxel's tile:
//                  { tile key << 9 | voxel in tile, info byte << 24 | integration position, sdf, update weight } = 20 bytes
//   the owner      takes the frames of a round in frame order; per frame: its tiles' updates from the rank that marched the
//                  frame, into its own table (get-or-insert), stable sort by voxel, the state recurrence per voxel run.
//
// Every voxel then sees exactly the update sequence of the sequential integration, (frame, position) ascending, with the
// same operands: the owners' tiles are bit for bit the one-GPU map.  On the wire: 20 bytes per update (~8 MB per 640x480
// frame in all, an eighth of it per peer) instead of 64 KiB per touched tile.
//
// One coupling between frames survives in the reference: ApproxHashSet's zero-initialised slots "contain" hash 0, so the
// voxel whose index hashes to 0 — the one at the world origin, in any scene smaller than kilometres — is seen as already
// observed while its slot has never been written, and whether it has been written depends on every earlier frame
// ([K:include/kimera_semantics/semantic_tsdf_integrator_fast.h:114-130]).  A rank that marches only its own frames has not
// written what the frames in between did.  k_shard_export raises a flag when an update touches that voxel; the round
// reports it (ks_round_stats::origin_voxel_touched) and the caller knows the result may differ there from the sequential one.
#pragma once
#include "ks_types.h"

namespace ksk {

__device__ __forceinline__ uint64_t shard_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

constexpr uint32_t kShardSeqBits = 24;   // the integration position inside a record / a rebuilt pair key

// Lane per update of the marched frame, in emission (= integration) order: its record, its owner.
// okey[i] = owner << 56 | i is what the stable one-pass sort by owner permutes.
__global__ void __launch_bounds__(256) k_shard_export(FrameParams F, unsigned long long n_pairs, const uint64_t* __restrict__ pairs,
                                                      const RayDesc* __restrict__ rays, const uint64_t* __restrict__ slot_keys,
                                                      uint32_t world, uint64_t* __restrict__ okey, uint64_t* __restrict__ gkey,
                                                      uint32_t* __restrict__ seq, float* __restrict__ sdf_out, float* __restrict__ uw_out,
                                                      uint32_t* __restrict__ counts /* [world] + flag */) {
  __shared__ uint32_t s_cnt[64];
  if (threadIdx.x < 64u) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (i < n_pairs) {
    const uint64_t key = pairs[i];
    const uint32_t vox = (uint32_t)(key >> F.seq_bits);
    const uint32_t pos = (uint32_t)key & F.point_mask;
    const uint64_t tk = slot_keys[vox >> 9];
    const uint32_t local = vox & 511u;
    int tx, ty, tz;
    unpack_tile(tk, tx, ty, tz);
    const int vx = tx * 8 + (int)(local & 7u), vy = ty * 8 + (int)((local >> 3) & 7u), vz = tz * 8 + (int)(local >> 6);
    const uint4 d0 = *((const uint4*)rays + (size_t)ray_index(F, pos) * 2);
    float sdf, uw;
    tsdf_operands(F.tsdf, F.T.t, {__uint_as_float(d0.x), __uint_as_float(d0.y), __uint_as_float(d0.z)}, vx, vy, vz, __uint_as_float(d0.w), sdf, uw);
    const uint32_t owner = (uint32_t)(shard_splitmix64(tk) % (uint64_t)world);
    okey[i] = ((uint64_t)owner << 56) | (uint64_t)i;
    gkey[i] = (tk << 9) | (uint64_t)local;
    seq[i] = ((uint32_t)(key >> 56) << kShardSeqBits) | pos;
    sdf_out[i] = sdf;
    uw_out[i] = uw;
    atomicAdd(&s_cnt[owner], 1u);
    if (index_hash(vx, vy, vz) == 0u) counts[world] = 1u;   // (the voxel whose slot "contains" it from the start: see above)
  }
  __syncthreads();
  if (threadIdx.x < world && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

// The records in owner order (the sorted okey holds where each came from).
__global__ void __launch_bounds__(256) k_shard_gather(unsigned long long n, const uint64_t* __restrict__ okey_sorted,
                                                      const uint64_t* __restrict__ gkey, const uint32_t* __restrict__ seq,
                                                      const float* __restrict__ sdf, const float* __restrict__ uw,
                                                      uint64_t* __restrict__ gkey_o, uint32_t* __restrict__ seq_o,
                                                      float* __restrict__ sdf_o, float* __restrict__ uw_o) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (i >= n) return;
  const unsigned long long j = okey_sorted[i] & 0x00ffffffffffffffull;
  gkey_o[i] = gkey[j];
  seq_o[i] = seq[j];
  sdf_o[i] = sdf[j];
  uw_o[i] = uw[j];
}

// Owner side.  The tile keys of a segment of records (for the get-or-insert of the tiles) ...
__global__ void __launch_bounds__(256) k_shard_tile_keys(uint32_t n, const uint64_t* __restrict__ gkey, uint64_t* __restrict__ tile_keys) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) tile_keys[i] = gkey[i] >> 9;
}
// ... and the pair keys in this rank's own slot numbering: [63:56] info byte | voxel << 24 | integration position; vals = the record.
__global__ void __launch_bounds__(256) k_shard_import(uint32_t n, TileTable T, Pool P, const uint64_t* __restrict__ gkey,
                                                      const uint32_t* __restrict__ seq, uint64_t* __restrict__ pairs,
                                                      uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint64_t g = gkey[i];
  const uint32_t slot = tile_lookup(T, g >> 9);
  const uint32_t s = seq[i];
  P.updated[slot] = 1;
  P.dirty[slot] = 1;
  pairs[i] = ((uint64_t)(s >> kShardSeqBits) << 56) | ((uint64_t)(slot * (uint32_t)kTileVoxels + (uint32_t)(g & 511u)) << kShardSeqBits) |
             (uint64_t)(s & ((1u << kShardSeqBits) - 1u));
  vals[i] = i;
}

// The state recurrence of every voxel run of a frame's sorted updates, operands from the records: a lane per run.
// (`fast` with the early-out: runs are a few updates long; a run of thousands is walked by its one lane — correct, slow.)
template <int COLOR_MODE>
__global__ void __launch_bounds__(256) k_shard_apply(FrameParams F, uint32_t n, const uint64_t* __restrict__ pairs,
                                                     const uint32_t* __restrict__ vals, const float* __restrict__ sdf_in,
                                                     const float* __restrict__ uw_in, Pool P, const uint32_t* __restrict__ label_lut) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint64_t key0 = pairs[i];
  const uint32_t vox = (uint32_t)(key0 >> kShardSeqBits);
  if (i > 0 && (uint32_t)(pairs[i - 1] >> kShardSeqBits) == vox) return;   // not the head of its run
  uint32_t* rec = (uint32_t*)(P.vox + (size_t)vox * 8);
  float dist = __uint_as_float(rec[0]), weight = __uint_as_float(rec[1]);
  uint32_t color = rec[2];
  float p[kNumLabels];
#pragma unroll
  for (int l = 0; l < kNumLabels; ++l) p[l] = __uint_as_float(rec[4 + l]);
  for (uint32_t j = i; j < n; ++j) {
    const uint64_t key = pairs[j];
    if ((uint32_t)(key >> kShardSeqBits) != vox) break;
    const uint32_t r = vals[j];
    tsdf_combine<false>(F.tsdf, sdf_in[r], uw_in[r], 0u, dist, weight, color);
    const uint32_t b = (uint32_t)(key >> 56);
    const uint32_t kind = (b >> 5) & 3u, lab = b & 0x1fu;
    if (kind == 1u) {
#pragma unroll
      for (int l = 0; l < kNumLabels; ++l) p[l] += ((uint32_t)l == lab) ? F.log_match : F.log_non_match;
    }
  }
  float bv = p[0];
  uint32_t bi = 0u;
#pragma unroll
  for (int l = 1; l < kNumLabels; ++l)
    if (p[l] > bv) {
      bv = p[l];
      bi = (uint32_t)l;
    }
  if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC) color = label_lut[bi & 255u];
  else if (COLOR_MODE == KS_COLOR_MODE_SEMANTIC_PROBABILITY) color = rainbow_color_map((double)(float)exp((double)bv));
  rec[0] = __float_as_uint(dist);
  rec[1] = __float_as_uint(weight);
  rec[2] = color;
  rec[3] = bi;
#pragma unroll
  for (int l = 0; l < kNumLabels; ++l) rec[4 + l] = __float_as_uint(p[l]);
  rec[25] = 1u;  // updated since the last voxel-level host sync
}

}  // namespace ksk
