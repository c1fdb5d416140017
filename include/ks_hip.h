/*
 * ks_hip.h — C ABI of the MI355X-native semantic TSDF integrator (libks_hip.so).
 *
 * This is the drop-in boundary underneath the reference's plugin surface.  The only caller
 * in a Kimera-Semantics deployment is the host adapter class
 * (kimera_semantics_amd/host/hip_semantic_tsdf_integrator.h) which derives from
 * voxblox::TsdfIntegratorBase + kimera::SemanticIntegratorBase exactly like the two CPU
 * integrators it replaces, and is returned by SemanticTsdfIntegratorFactory::create
 * (kimera_semantics/src/semantic_tsdf_integrator_factory.cpp:43-88).  Plain pointers and
 * sizes only; no C++/torch types.  One ks_ctx per GPU; a ks_ctx is NOT thread-safe (the
 * reference calls integratePointCloud from one ROS spinner thread).
 *
 * All functions return 0 on success or a negative KS_ERR_* code; ks_last_error() gives
 * text.  Where the reference would CHECK-abort (glog), the C ABI returns an error and the
 * adapter turns it back into LOG(FATAL) to preserve the convention.
 */
#ifndef KS_HIP_H_
#define KS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KS_NUM_LABELS 21 /* kimera::kTotalNumberOfLabels, kimera_semantics/include/kimera_semantics/common.h:26 */

enum {
  KS_OK = 0,
  KS_ERR_INVALID_ARG = -1,
  KS_ERR_LABEL_RANGE = -2,   /* label >= 21: CHECK_LT at semantic_tsdf_integrator_fast.cpp:134 / merged.cpp:278 */
  KS_ERR_PROBABILITY = -3,   /* CHECKs of semantic_integrator_base.cpp:98-107 */
  KS_ERR_HIP = -4,           /* HIP runtime failure */
  KS_ERR_POOL_FULL = -5,     /* voxel-tile pool exhausted (raise ks_config.max_tiles) */
  KS_ERR_INDEX_RANGE = -6,   /* a voxel index left the +-2^23 range the device packs */
  KS_ERR_NO_DEVICE = -7,
  KS_ERR_UNSUPPORTED = -8
};

enum { KS_METHOD_FAST = 0, KS_METHOD_MERGED = 1 };          /* factory names "fast"/"merged", semantic_tsdf_integrator_factory.h:49-54 */
enum { KS_COLOR_MODE_COLOR = 0, KS_COLOR_MODE_SEMANTIC = 1, KS_COLOR_MODE_SEMANTIC_PROBABILITY = 2 }; /* ColorMode, semantic_integrator_base.h:54-58 */
/* voxblox integration_order_mode (vxb::ThreadSafeIndexFactory::get, called at semantic_tsdf_integrator_fast.cpp:172-174 and
 * semantic_tsdf_integrator_merged.cpp:115-117).  Voxblox is not part of the reference tree (un-pinned upstream), so the
 * permutation behind "mixed" cannot be read there; both readings of MixedThreadSafeIndex::getNextIndexImpl are implemented,
 * with q = N / 1024 and positions s >= q * 1024 mapping to themselves:
 *   KS_ORDER_MIXED             idx = (s % q) * 1024 + s / q     upstream as published (number_of_groups_ = N / step_size_,
 *                              group_num = s % number_of_groups_, position_in_group = s / number_of_groups_)
 *   KS_ORDER_MIXED_1024_GROUPS idx = (s % 1024) * q + s / 1024  (what rounds 1-4 of this library assumed)
 * The C++ adapter does not guess: it reads the sequence of the ThreadSafeIndexFactory it is compiled against and selects
 * the matching value, or aborts (host/hip_semantic_tsdf_integrator.cpp: probe_mixed_order). */
enum { KS_ORDER_MIXED = 0, KS_ORDER_SORTED = 1, KS_ORDER_MIXED_1024_GROUPS = 2 };
/* merged: the order the ray bundles are integrated in.
 * REFERENCE (default): the iteration order of the libstdc++ std::unordered_map the reference keeps them in
 *   (semantic_tsdf_integrator_merged.cpp:108-124, 200-232; VoxelMap, common.h:37) — what the reference does at
 *   integrator_threads = 1, reproduced on the GPU from the keys' insertion order, hash codes and the container's
 *   rehash schedule (csrc/ks_k_bundle_order.h): results are bit-identical to the reference's.
 * CANONICAL: first-insertion order (a container-independent order; a few launches cheaper per frame). */
enum { KS_BUNDLE_ORDER_REFERENCE = 0, KS_BUNDLE_ORDER_CANONICAL = 1 };
enum { KS_EARLY_OUT_EXACT = 1 };                             /* value of ks_config.early_out_phase_growth, see there */

/* voxblox::TsdfIntegratorBase::Config + kimera SemanticIntegratorBase::SemanticConfig
 * (semantic_integrator_base.h:68-87) + layer geometry + device sizing, as one POD.
 * The first block of fields is laid out exactly like the oracle's ko_config so tests can
 * fill both from one dict. */
typedef struct ks_config {
  float voxel_size;
  int32_t voxels_per_side;          /* host Layer block edge (8, 16 or 32); device tiles are always 8^3 */
  float truncation_distance;
  float max_weight;
  float min_ray_length_m;
  float max_ray_length_m;
  int32_t voxel_carving_enabled;
  int32_t use_const_weight;
  int32_t allow_clear;
  int32_t use_weight_dropoff;
  int32_t use_sparsity_compensation_factor;
  float sparsity_compensation_factor;
  int32_t enable_anti_grazing;
  float start_voxel_subsampling_factor;
  int32_t max_consecutive_ray_collisions;
  int32_t clear_checks_every_n_frames;
  int32_t integration_order_mode;
  int32_t integrator_threads;       /* ignored on the GPU (kept for config compatibility) */
  int32_t method;
  int32_t bundle_order;             /* KS_BUNDLE_ORDER_* (merged only) */
  float semantic_measurement_probability;
  int32_t color_mode;
  int32_t n_dynamic_labels;
  uint8_t dynamic_labels[32];
  uint8_t label_rgba[256][4];       /* label -> colour (SemanticLabel2Color::semantic_label_to_color_map_) */
  /* fast integrator with the early-out enabled (max_consecutive_ray_collisions below the ray length):
   * the reference's loop (semantic_tsdf_integrator_fast.cpp:110-122) is serial by construction — ray k
   * stops on the marks rays 1..k-1 left in the approximate set.
   * 0 (default) or KS_EARLY_OUT_EXACT: the reference's SERIAL result itself (what integrator_threads = 1 produces, bit
   *    for bit, including the ApproxHashSet's zero-initialised slots that "contain" hash 0).  The serial result is the
   *    unique fixed point of "how far does every ray get against the marks of the rays before it"; the GPU reaches it
   *    with an event-driven iteration on the device (csrc/ks_k_exact.h: the seed's marks sorted once, then only rays
   *    whose inputs changed are re-evaluated; no host read in the loop), pipelined like every other mode when
   *    clear_checks_every_n_frames = 1 (with a larger value a frame's marks are inputs of the next frame's loop:
   *    pipeline_frames is then treated as 0).  Long rays (more than 400 voxels: 2 cm voxels / 10 m rays): whole-ray marks
   *    sorted once, then sweeps along the chains of the integration order on the device; one frame at a time
   *    (pipeline_frames treated as 0), the host reads two words per dozen sweeps.
   * 16 .. 4096: the ORDERED-PHASE schedule alone (restated for the CPU in oracle/ks_oracle.cpp, against
   *    which it is bit-exact): integration positions are cut into phases whose length grows by this factor (in
   *    1/16ths) — 32 = doubling, 16 = one generation (one integration position per chain = per group of the integration order) per phase.  Deterministic for every value, a few
   *    launches cheaper per frame than the exact mode, and NOT the reference's map (touched-voxel Jaccard against the serial order, 640x480 / 5 cm, "mixed" order in upstream's
   *    form: 0.9998 for 16, 0.976 for 24, 0.962 for 32, 0.916 for 64 — DESIGN.md §3.2; tests/test_early_out_fidelity.py asserts >= 0.93 for 32): a throughput option for callers who accept that. */
  int32_t early_out_phase_growth;
  /* ---- device sizing ---- */
  int32_t device_id;                /* HIP device ordinal */
  uint32_t max_tiles;               /* INITIAL capacity of the 8^3-voxel tile pool (64 KiB each); the pool is doubled
                                       between frames whenever more than half of it is in use (the reference
                                       allocates blocks without a cap, semantic_integrator_base.cpp:205-254).
                                       KS_ERR_POOL_FULL remains for a single frame that needs more than the free half. */
  uint32_t max_points;              /* largest cloud per call (buffers grow on demand if exceeded) */
  /* 0 (default): an integrate call returns with its frame fully enqueued and its own statistics.
   * 1 .. 16: frame pipelining for streams of frames (bag replay): the value is how many calls the second
   *    half of a frame (pair sort + voxel update) lags behind.  A call enqueues stages A and B of its frame
   *    (points .. early-out phases and pair emission) and finishes the frame `pipeline_frames` calls back; the
   *    one host wait of a frame then overlaps GPU work of later frames, and stage B of up to four consecutive
   *    frames runs concurrently: on four streams (values below 8), or — values 8 .. 15 — as one batched launch sequence
   *    per four frames (every kernel of stage B covers the batch), or — value 16 — per eight frames.  Larger values
   *    keep the host further ahead at the price of that many frames of latency (and of frame slots: 12 up to 8, 24 above).
   *    The statistics a call returns are those of the frames completed since statistics were last
   *    returned (summed if several), i.e. they lag by `pipeline_frames` calls; so do KS_ERR_LABEL_RANGE /
   *    pool errors.  Every other entry point (queries, download, export, ks_synchronize, ks_flush)
   *    completes the outstanding frames first, so the map they see is the same as without
   *    pipelining. */
  int32_t pipeline_frames;
} ks_config;

typedef struct ks_frame_stats {
  uint64_t n_points;
  uint64_t n_valid_points;
  uint64_t n_rays_cast;        /* fast: points surviving start-voxel dedup; merged: bundles */
  uint64_t n_voxel_updates;    /* (ray, voxel) pairs applied = reference updateTsdfVoxel+updateSemanticVoxel calls */
  uint64_t n_blocks_allocated; /* new 8^3 device tiles this frame */
} ks_frame_stats;

/* Accumulated HIP-event timings per pipeline stage (enabled with ks_profile_enable). */
enum {
  KS_STAGE_POINTS = 0,   /* per-point validity/transform/keys */
  KS_STAGE_SORT_POINTS,  /* dedup / bundling sort */
  KS_STAGE_RAYS,         /* dedup decision or bundle merge */
  KS_STAGE_MARCH,        /* ONE DDA walk per ray: tile allocation, observed-set early-out, (voxel, ray) pair emission,
                            counter snapshot (k_march + k_publish) */
  KS_STAGE_EMIT,         /* start of the tail: initialisation of the tiles the march allocated (the name dates from a
                            two-pass march; kept for ABI stability) */
  KS_STAGE_SORT_PAIRS,   /* group pairs by voxel in ray order */
  KS_STAGE_APPLY,        /* k_apply: per-voxel TSDF + semantic log-likelihood update (runs < 32 updates) */
  KS_STAGE_APPLY_LONG,   /* k_apply_long: one wavefront per voxel with >= 32 updates */
  KS_STAGE_COUNT
};
typedef struct ks_profile {
  double ms[KS_STAGE_COUNT];       /* summed over profiled frames */
  uint64_t launches[KS_STAGE_COUNT];
  uint64_t frames;
  uint64_t updates;                /* voxel updates over profiled frames */
  uint64_t points;
  double apply_kernel_ms;          /* k_apply dispatch begin->end (hipExtLaunchKernel events), summed */
  uint64_t apply_kernel_launches;  /* ... over this many timed launches */
  uint64_t apply_kernel_updates;   /* ... which performed this many voxel updates */
  double host_ms;                  /* wall time spent inside the integrate calls (enqueue + wait) */
  double host_wait_ms;             /* ... of which blocked on the per-frame counter snapshot */
} ks_profile;

typedef struct ks_ctx ks_ctx;

int ks_default_config(ks_config* cfg);
int ks_create(const ks_config* cfg, ks_ctx** out);
void ks_destroy(ks_ctx* ctx);
const char* ks_last_error(ks_ctx* ctx); /* ctx may be NULL: returns the last create-time error */

/* colour -> label map (SemanticLabel2Color::color_to_semantic_label_, color.cpp:57-66);
 * keys are RGBA with the alpha the CSV holds.  Lookups force alpha to 255 like
 * semantic_tsdf_integrator_fast.cpp:157 / merged.cpp:87; unknown colours map to label 0
 * (color.cpp:72-81). */
int ks_set_color_to_label(ks_ctx* ctx, const uint8_t* rgba_keys, const uint8_t* labels, size_t n);

/* vxb::TsdfIntegratorBase::integratePointCloud(T_G_C, points_C, colors, freespace)
 * (override at semantic_tsdf_integrator_fast.h:82-86 / merged.h:70-73) and the label-aware
 * overload merged.h:82-86.  Host pointers.  T_G_C = {qw,qx,qy,qz,tx,ty,tz}.
 * labels == NULL -> labels are derived from rgba through the colour map (the reference's
 * serial host loop, fast.cpp:150-158).  rgba == NULL -> colours are (0,0,0,0) (what the
 * reference's merged colour overload effectively integrates, merged.cpp:70,92-93).
 * Lifetime of the host buffers: pageable memory is staged before the call returns and may be reused at once; page-locked
 * memory (ks_host_alloc, hipHostMalloc, hipHostRegister) is read by the copy engine and may still be in use when a call of
 * a PIPELINED context returns (pipeline_frames = 0: the call completes its frame, the buffer is free): leave it unchanged
 * until `pipeline_frames` further integrate calls have returned (the call that finishes a frame has waited for it), or
 * until a call that completes outstanding work (ks_synchronize, ks_flush, any query) — i.e. rotate pipeline_frames + 1
 * buffers. */
int ks_integrate_points(ks_ctx* ctx, const float T_G_C[7], const float* xyz, const uint8_t* rgba,
                        const uint8_t* labels, size_t n, int freespace, ks_frame_stats* stats);
/* Same, but xyz/rgba/labels are DEVICE pointers already resident in HBM (bench timed region). */
int ks_integrate_points_device(ks_ctx* ctx, const float T_G_C[7], const float* d_xyz, const uint8_t* d_rgba,
                               const uint8_t* d_labels, size_t n, int freespace, ks_frame_stats* stats);

/* Depth + label image entry: the step BEFORE the hot path fused into the
 * GPU frontend.  Replaces PointCloudFromDepth::convert
 * (kimera_semantics_ros/include/kimera_semantics_ros/depth_map_to_pointcloud.h:213-275) plus the
 * colour->label host loop: back-projects with K = {fx, fy, cx, cy}, drops invalid pixels in image
 * order, then integrates exactly as ks_integrate_points would on the resulting cloud.
 * depth_fmt 0 = float32 metres (invalid: non-finite), 1 = uint16 millimetres (invalid: 0).
 * label_img (u8 per pixel) is preferred; with label_img == NULL the rgba8 segmentation image is
 * decoded through the colour map.  Host pointers / device pointers respectively.  The host-pointer entry counts the valid
 * pixels itself while its copies are in flight (no read-back, no host wait; buffer lifetime as for ks_integrate_points);
 * the device-pointer entry reads the compacted count back (4 bytes, one stream synchronisation) before it enqueues the frame. */
int ks_integrate_depth(ks_ctx* ctx, const float T_G_C[7], const void* depth, int depth_fmt, const uint8_t* label_img,
                       const uint8_t* rgba_img, int width, int height, const float K[4], int freespace,
                       ks_frame_stats* stats);
int ks_integrate_depth_device(ks_ctx* ctx, const float T_G_C[7], const void* d_depth, int depth_fmt,
                              const uint8_t* d_label_img, const uint8_t* d_rgba_img, int width, int height,
                              const float K[4], int freespace, ks_frame_stats* stats);

/* Layer views (host Layer<TsdfVoxel> / Layer<SemanticVoxel> contract, block edge = voxels_per_side). */
int ks_num_blocks(ks_ctx* ctx, size_t* n);
int ks_get_block_indices(ks_ctx* ctx, int32_t* out_xyz, size_t cap, size_t* n); /* sorted (x,y,z) */
/* Blocks touched since the last call with reset=1 (Block::updated(), semantic_integrator_base.cpp:248). */
int ks_get_updated_block_indices(ks_ctx* ctx, int32_t* out_xyz, size_t cap, size_t* n, int reset);
/* tsdf_out: n * vps^3 * 12 B {f32 distance, f32 weight, u8 rgba[4]};
 * sem_out:  n * vps^3 * 92 B {u8 label, 3 pad, f32 priors[21], u8 rgba[4]} (semantic_voxel.h:14-27).
 * Either may be NULL.  Absent blocks yield default-constructed voxels. */
int ks_download_blocks(ks_ctx* ctx, const int32_t* idx_xyz, size_t n, void* tsdf_out, void* sem_out);
/* Inverse of ks_download_blocks: seed / overwrite n host-layout blocks in the GPU map (same record
 * layouts; either array may be NULL to leave that half untouched).  This is how a map the host
 * already holds — the layers handed to the integrator constructor
 * (semantic_integrator_base.cpp:92-96, filled e.g. by TsdfServer::loadMap) — reaches the GPU.
 * Block indices must be distinct.  Errors: KS_ERR_POOL_FULL, KS_ERR_INDEX_RANGE. */
int ks_upload_blocks(ks_ctx* ctx, const int32_t* idx_xyz, size_t n, const void* tsdf_in, const void* sem_in);
/* Voxel-level sync for the strict drop-in mode (host Layers current after every integratePointCloud, the
 * contract behind Block::updated(), semantic_integrator_base.cpp:248): only the voxels the integrator has
 * written since the previous call travel, as KS_VOXEL_RECORD_BYTES-byte records
 *   { int32 block_x, block_y, block_z; uint32 linear_index (x + vps*(y + vps*z)); TsdfVoxel 12 B; SemanticVoxel 92 B }
 * with the voxels of one 8^3 device tile contiguous (consecutive records mostly share their block).
 * ks_count_updated_voxels sizes the buffer; ks_download_updated_voxels fills it (a page-locked buffer from
 * ks_host_alloc makes the copy run at link rate) and clears the marks.  Voxels merged in by ks_merge_tiles_device /
 * ks_reduce are reported too; voxels written by ks_upload_blocks are not (the host has them).
 * The voxel-level sync and the block-level sync (ks_get_updated_block_indices) share the per-tile "updated" flag:
 * a host uses ONE of the two (either call consumes the flag the other one reads). */
#define KS_VOXEL_RECORD_BYTES 120
/* The records of one device tile form a run that lies inside ONE host block: with the run list the host
 * finds its blocks without reading the records (runs may be NULL). */
typedef struct ks_voxel_run {
  int32_t block[3];
  uint32_t first; /* index of the run's first record */
  uint32_t count;
} ks_voxel_run;
int ks_count_updated_voxels(ks_ctx* ctx, size_t* n_records, size_t* n_runs /* may be NULL */);
int ks_download_updated_voxels(ks_ctx* ctx, void* out, size_t cap_records, size_t* n_records, ks_voxel_run* runs,
                               size_t cap_runs, size_t* n_runs);
/* Page-locked host memory for the buffers handed to ks_download_blocks / ks_upload_blocks /
 * ks_integrate_points (transfers from pageable memory go through a staging copy and run at a
 * fraction of the link rate).  ks_host_alloc returns NULL on failure. */
void* ks_host_alloc(size_t bytes);
void ks_host_free(void* p);

/* ---- multi-GPU exchange (new functionality: the reference is single-process) ----
 * The map is a set of 8^3-voxel tiles; a tile travels as its packed 63-bit key plus a raw
 * 64 KiB record block (512 voxels x 128 B).  ks_get_tile_keys lists the resident tiles in slot
 * order; ks_export_tiles_device gathers the tiles at the given slots into a DEVICE buffer
 * (n x 65536 B); ks_merge_tiles_device merges n incoming tiles (HOST keys, DEVICE payload) into
 * the resident map: weight-averaged distance/colour and summed weight (Voxblox's layer-merge
 * rule), additive class log-likelihoods, then argmax + colour.  Keys may repeat within one
 * call (tiles of the same key received from several ranks): they are folded in array order, in
 * one kernel launch, so the result is deterministic.  ks_clear empties the map. */
#define KS_TILE_BYTES 65536
int ks_get_tile_keys(ks_ctx* ctx, uint64_t* out, size_t cap, size_t* n);
int ks_export_tiles_device(ks_ctx* ctx, const uint32_t* slots, size_t n, void* d_payload);
int ks_merge_tiles_device(ks_ctx* ctx, const uint64_t* keys, size_t n, const void* d_payload);
int ks_clear(ks_ctx* ctx);
/* Empties the MAP only: frames in flight are completed first, then every tile goes; the integrator's own state — the two
 * approximate sets of `fast`, their offsets, the clear_checks_every_n_frames counter — stays as the frames so far left it.
 * This is what voxblox::TsdfServer::clear() does to the reference's integrator (nothing); integration/server.patch calls
 * it, then uploads the semantic layer that survives clear() in the reference. */
int ks_clear_voxels(ks_ctx* ctx);
/* Resets the tiles at the given slots to the empty state (a rank that has handed tiles to their owner keeps
 * them as empty deltas). */
int ks_reset_tiles(ks_ctx* ctx, const uint32_t* slots, size_t n);
/* Owner rank of a tile: splitmix64(key) % world. */
int ks_tile_owner(uint64_t tile_key, int world);
/* The frame-sharded path's ONE exchange step, for a C/C++ host: every rank of the RCCL
 * communicator calls it after integrating its share of a batch of frames.  rccl_comm is the caller's
 * ncclComm_t (one rank per GPU; librccl is loaded on first use, KS_RCCL_LIB overrides its path).  Tiles
 * touched since the previous reduce travel to their owner rank (all peers at once: one grouped send/recv for
 * the keys, one for the raw 64 KiB records, over xGMI); the owner folds them into its map in ascending
 * source-rank order (deterministic; weight-averaged TSDF, additive class log-likelihoods, argmax + colour);
 * the sender's copies start over as empty deltas, so the call can be repeated batch after batch without
 * counting anything twice.  Afterwards rank r holds the authoritative state of the tiles it owns.
 * COLLECTIVE: every rank calls it (a rank that returns early on a local error leaves its peers waiting, as with any
 * RCCL collective).  Steady state: the dirty-tile lists are built on the device, the exchange buffers are kept and
 * only grow; the host reads the world x world count matrix and the received keys (8 bytes per tile). */
typedef struct ks_reduce_stats {
  uint64_t tiles_sent, tiles_received, tiles_local, bytes_sent;
} ks_reduce_stats;
int ks_reduce(ks_ctx* ctx, void* rccl_comm, int rank, int world, ks_reduce_stats* stats);

/* EXACT frame-sharded integration of `fast` — one ROUND: `world` consecutive frames, frame first_frame + r marched by rank r.
 * COLLECTIVE: every rank calls it once per round (a rank without a frame passes n = 0), rounds in frame order.
 * `marcher` casts this rank's frame — stage A, the early-out, the emission, with the approximate sets' offsets of that GLOBAL
 * frame number — against a slot numbering of its own and evaluates the state-independent half of every update; each update
 * travels to the rank that owns its voxel's tile (ks_tile_owner) as 20 bytes { tile key << 9 | voxel in tile, info byte <<
 * 24 | integration position, sdf, update weight }; `owner` applies the frames of the round in frame order.  The tiles a rank
 * owns are then, bit for bit, what ONE context integrating all frames in order holds for them (new: the reference is a
 * single process; replaces merging per-rank maps with ks_reduce, which is a different arithmetic — SURVEY.md par. 8e).
 * Both contexts: method fast, colours from the labels, pipeline_frames = 0, "mixed" order, clear_checks_every_n_frames = 1 (or the
 * early-out off); same configuration on every rank; `marcher` is used for nothing else (its map stays empty of data).
 * origin_voxel_touched: a frame of the round updated the voxel whose index hashes to 0 (the world origin) — the one place
 * where the reference's approximate sets couple frames (their zero-initialised slots "contain" hash 0), i.e. where the
 * result may differ from the sequential one.  world = 1 needs no communicator. */
typedef struct ks_round_stats {
  uint64_t updates_marched, updates_applied, bytes_sent, origin_voxel_touched, rays_cast;
} ks_round_stats;
int ks_integrate_round_exact(ks_ctx* marcher, ks_ctx* owner, void* rccl_comm, int rank, int world, uint64_t first_frame,
                             const float T_G_C[7], const float* xyz, const uint8_t* rgba, const uint8_t* labels, size_t n,
                             int freespace_points, ks_round_stats* stats);

/* Diagnostics (used by tests): stable LSD radix sort of n HOST keys (key_bits = 32 or 64, bits
 * [0,end_bit)) and optional u32 payload with the library's own GPU sort. */
int ks_debug_radix_sort(ks_ctx* ctx, void* keys, uint32_t* vals, size_t n, int key_bits, unsigned end_bit);

int ks_synchronize(ks_ctx* ctx);
void* ks_stream(ks_ctx* ctx); /* the hipStream_t that reads the caller's device inputs (stage A; with
                                * pipeline_frames the later stages run on further internal streams) */
/* Finish the frames a pipelined context still holds (no-op otherwise); stats = theirs, summed. */
int ks_flush(ks_ctx* ctx, ks_frame_stats* stats);
/* level 0: off; 1: events around every stage and every k_apply dispatch (costs ~50 us of stream
 * bubbles per frame); 2: only the k_apply dispatch of every 4th frame is timed (a few us/frame).
 * Events are resolved lazily, never by a host wait inside a frame. */
int ks_profile_enable(ks_ctx* ctx, int level);
/* Exact early-out contexts: frames integrated and fix-point rounds run so far (either may be NULL). */
int ks_early_out_iterations(ks_ctx* ctx, uint64_t* frames, uint64_t* iterations);
/* ... and out[0] = frames, out[1] = rounds, out[2] = frames that fell back to the host-driven loop (buffers that had to
 * grow, round limit), out[3] = 1 if the event-driven loop is in use, out[4] = 1 if frames are pipelined (returns KS_OK). */
int ks_early_out_stats(ks_ctx* ctx, uint64_t out[5]);
int ks_profile_get(ks_ctx* ctx, ks_profile* out, int reset);
/* The voxel update's handling of the runs of more than 1024 updates (the voxels next to the sensor) since the context was
 * created: out[0] = runs whose class sums went through the integer-sum path (csrc/ks_k_apply_xl.h), out[1] = runs the serial
 * kernel took, out[2] = 64-update chunks on the first path, out[3] = of them, replayed update by update.  Completes the frames
 * in flight first. */
int ks_update_stats(ks_ctx* ctx, uint64_t out[4]);
/* How the context pipelines: out[0] = frames of lag in effect (0: one frame at a time — what ks_create made of
 * ks_config.pipeline_frames), out[1] = frame slots, out[2] = frames per stage-B batch, out[3] = march streams. */
int ks_pipeline_shape(ks_ctx* ctx, int32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* KS_HIP_H_ */
