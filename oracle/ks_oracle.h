/*
 * ks_oracle.h — C API of the CPU ORACLE for the semantic TSDF integration hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (kimera_semantics_amd/, include/ks_hip.h) never includes, links or calls it.
 *
 * What it is: a dependency-free C++17 restatement of
 *   - kimera::FastSemanticTsdfIntegrator   (kimera_semantics/src/semantic_tsdf_integrator_fast.cpp:57-199)
 *   - kimera::MergedSemanticTsdfIntegrator (kimera_semantics/src/semantic_tsdf_integrator_merged.cpp:65-329)
 *   - kimera::SemanticIntegratorBase       (kimera_semantics/src/semantic_integrator_base.cpp:93-380)
 * and of the Voxblox / minkindr / Eigen primitives those call (un-vendored, un-pinned
 * third-party code that is NOT under /root/reference: ethz-asl/voxblox @ default branch,
 * ethz-asl/minkindr @ default branch, Eigen 3.3 — see
 * install/kimera_semantics_https.rosinstall:7-9,19-21,34-36 and SURVEY.md Appendix A).
 *
 * Parity pinning status:
 *   - Kimera half (the three files above): PINNED against the real reference sources
 *     compiled from /root/reference into oracle/_ref (see oracle/Makefile, oracle/ref/),
 *     checked bit-exactly by tests/test_oracle_vs_ref.py.
 *   - Voxblox/minkindr/Eigen half: PARITY UNPINNED — the reference has no tests and no
 *     golden vectors and those libraries are absent; the arithmetic
 *     here is a restatement of their published algorithm, anchored on the reference's
 *     call sites.  Hand-derived known-answer tests live in tests/test_oracle_kat.py.
 */
#ifndef KS_ORACLE_H_
#define KS_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_NUM_LABELS 21

enum { KO_METHOD_FAST = 0, KO_METHOD_MERGED = 1 };
enum { KO_COLOR_MODE_COLOR = 0, KO_COLOR_MODE_SEMANTIC = 1, KO_COLOR_MODE_SEMANTIC_PROBABILITY = 2 };
/* "mixed" = Voxblox's MixedThreadSafeIndex (un-pinned upstream, absent from /root/reference).  Two readings exist:
 *   KO_ORDER_MIXED             idx = (s % q) * 1024 + s / q, q = N / 1024   (upstream as published: number_of_groups_ = N /
 *                              step_size_, group_num = s % number_of_groups_, position_in_group = s / number_of_groups_)
 *   KO_ORDER_MIXED_1024_GROUPS idx = (s % 1024) * q + s / 1024              (rounds 1-4 of this repository)
 * positions >= q * 1024 map to themselves in both. */
enum { KO_ORDER_MIXED = 0, KO_ORDER_SORTED = 1, KO_ORDER_MIXED_1024_GROUPS = 2 };
/* merged: order in which bundles are integrated.
 * REFERENCE = iteration order of std::unordered_map<GlobalIndex, vector, LongIndexHash>
 *             (what the reference does, semantic_tsdf_integrator_merged.cpp:210-231,
 *             under libstdc++);
 * CANONICAL = ascending position of each bundle's first point in the index-getter
 *             order (first-insertion order) — the order the GPU path reproduces. */
enum { KO_BUNDLE_ORDER_REFERENCE = 0, KO_BUNDLE_ORDER_CANONICAL = 1 };

/* Mirrors voxblox::TsdfIntegratorBase::Config + kimera SemanticConfig
 * (semantic_integrator_base.h:68-87) + the layer geometry. */
typedef struct ko_config {
  float voxel_size;                 /* tsdf_voxel_size */
  int32_t voxels_per_side;          /* power of two */
  float truncation_distance;        /* default_truncation_distance */
  float max_weight;                 /* 1e4 */
  float min_ray_length_m;           /* 0.1 */
  float max_ray_length_m;           /* 5.0 */
  int32_t voxel_carving_enabled;    /* 1 */
  int32_t use_const_weight;         /* 0 */
  int32_t allow_clear;              /* 1 */
  int32_t use_weight_dropoff;       /* 1 */
  int32_t use_sparsity_compensation_factor; /* 0 */
  float sparsity_compensation_factor;       /* 1.0 */
  int32_t enable_anti_grazing;      /* 0 */
  float start_voxel_subsampling_factor;     /* 2.0 */
  int32_t max_consecutive_ray_collisions;   /* 2 */
  int32_t clear_checks_every_n_frames;      /* 1 */
  int32_t integration_order_mode;   /* KO_ORDER_* */
  int32_t integrator_threads;       /* 1 = deterministic */
  int32_t method;                   /* KO_METHOD_* */
  int32_t bundle_order;             /* KO_BUNDLE_ORDER_* (merged only) */
  float semantic_measurement_probability;   /* 0.9 struct default; launches use 0.8 */
  int32_t color_mode;               /* KO_COLOR_MODE_* */
  int32_t n_dynamic_labels;
  uint8_t dynamic_labels[32];
  uint8_t label_rgba[256][4];       /* SemanticLabel2Color::semantic_label_to_color_map_ flattened;
                                       absent ids -> (0,0,0,0) (color.cpp:89-92) */
  /* fast, early-out enabled: 0 = the reference's serial order (one ray after the other,
   * semantic_tsdf_integrator_fast.cpp:110-122).  >= 16 = the GPU path's ORDERED-PHASE schedule, restated
   * here so the HIP kernels can be checked bit for bit against it (see integrate_fast_phased in
   * ks_oracle.cpp for the definition): phase boundaries grow by this factor / 16 (16 = one generation — one
   * integration position per chain, see ko_mixed_chains — per phase, 32 = doubling). */
  int32_t early_out_phase_growth;
} ko_config;

typedef struct ko_frame_stats {
  uint64_t n_points;
  uint64_t n_valid_points;   /* passed isPointValid (+ label filter in fast) */
  uint64_t n_rays_cast;      /* fast: survived start-voxel dedup; merged: bundles (normal+clearing) */
  uint64_t n_voxel_updates;  /* (ray-or-bundle, voxel) pairs for which updateTsdfVoxel+updateSemanticVoxel ran */
  uint64_t n_blocks_allocated; /* new TSDF blocks this frame */
} ko_frame_stats;

typedef struct ko_ctx ko_ctx;

void ko_default_config(ko_config* cfg);
int ko_create(const ko_config* cfg, ko_ctx** out);
void ko_destroy(ko_ctx* ctx);
const char* ko_last_error(ko_ctx* ctx);

/* T_G_C = {qw, qx, qy, qz, tx, ty, tz}.  labels == NULL is an error (the colour->label
 * map lives in the host adapter / test harness).  rgba may be NULL (treated as (0,0,0,0)).
 * Returns 0, or <0 on error (label >= 21 -> -2, mirroring CHECK_LT at
 * semantic_tsdf_integrator_fast.cpp:134 / merged.cpp:278). */
int ko_integrate_points(ko_ctx* ctx, const float T_G_C[7], const float* xyz,
                        const uint8_t* rgba, const uint8_t* labels, size_t n,
                        int freespace, ko_frame_stats* stats);

size_t ko_num_blocks(ko_ctx* ctx);                       /* TSDF layer */
size_t ko_num_semantic_blocks(ko_ctx* ctx);
void ko_get_block_indices(ko_ctx* ctx, int32_t* out_xyz); /* n*3, sorted lexicographically (x,y,z) */
void ko_get_semantic_block_indices(ko_ctx* ctx, int32_t* out_xyz);
/* Copy one block: tsdf_out = vps^3 * 12 B (float distance, float weight, u8 rgba[4]);
 * sem_out = vps^3 * 92 B (u8 label, 3 pad, float priors[21], u8 rgba[4]).
 * Voxel linear index x + vps*(y + vps*z).  Returns 0, or 1 if the block is absent
 * (outputs then hold default-constructed voxels). */
int ko_get_block(ko_ctx* ctx, const int32_t idx[3], void* tsdf_out, void* sem_out);

/* ---- pure functions exposed for known-answer tests ---- */
void ko_transform_point(const float T_G_C[7], const float p[3], float out[3]);
void ko_grid_index_from_point(const float p[3], float grid_size_inv, int64_t out[3]);
/* Full (un-terminated) RayCaster voxel list; returns count (<= cap written). */
size_t ko_cast_ray(const float origin[3], const float point_G[3], int is_clearing,
                   int carving, float max_ray_length_m, float voxel_size_inv,
                   float truncation, int cast_from_origin, int64_t* out_xyz, size_t cap);
void ko_log_likelihood(float p_match, float out_21x21_rowmajor[KO_NUM_LABELS * KO_NUM_LABELS]);
uint32_t ko_long_index_hash(const int64_t idx[3]);
size_t ko_mixed_index(size_t sequential_idx, size_t num_elements, int mode /* KO_ORDER_MIXED* */);
uint32_t ko_mixed_chains(size_t num_elements, int mode); /* chains of the ordered-phase schedule */
/* One updateTsdfVoxel step on a caller-held voxel (distance, weight, rgba). */
void ko_update_tsdf_voxel(const ko_config* cfg, const float origin[3], const float point_G[3],
                          const int64_t voxel_idx[3], const uint8_t rgba[4], float weight,
                          float* distance, float* voxel_weight, uint8_t voxel_rgba[4]);
void ko_blend_two_colors(const uint8_t c1[4], float w1, const uint8_t c2[4], float w2, uint8_t out[4]);
void ko_rainbow_color_map(double h, uint8_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* KS_ORACLE_H_ */
