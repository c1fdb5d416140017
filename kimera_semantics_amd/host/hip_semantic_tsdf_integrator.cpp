#include "hip_semantic_tsdf_integrator.h"

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

#include <voxblox/utils/timing.h>
// Builds against a real Voxblox tree (or the reference-side shim, which has the same class) can ASK the integration order
// instead of assuming it: see probeMixedOrder below.
#if defined(__has_include)
#if __has_include(<voxblox/integrator/integrator_utils.h>)
#include <voxblox/integrator/integrator_utils.h>
#define KS_HAVE_VOXBLOX_THREAD_SAFE_INDEX 1
#endif
#endif

namespace kimera {

// Which permutation does vxb::ThreadSafeIndexFactory::get("mixed", ...) — the call the CPU integrators make at
// [K:src/semantic_tsdf_integrator_fast.cpp:172-174] and [K:src/semantic_tsdf_integrator_merged.cpp:115-117] — hand out in THIS
// build?  Voxblox is an un-pinned upstream dependency of the reference; the order decides which ray meets which ray's
// early-out marks and the f32 summation order of every voxel, i.e. every bit of the map.  The real index is driven over a
// cloud of 5 * 1024 + 7 points (groups, group size and the unpermuted tail all differ between the two known forms) and
// compared with both closed forms of include/ks_hip.h.
int HipSemanticTsdfIntegrator::probeMixedOrder() {
#ifdef KS_HAVE_VOXBLOX_THREAD_SAFE_INDEX
  const size_t n = 5 * 1024 + 7, q = n / 1024;
  const vxb::Pointcloud cloud(n, vxb::Point(0.0f, 0.0f, 1.0f));
  std::unique_ptr<vxb::ThreadSafeIndex> index_getter(vxb::ThreadSafeIndexFactory::get("mixed", cloud));
  bool upstream = true, by_1024 = true;
  size_t idx = 0, s = 0;
  for (; index_getter->getNextIndex(&idx); ++s) {
    if (s >= n) return -1;
    const size_t a = s < q * 1024 ? (s % q) * 1024 + s / q : s;
    const size_t b = s < q * 1024 ? (s % 1024) * q + s / 1024 : s;
    upstream = upstream && idx == a;
    by_1024 = by_1024 && idx == b;
  }
  if (s != n) return -1;
  return upstream ? KS_ORDER_MIXED : by_1024 ? KS_ORDER_MIXED_1024_GROUPS : -1;
#else
  return -2;  // no Voxblox in this build (the interface-only stand-in headers): nothing to ask
#endif
}

namespace {
ks_config makeConfig(HipSemanticTsdfIntegrator::Method method, const vxb::TsdfIntegratorBase::Config& c,
                     const SemanticIntegratorBase::SemanticConfig& sc, const vxb::Layer<vxb::TsdfVoxel>& layer,
                     const HipSemanticTsdfIntegrator::DeviceOptions& o) {
  ks_config k;
  ks_default_config(&k);
  k.voxel_size = layer.voxel_size();
  k.voxels_per_side = static_cast<int32_t>(layer.voxels_per_side());
  k.truncation_distance = c.default_truncation_distance;
  k.max_weight = c.max_weight;
  k.min_ray_length_m = c.min_ray_length_m;
  k.max_ray_length_m = c.max_ray_length_m;
  k.voxel_carving_enabled = c.voxel_carving_enabled;
  k.use_const_weight = c.use_const_weight;
  k.allow_clear = c.allow_clear;
  k.use_weight_dropoff = c.use_weight_dropoff;
  k.use_sparsity_compensation_factor = c.use_sparsity_compensation_factor;
  k.sparsity_compensation_factor = c.sparsity_compensation_factor;
  k.enable_anti_grazing = c.enable_anti_grazing;
  k.start_voxel_subsampling_factor = c.start_voxel_subsampling_factor;
  k.max_consecutive_ray_collisions = c.max_consecutive_ray_collisions;
  k.clear_checks_every_n_frames = c.clear_checks_every_n_frames;
  if (c.integration_order_mode == "mixed") {
    static const int probed = HipSemanticTsdfIntegrator::probeMixedOrder();
    if (probed == -1 && o.mixed_order < 0)  // (an explicit DeviceOptions::mixed_order is the caller's own answer)
      LOG(FATAL) << "vxb::ThreadSafeIndexFactory::get(\"mixed\", ...) of this build produces neither of the two permutations the "
                    "GPU integrator implements (include/ks_hip.h: KS_ORDER_MIXED, KS_ORDER_MIXED_1024_GROUPS): its results would "
                    "not be the CPU integrators'.";
    // (-2: built without Voxblox, against the stand-in headers — the form upstream publishes; o.mixed_order overrides)
    k.integration_order_mode = o.mixed_order >= 0 ? o.mixed_order : probed >= 0 ? probed : KS_ORDER_MIXED;
  } else if (c.integration_order_mode == "sorted") k.integration_order_mode = KS_ORDER_SORTED;
  else LOG(FATAL) << "Unknown integration order mode: '" << c.integration_order_mode << "'!";
  k.integrator_threads = static_cast<int32_t>(c.integrator_threads);
  k.method = static_cast<int32_t>(method);
  k.semantic_measurement_probability = sc.semantic_measurement_probability_;
  k.color_mode = static_cast<int32_t>(sc.color_mode);
  CHECK_LE(sc.dynamic_labels_.size(), 32u);
  k.n_dynamic_labels = static_cast<int32_t>(sc.dynamic_labels_.size());
  for (size_t i = 0; i < sc.dynamic_labels_.size(); ++i) k.dynamic_labels[i] = sc.dynamic_labels_[i];
  std::memset(k.label_rgba, 0, sizeof(k.label_rgba));
  CHECK(sc.semantic_label_to_color_);
  for (const auto& kv : sc.semantic_label_to_color_->semantic_label_to_color_map_) {
    k.label_rgba[kv.first][0] = kv.second.r;
    k.label_rgba[kv.first][1] = kv.second.g;
    k.label_rgba[kv.first][2] = kv.second.b;
    k.label_rgba[kv.first][3] = kv.second.a;
  }
  k.early_out_phase_growth = o.early_out_phase_growth;
  k.device_id = o.device_id;
  k.max_tiles = o.max_tiles;
  k.max_points = o.max_points;
  k.pipeline_frames = std::min(16, std::max(0, o.pipeline_frames));  // 0..16 (kMaxLag): above 8 = batches of eight frames
  return k;
}
}  // namespace

HipSemanticTsdfIntegrator::HipSemanticTsdfIntegrator(Method method, const Config& config,
                                                     const SemanticConfig& semantic_config,
                                                     vxb::Layer<vxb::TsdfVoxel>* tsdf_layer,
                                                     vxb::Layer<SemanticVoxel>* semantic_layer)
    : HipSemanticTsdfIntegrator(method, config, semantic_config, tsdf_layer, semantic_layer, DeviceOptions()) {}

HipSemanticTsdfIntegrator::HipSemanticTsdfIntegrator(Method method, const Config& config,
                                                     const SemanticConfig& semantic_config,
                                                     vxb::Layer<vxb::TsdfVoxel>* tsdf_layer,
                                                     vxb::Layer<SemanticVoxel>* semantic_layer,
                                                     const DeviceOptions& options)
    : vxb::TsdfIntegratorBase(config, CHECK_NOTNULL(tsdf_layer)),
      SemanticIntegratorBase(semantic_config, CHECK_NOTNULL(semantic_layer)),
      method_(method),
      options_(options),
      semantic_layer_ptr_(semantic_layer) {
  CHECK_EQ(tsdf_layer->voxels_per_side(), semantic_layer->voxels_per_side());
  const ks_config k = makeConfig(method, config, semantic_config, *tsdf_layer, options);
  const int rc = ks_create(&k, &ctx_);
  // programmer errors abort like the reference's CHECKs (semantic_integrator_base.cpp:98-107)
  CHECK_EQ(rc, KS_OK) << "ks_create failed (" << rc << "): " << ks_last_error(nullptr);
  // colour -> label table for the colour-encoded clouds the server delivers
  std::vector<uint8_t> keys, labels;
  for (const auto& kv : semantic_config.semantic_label_to_color_->color_to_semantic_label_) {
    keys.push_back(kv.first.r);
    keys.push_back(kv.first.g);
    keys.push_back(kv.first.b);
    keys.push_back(kv.first.a);
    labels.push_back(kv.second);
  }
  check(ks_set_color_to_label(ctx_, keys.data(), labels.data(), labels.size()), "ks_set_color_to_label");
  if (tsdf_layer->getNumberOfAllocatedBlocks() > 0 || semantic_layer->getNumberOfAllocatedBlocks() > 0) uploadLayers();
}

HipSemanticTsdfIntegrator::~HipSemanticTsdfIntegrator() { ks_destroy(ctx_); }

void HipSemanticTsdfIntegrator::check(int rc, const char* what) const {
  // The reference has no status returns: data/programmer errors are CHECK/LOG(FATAL).
  if (rc != KS_OK) LOG(FATAL) << what << " failed (" << rc << "): " << ks_last_error(ctx_);
}

void HipSemanticTsdfIntegrator::integratePointCloud(const vxb::Transformation& T_G_C,
                                                    const vxb::Pointcloud& points_C, const vxb::Colors& colors,
                                                    const bool freespace_points) {
  CHECK_EQ(points_C.size(), colors.size());
  static_assert(sizeof(vxb::Point) == 12 && sizeof(vxb::Color) == 4, "cloud element layout");
  // the scope names the CPU integrators record, so the server's verbose timing print keeps its rows:
  // "integrate/fast" [K:src/semantic_tsdf_integrator_fast.cpp:160], "semantic_tsdf/integrate" +
  // "integrate/semantic_merged" [K:src/semantic_tsdf_integrator_merged.cpp:90-91,106]; the copy-back of
  // the touched blocks plays the part of "inserting_missed_blocks" [K:...fast.cpp:195, ...merged.cpp:193]
  vxb::timing::Timer outer_timer(method_ == Method::kFast ? "integrate/fast" : "semantic_tsdf/integrate");
  vxb::timing::Timer merged_timer("integrate/semantic_merged", /*construct_stopped=*/method_ == Method::kFast);
  const float T[7] = {T_G_C.qw(),          T_G_C.qvec().x(),        T_G_C.qvec().y(),       T_G_C.qvec().z(),
                      T_G_C.getPosition().x(), T_G_C.getPosition().y(), T_G_C.getPosition().z()};
  // merged: the reference's colour overload integrates default-constructed colours
  // (hash_colors is sized but never filled, semantic_tsdf_integrator_merged.cpp:70,92-93);
  // labels still come from the real colours.  The C ABI call below keeps both behaviours:
  // labels from rgba through the map; for merged the blended colour is irrelevant unless
  // ColorMode::kColor, where the reference blends zeros -> pass them through a second call path.
  const uint8_t* rgba = colors.empty() ? nullptr : reinterpret_cast<const uint8_t*>(colors.data());
  const float* xyz = points_C.empty() ? nullptr : reinterpret_cast<const float*>(points_C.data());
  if (method_ == Method::kMerged && semantic_config_.color_mode == ColorMode::kColor) {
    // decode labels on the host map once, then integrate with zero colours (exactly what the
    // reference computes in this corner)
    SemanticLabels labels(colors.size());
    for (size_t i = 0; i < colors.size(); ++i)
      labels[i] = semantic_config_.semantic_label_to_color_->getSemanticLabelFromColor(
          HashableColor(colors[i].r, colors[i].g, colors[i].b, 255u));
    check(ks_integrate_points(ctx_, T, xyz, nullptr, labels.data(), points_C.size(), freespace_points, &last_stats_),
          "ks_integrate_points");
  } else {
    // Config::max_integration_time_s [K:src/semantic_tsdf_integrator_fast.cpp:66-70] is a wall-clock budget a 0.3 ms GPU frame never
    // reaches; its one deterministic case is kept: a budget <= 0 lets no point into the fast integrator's loop (the
    // frame-level bookkeeping of the sets still happens: a call with zero points)
    const bool no_budget = method_ == Method::kFast && !(config_.max_integration_time_s > 0.0f);
    check(ks_integrate_points(ctx_, T, xyz, rgba, nullptr, no_budget ? 0 : points_C.size(), freespace_points, &last_stats_),
          "ks_integrate_points");
  }
  merged_timer.Stop();
  if (options_.sync_policy == SyncPolicy::kEveryFrame) {
    vxb::timing::Timer insertion_timer("inserting_missed_blocks");
    // (a context that can pipeline: the frame is completed here, and its statistics are this call's)
    check(ks_flush(ctx_, &last_stats_), "ks_flush");
    syncLayers();
  }
}

void HipSemanticTsdfIntegrator::integratePointCloud(const vxb::Transformation& T_G_C,
                                                    const vxb::Pointcloud& points_C, const HashableColors& colors,
                                                    const SemanticLabels& semantic_labels,
                                                    const bool freespace_points) {
  CHECK_EQ(points_C.size(), colors.size());
  CHECK_EQ(points_C.size(), semantic_labels.size());
  const float T[7] = {T_G_C.qw(),          T_G_C.qvec().x(),        T_G_C.qvec().y(),       T_G_C.qvec().z(),
                      T_G_C.getPosition().x(), T_G_C.getPosition().y(), T_G_C.getPosition().z()};
  static_assert(sizeof(HashableColor) == 4, "colour layout");
  vxb::timing::Timer integrate_timer("integrate/semantic_merged");
  check(ks_integrate_points(ctx_, T, points_C.empty() ? nullptr : reinterpret_cast<const float*>(points_C.data()),
                            colors.empty() ? nullptr : reinterpret_cast<const uint8_t*>(colors.data()),
                            semantic_labels.data(),
                            (method_ == Method::kFast && !(config_.max_integration_time_s > 0.0f)) ? 0 : points_C.size(),  // (budget <= 0: see above)
                            freespace_points, &last_stats_),
        "ks_integrate_points");
  integrate_timer.Stop();
  if (options_.sync_policy == SyncPolicy::kEveryFrame) {
    vxb::timing::Timer insertion_timer("inserting_missed_blocks");
    check(ks_flush(ctx_, &last_stats_), "ks_flush");
    syncLayers();
  }
}

void HipSemanticTsdfIntegrator::clearDeviceMap(bool keep_integrator_state) {
  if (keep_integrator_state) check(ks_clear_voxels(ctx_), "ks_clear_voxels");
  else check(ks_clear(ctx_), "ks_clear");
}

HipSemanticTsdfIntegrator::Workers::~Workers() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_work_.notify_all();
  for (auto& t : threads_) t.join();
}

void HipSemanticTsdfIntegrator::Workers::loop(size_t) {
  std::unique_lock<std::mutex> lk(mu_);
  uint64_t seen = 0;
  for (;;) {
    cv_work_.wait(lk, [&] { return stop_ || (epoch_ != seen && next_ < parts_); });
    if (stop_) return;
    while (next_ < parts_) {
      const size_t part = next_++;
      const size_t chunk = (n_ + parts_ - 1) / parts_;
      const size_t b = std::min(n_, part * chunk), e = std::min(n_, (part + 1) * chunk);
      const auto* fn = fn_;
      lk.unlock();
      (*fn)(b, e);
      lk.lock();
      if (--pending_ == 0) cv_done_.notify_all();
    }
    seen = epoch_;
  }
}

void HipSemanticTsdfIntegrator::Workers::run(size_t n, size_t parts, const std::function<void(size_t, size_t)>& fn) {
  if (parts <= 1) {
    fn(0, n);
    return;
  }
  std::unique_lock<std::mutex> lk(mu_);
  while (threads_.size() + 1 < parts) {
    const size_t id = threads_.size();
    threads_.emplace_back([this, id] { loop(id); });
  }
  fn_ = &fn;
  n_ = n;
  parts_ = parts;
  next_ = 0;
  pending_ = parts;
  ++epoch_;
  cv_work_.notify_all();
  // the caller works too
  while (next_ < parts_) {
    const size_t part = next_++;
    const size_t chunk = (n_ + parts_ - 1) / parts_;
    const size_t b = std::min(n_, part * chunk), e = std::min(n_, (part + 1) * chunk);
    lk.unlock();
    fn(b, e);
    lk.lock();
    --pending_;
  }
  cv_done_.wait(lk, [&] { return pending_ == 0; });
  fn_ = nullptr;
}

uint8_t* HipSemanticTsdfIntegrator::Staging::reserve(size_t bytes) {
  if (bytes > cap) {
    ks_host_free(p);
    cap = bytes + bytes / 4;
    p = static_cast<uint8_t*>(ks_host_alloc(cap));
    CHECK(p != nullptr) << "ks_host_alloc(" << cap << ") failed";
  }
  return p;
}
HipSemanticTsdfIntegrator::Staging::~Staging() { ks_host_free(p); }

namespace {
// The wire records of ks_download_blocks / ks_upload_blocks ARE the host voxel types when those
// are laid out as upstream's (TsdfVoxel 12 B: distance, weight, rgba; SemanticVoxel 92 B: label,
// 21 priors, rgba): a block is then one memcpy instead of a per-field loop.
constexpr bool kTsdfLayoutMatches = sizeof(vxb::TsdfVoxel) == 12 && offsetof(vxb::TsdfVoxel, weight) == 4 &&
                                    offsetof(vxb::TsdfVoxel, color) == 8 && sizeof(vxb::Color) == 4;
constexpr bool kSemLayoutMatches = sizeof(SemanticVoxel) == 92 && offsetof(SemanticVoxel, semantic_priors) == 4 &&
                                   offsetof(SemanticVoxel, color) == 88 && sizeof(HashableColor) == 4;
}  // namespace

void HipSemanticTsdfIntegrator::syncLayers() {
  if (!options_.voxel_sync) {
    syncLayersByBlock();
    return;
  }
  // Only the voxels the integrator wrote since the last sync travel (a frame touches ~1.7e5 voxels of ~150
  // blocks: 20 MB instead of 60-85 MB of whole blocks); a tile's voxels are contiguous in the buffer, so the
  // block lookup happens once per tile, not once per voxel.
  // The staging buffers keep the size of the largest sync so far: the download is tried with what is there,
  // and only a sync that outgrows it pays for a second attempt (the call reports the sizes it needs).
  size_t n = 0, n_runs = 0;
  vxb::timing::Timer t_dl("sync/download");
  int rc = ks_download_updated_voxels(ctx_, vox_buf_.p, vox_buf_.cap / KS_VOXEL_RECORD_BYTES, &n,
                                      reinterpret_cast<ks_voxel_run*>(run_buf_.p), run_buf_.cap / sizeof(ks_voxel_run), &n_runs);
  if (rc == KS_ERR_INVALID_ARG && (n * KS_VOXEL_RECORD_BYTES > vox_buf_.cap || n_runs * sizeof(ks_voxel_run) > run_buf_.cap)) {
    vox_buf_.reserve((n + n / 4) * KS_VOXEL_RECORD_BYTES);
    run_buf_.reserve((n_runs + n_runs / 4) * sizeof(ks_voxel_run));
    rc = ks_download_updated_voxels(ctx_, vox_buf_.p, vox_buf_.cap / KS_VOXEL_RECORD_BYTES, &n,
                                    reinterpret_cast<ks_voxel_run*>(run_buf_.p), run_buf_.cap / sizeof(ks_voxel_run), &n_runs);
  }
  check(rc, "ks_download_updated_voxels");
  if (n == 0) return;
  const uint8_t* buf = vox_buf_.p;
  const ks_voxel_run* runs = reinterpret_cast<const ks_voxel_run*>(run_buf_.p);
  t_dl.Stop();
  vxb::timing::Timer t_scatter("sync/scatter");
  // pass 1 (serial, one step per device tile): make sure the host blocks exist and are flagged
  for (size_t r = 0; r < n_runs; ++r) {
    if (runs[r].count == 0) continue;
    const vxb::BlockIndex idx(runs[r].block[0], runs[r].block[1], runs[r].block[2]);
    layer_->allocateBlockPtrByIndex(idx)->updated() = true;
    semantic_layer_ptr_->allocateBlockPtrByIndex(idx)->updated() = true;
  }
  // pass 2 (parallel over record ranges; distinct records are distinct voxels, block lookups are read-only)
  auto scatter = [this, buf](size_t begin, size_t end) {
    int32_t last[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    vxb::Block<vxb::TsdfVoxel>::Ptr tb;
    vxb::Block<SemanticVoxel>::Ptr sb;
    for (size_t i = begin; i < end; ++i) {
      const uint8_t* r = buf + i * KS_VOXEL_RECORD_BYTES;
      int32_t h[4];
      std::memcpy(h, r, 16);
      if (h[0] != last[0] || h[1] != last[1] || h[2] != last[2]) {
        const vxb::BlockIndex idx(h[0], h[1], h[2]);
        tb = layer_->getBlockPtrByIndex(idx);
        sb = semantic_layer_ptr_->getBlockPtrByIndex(idx);
        last[0] = h[0]; last[1] = h[1]; last[2] = h[2];
      }
      const size_t lin = static_cast<uint32_t>(h[3]);
      const uint8_t* t = r + 16;
      const uint8_t* sv_in = r + 28;
      vxb::TsdfVoxel& v = tb->getVoxelByLinearIndex(lin);
      if (kTsdfLayoutMatches) {
        std::memcpy(static_cast<void*>(&v), t, 12);
      } else {
        std::memcpy(&v.distance, t, 4);
        std::memcpy(&v.weight, t + 4, 4);
        v.color = vxb::Color(t[8], t[9], t[10], t[11]);
      }
      SemanticVoxel& sv = sb->getVoxelByLinearIndex(lin);
      if (kSemLayoutMatches) {
        std::memcpy(static_cast<void*>(&sv), sv_in, 92);
      } else {
        sv.semantic_label = sv_in[0];
        for (size_t l = 0; l < kTotalNumberOfLabels; ++l) {
          float p;
          std::memcpy(&p, sv_in + 4 + 4 * l, 4);
          sv.semantic_priors[l] = p;
        }
        sv.color = HashableColor(sv_in[88], sv_in[89], sv_in[90], sv_in[91]);
      }
    }
  };
  static const size_t kThreadsEnv = std::getenv("KS_SYNC_THREADS") ? std::strtoul(std::getenv("KS_SYNC_THREADS"), nullptr, 10) : 0;
  const size_t n_threads = kThreadsEnv ? kThreadsEnv
                                       : (n < 20000 ? 1 : std::min<size_t>(16, std::max<size_t>(1, std::thread::hardware_concurrency())));
  workers_.run(n, n_threads, scatter);
  t_scatter.Stop();
}

void HipSemanticTsdfIntegrator::syncLayersByBlock() {
  size_t n = 0;
  check(ks_get_updated_block_indices(ctx_, nullptr, 0, &n, 0), "ks_get_updated_block_indices");
  if (n == 0) return;
  idx_buf_.resize(3 * n);
  check(ks_get_updated_block_indices(ctx_, idx_buf_.data(), n, &n, 1), "ks_get_updated_block_indices");
  const size_t vps = layer_->voxels_per_side();
  const size_t nv = vps * vps * vps;
  uint8_t* tsdf = tsdf_buf_.reserve(n * nv * 12);
  uint8_t* sem = sem_buf_.reserve(n * nv * 92);
  check(ks_download_blocks(ctx_, idx_buf_.data(), n, tsdf, sem), "ks_download_blocks");
  for (size_t b = 0; b < n; ++b) {
    const vxb::BlockIndex idx(idx_buf_[3 * b], idx_buf_[3 * b + 1], idx_buf_[3 * b + 2]);
    auto tb = layer_->allocateBlockPtrByIndex(idx);
    auto sb = semantic_layer_ptr_->allocateBlockPtrByIndex(idx);
    const uint8_t* t = tsdf + b * nv * 12;
    const uint8_t* s = sem + b * nv * 92;
    if (kTsdfLayoutMatches) {
      std::memcpy(static_cast<void*>(&tb->getVoxelByLinearIndex(0)), t, nv * 12);
    } else {
      for (size_t i = 0; i < nv; ++i) {
        vxb::TsdfVoxel& v = tb->getVoxelByLinearIndex(i);
        std::memcpy(&v.distance, t + 12 * i, 4);
        std::memcpy(&v.weight, t + 12 * i + 4, 4);
        v.color = vxb::Color(t[12 * i + 8], t[12 * i + 9], t[12 * i + 10], t[12 * i + 11]);
      }
    }
    if (kSemLayoutMatches) {
      std::memcpy(static_cast<void*>(&sb->getVoxelByLinearIndex(0)), s, nv * 92);
    } else {
      for (size_t i = 0; i < nv; ++i) {
        SemanticVoxel& sv = sb->getVoxelByLinearIndex(i);
        const uint8_t* r = s + 92 * i;
        sv.semantic_label = r[0];
        for (size_t l = 0; l < kTotalNumberOfLabels; ++l) {
          float p;
          std::memcpy(&p, r + 4 + 4 * l, 4);
          sv.semantic_priors[l] = p;
        }
        sv.color = HashableColor(r[88], r[89], r[90], r[91]);
      }
    }
    tb->updated() = true;
    sb->updated() = true;
  }
}

void HipSemanticTsdfIntegrator::uploadLayers() {
  vxb::BlockIndexList blocks, sem_only;
  layer_->getAllAllocatedBlocks(&blocks);
  semantic_layer_ptr_->getAllAllocatedBlocks(&sem_only);
  for (const vxb::BlockIndex& idx : sem_only)
    if (!layer_->hasBlock(idx)) blocks.push_back(idx);
  const size_t n = blocks.size();
  if (n == 0) return;
  const size_t vps = layer_->voxels_per_side();
  const size_t nv = vps * vps * vps;
  idx_buf_.resize(3 * n);
  uint8_t* tsdf_stage = tsdf_buf_.reserve(n * nv * 12);
  uint8_t* sem_stage = sem_buf_.reserve(n * nv * 92);
  std::memset(tsdf_stage, 0, n * nv * 12);
  std::memset(sem_stage, 0, n * nv * 92);
  const vxb::TsdfVoxel default_tsdf;
  const SemanticVoxel default_sem;
  for (size_t b = 0; b < n; ++b) {
    const vxb::BlockIndex& idx = blocks[b];
    idx_buf_[3 * b] = idx.x();
    idx_buf_[3 * b + 1] = idx.y();
    idx_buf_[3 * b + 2] = idx.z();
    const auto tb = layer_->getBlockPtrByIndex(idx);
    const auto sb = semantic_layer_ptr_->getBlockPtrByIndex(idx);
    uint8_t* t = tsdf_stage + b * nv * 12;
    uint8_t* s = sem_stage + b * nv * 92;
    for (size_t i = 0; i < nv; ++i) {
      const vxb::TsdfVoxel& v = tb ? tb->getVoxelByLinearIndex(i) : default_tsdf;
      std::memcpy(t + 12 * i, &v.distance, 4);
      std::memcpy(t + 12 * i + 4, &v.weight, 4);
      t[12 * i + 8] = v.color.r;
      t[12 * i + 9] = v.color.g;
      t[12 * i + 10] = v.color.b;
      t[12 * i + 11] = v.color.a;
      const SemanticVoxel& sv = sb ? sb->getVoxelByLinearIndex(i) : default_sem;
      uint8_t* r = s + 92 * i;
      r[0] = sv.semantic_label;
      for (size_t l = 0; l < kTotalNumberOfLabels; ++l) {
        const float p = sv.semantic_priors[l];
        std::memcpy(r + 4 + 4 * l, &p, 4);
      }
      r[88] = sv.color.r;
      r[89] = sv.color.g;
      r[90] = sv.color.b;
      r[91] = sv.color.a;
    }
  }
  check(ks_upload_blocks(ctx_, idx_buf_.data(), n, tsdf_stage, sem_stage), "ks_upload_blocks");
  // what was just uploaded is not "updated by integration"
  size_t m = 0;
  check(ks_get_updated_block_indices(ctx_, nullptr, 0, &m, 1), "ks_get_updated_block_indices");
}

std::unique_ptr<vxb::TsdfIntegratorBase> HipSemanticTsdfIntegratorFactory::create(
    const std::string& integrator_type_name, const vxb::TsdfIntegratorBase::Config& config,
    const SemanticIntegratorBase::SemanticConfig& semantic_config, vxb::Layer<vxb::TsdfVoxel>* tsdf_layer,
    vxb::Layer<SemanticVoxel>* semantic_layer, const HipSemanticTsdfIntegrator::DeviceOptions& options) {
  CHECK(!integrator_type_name.empty());
  CHECK_NOTNULL(tsdf_layer);
  if (integrator_type_name == "fast" || integrator_type_name == "fast_hip") {
    return std::unique_ptr<vxb::TsdfIntegratorBase>(new HipSemanticTsdfIntegrator(
        HipSemanticTsdfIntegrator::Method::kFast, config, semantic_config, tsdf_layer, semantic_layer, options));
  }
  if (integrator_type_name == "merged" || integrator_type_name == "merged_hip") {
    return std::unique_ptr<vxb::TsdfIntegratorBase>(new HipSemanticTsdfIntegrator(
        HipSemanticTsdfIntegrator::Method::kMerged, config, semantic_config, tsdf_layer, semantic_layer, options));
  }
  LOG(FATAL) << "Unknown TSDF integrator type: " << integrator_type_name;
  return nullptr;
}

std::unique_ptr<vxb::TsdfIntegratorBase> HipSemanticTsdfIntegratorFactory::create(
    int integrator_type, const vxb::TsdfIntegratorBase::Config& config,
    const SemanticIntegratorBase::SemanticConfig& semantic_config, vxb::Layer<vxb::TsdfVoxel>* tsdf_layer,
    vxb::Layer<SemanticVoxel>* semantic_layer, const HipSemanticTsdfIntegrator::DeviceOptions& options) {
  CHECK_NOTNULL(tsdf_layer);
  switch (integrator_type) {
    case 1:  // SemanticTsdfIntegratorType::kFast
    case 3:  // kFastHip (integration/factory.patch)
      return create("fast", config, semantic_config, tsdf_layer, semantic_layer, options);
    case 0:  // kMerged
    case 2:  // kMergedHip
      return create("merged", config, semantic_config, tsdf_layer, semantic_layer, options);
    default:
      LOG(FATAL) << "Unknown Semantic/TSDF integrator type: " << integrator_type;
      break;
  }
  return nullptr;
}

}  // namespace kimera
