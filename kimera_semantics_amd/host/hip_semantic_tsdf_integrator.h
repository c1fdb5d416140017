// kimera::HipSemanticTsdfIntegrator — the MI355X integrator behind the reference's plugin
// surface.  Same bases, same constructor arguments and the same virtual as
//   kimera::FastSemanticTsdfIntegrator   (kimera_semantics/include/kimera_semantics/semantic_tsdf_integrator_fast.h:63-86)
//   kimera::MergedSemanticTsdfIntegrator (kimera_semantics/include/kimera_semantics/semantic_tsdf_integrator_merged.h:56-86)
// so SemanticTsdfServer (kimera_semantics_ros/src/semantic_tsdf_server.cpp:71-78) can hold it
// through std::unique_ptr<vxb::TsdfIntegratorBase> unchanged.  All integration work happens in
// libks_hip.so (include/ks_hip.h); this class only marshals arguments and keeps the host
// Layers consistent.
#pragma once
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <voxblox/integrator/tsdf_integrator.h>

#include "kimera_types.h"
#include "ks_hip.h"

namespace kimera {

class HipSemanticTsdfIntegrator : public vxb::TsdfIntegratorBase, public SemanticIntegratorBase {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  enum class Method : int { kFast = KS_METHOD_FAST, kMerged = KS_METHOD_MERGED };

  /// When the host Layers are refreshed from the GPU map.
  enum class SyncPolicy {
    kEveryFrame,  ///< strict drop-in: Layers are current when integratePointCloud returns
    kOnDemand     ///< call syncLayers() before meshing / saving (e.g. from the mesh timer)
  };

  struct DeviceOptions {
    int device_id = 0;
    uint32_t max_tiles = 1u << 16;   ///< 8^3-voxel tiles (49.7 KB each)
    uint32_t max_points = 1u << 20;
    SyncPolicy sync_policy = SyncPolicy::kEveryFrame;
    /// ks_config.pipeline_frames (0 .. 16; 16 = batches of eight frames, twice the frame slots): how many calls the second half of a frame may lag behind.  The context is
    /// created able to pipeline whatever the policy: under kEveryFrame every call completes its frame anyway (the layer sync
    /// does), and a server that switches to kOnDemand AFTER the factory handed the integrator out (integration/server.patch)
    /// gets overlapping frames without the integrator being rebuilt.  0: never pipeline.
    int pipeline_frames = 8;
    /// syncLayers() moves only the VOXELS written since the previous sync (ks_download_updated_voxels: 120-byte
    /// records scattered into the host blocks) instead of whole updated blocks (ks_download_blocks).
    bool voxel_sync = true;
    /// `fast` with the early-out enabled (ks_config.early_out_phase_growth): 0 = the map the reference produces at
    /// integrator_threads = 1 (its deterministic case), bit for bit — the default, whatever Config::integrator_threads
    /// says (with more threads the reference itself is racy: any of its results is within its own spread of this
    /// one); 16..4096 = the ordered-phase schedule alone (a few launches cheaper, not the reference's map).
    int early_out_phase_growth = 0;
    /// integration_order_mode "mixed": -1 (default) = the permutation probeMixedOrder() reads from the Voxblox this build
    /// links (LOG(FATAL) if it is neither known form); KS_ORDER_MIXED / KS_ORDER_MIXED_1024_GROUPS force one.
    int mixed_order = -1;
  };

  /// The permutation vxb::ThreadSafeIndexFactory::get("mixed", ...) of this build hands out: KS_ORDER_MIXED,
  /// KS_ORDER_MIXED_1024_GROUPS, -1 (neither), -2 (built against the interface-only stand-in headers: no Voxblox to ask).
  static int probeMixedOrder();

  HipSemanticTsdfIntegrator(Method method, const Config& config, const SemanticConfig& semantic_config,
                            vxb::Layer<vxb::TsdfVoxel>* tsdf_layer, vxb::Layer<SemanticVoxel>* semantic_layer,
                            const DeviceOptions& options);
  HipSemanticTsdfIntegrator(Method method, const Config& config, const SemanticConfig& semantic_config,
                            vxb::Layer<vxb::TsdfVoxel>* tsdf_layer, vxb::Layer<SemanticVoxel>* semantic_layer);
  ~HipSemanticTsdfIntegrator() override;

  /// The virtual the server calls; labels are decoded from the colours on the GPU through the
  /// colour->label table (the reference's serial host loop, fast.cpp:150-158 / merged.cpp:73-88).
  void integratePointCloud(const vxb::Transformation& T_G_C, const vxb::Pointcloud& points_C,
                           const vxb::Colors& colors, const bool freespace_points = false) override;

  /// Label-aware overload (MergedSemanticTsdfIntegrator, merged.h:82-86).
  void integratePointCloud(const vxb::Transformation& T_G_C, const vxb::Pointcloud& points_C,
                           const HashableColors& colors, const SemanticLabels& semantic_labels,
                           const bool freespace_points = false);

  /// Copies every block touched since the last call into the host Layers (allocating blocks
  /// as needed) and sets Block::updated(), as semantic_integrator_base.cpp:248 does.
  void syncLayers();
  void syncLayersByBlock();  ///< the whole-block variant (DeviceOptions::voxel_sync = false)
  /// Host layers -> GPU map (every allocated block).  Called by the constructor when the layers it
  /// is handed are not empty (a map loaded with TsdfServer::loadMap), so the integrator continues
  /// from the map the host holds exactly like the CPU integrators do.
  void uploadLayers();

  /// Switch between the strict and the on-demand policy at run time (integration/server.patch: a server that calls
  /// syncLayers() where it reads the Layers selects kOnDemand right after the factory handed the integrator out).
  void setSyncPolicy(SyncPolicy policy) { options_.sync_policy = policy; }
  /// Empties the GPU map.  keep_integrator_state: only the voxels go, the approximate sets and frame counters of `fast` stay
  /// (ks_clear_voxels) — vxb::TsdfServer::clear() removes the TSDF blocks and leaves the integrator and the semantic layer
  /// alone: integration/server.patch syncs, lets the base class clear, empties the GPU map this way and uploads what survived.
  void clearDeviceMap(bool keep_integrator_state = false);
  SyncPolicy syncPolicy() const { return options_.sync_policy; }

  ks_ctx* context() { return ctx_; }
  const ks_frame_stats& lastFrameStats() const { return last_stats_; }

 private:
  void check(int rc, const char* what) const;
  ks_ctx* ctx_ = nullptr;
  Method method_;
  DeviceOptions options_;
  ks_frame_stats last_stats_{};
  vxb::Layer<SemanticVoxel>* semantic_layer_ptr_;
  // page-locked staging for layer transfers (ks_host_alloc); grows on demand
  struct Staging {
    uint8_t* p = nullptr;
    size_t cap = 0;
    uint8_t* reserve(size_t bytes);
    ~Staging();
  };
  std::vector<int32_t> idx_buf_;
  Staging tsdf_buf_, sem_buf_, vox_buf_, run_buf_;

  /// Persistent workers for the scatter of downloaded voxels into the host layers (thread start-up per frame
  /// costs as much as the scatter itself).
  class Workers {
   public:
    ~Workers();
    /// fn(begin, end) over [0, n) cut into `parts` ranges; returns when all ranges are done.
    void run(size_t n, size_t parts, const std::function<void(size_t, size_t)>& fn);

   private:
    void loop(size_t id);
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t, size_t)>* fn_ = nullptr;
    size_t n_ = 0, parts_ = 0, next_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
  };
  Workers workers_;
};

/// Same shape as kimera::SemanticTsdfIntegratorFactory
/// (kimera_semantics/include/kimera_semantics/semantic_tsdf_integrator_factory.h:57-100):
/// "fast" / "merged" (and the explicit aliases "fast_hip" / "merged_hip") return the HIP
/// integrator as std::unique_ptr<vxb::TsdfIntegratorBase>; anything else is LOG(FATAL).
class HipSemanticTsdfIntegratorFactory {
 public:
  static std::unique_ptr<vxb::TsdfIntegratorBase> create(
      const std::string& integrator_type_name, const vxb::TsdfIntegratorBase::Config& config,
      const SemanticIntegratorBase::SemanticConfig& semantic_config, vxb::Layer<vxb::TsdfVoxel>* tsdf_layer,
      vxb::Layer<SemanticVoxel>* semantic_layer,
      const HipSemanticTsdfIntegrator::DeviceOptions& options = HipSemanticTsdfIntegrator::DeviceOptions());

  /// The enum overload of the reference factory (semantic_tsdf_integrator_factory.h:84-93): kMerged = 0, kFast = 1,
  /// plus the values integration/factory.patch adds for the explicit names, kMergedHip = 2, kFastHip = 3.  Taken
  /// as the enum's underlying int so that this header does not depend on the (patched) factory header;
  /// anything else is LOG(FATAL), like the reference's default branch (semantic_tsdf_integrator_factory.cpp:82-86).
  static std::unique_ptr<vxb::TsdfIntegratorBase> create(
      int integrator_type, const vxb::TsdfIntegratorBase::Config& config,
      const SemanticIntegratorBase::SemanticConfig& semantic_config, vxb::Layer<vxb::TsdfVoxel>* tsdf_layer,
      vxb::Layer<SemanticVoxel>* semantic_layer,
      const HipSemanticTsdfIntegrator::DeviceOptions& options = HipSemanticTsdfIntegrator::DeviceOptions());
  template <typename Enum, typename = typename std::enable_if<std::is_enum<Enum>::value>::type>
  static std::unique_ptr<vxb::TsdfIntegratorBase> create(
      const Enum& integrator_type, const vxb::TsdfIntegratorBase::Config& config,
      const SemanticIntegratorBase::SemanticConfig& semantic_config, vxb::Layer<vxb::TsdfVoxel>* tsdf_layer,
      vxb::Layer<SemanticVoxel>* semantic_layer,
      const HipSemanticTsdfIntegrator::DeviceOptions& options = HipSemanticTsdfIntegrator::DeviceOptions()) {
    return create(static_cast<int>(integrator_type), config, semantic_config, tsdf_layer, semantic_layer, options);
  }

 private:
  HipSemanticTsdfIntegratorFactory() = default;
};

}  // namespace kimera
