// INTERFACE-ONLY stand-in for voxblox/integrator/tsdf_integrator.h (product side).
// Carries TsdfIntegratorBase::Config and the virtual the server calls; it deliberately has
// NO CPU integration arithmetic — the product path has no CPU fallback.  (The oracle's
// reference build uses oracle/ref_shim/voxblox/integrator/tsdf_integrator.h instead.)
#pragma once
#include <cfloat>
#include <memory>
#include <string>
#include <thread>

#include "voxblox/core/color.h"
#include "voxblox/core/common.h"
#include "voxblox/core/layer.h"
#include "voxblox/core/voxel.h"

namespace voxblox {

class TsdfIntegratorBase {
 public:
  typedef std::shared_ptr<TsdfIntegratorBase> Ptr;

  struct Config {
    float default_truncation_distance = 0.1;
    float max_weight = 10000.0;
    bool voxel_carving_enabled = true;
    FloatingPoint min_ray_length_m = 0.1;
    FloatingPoint max_ray_length_m = 5.0;
    bool use_const_weight = false;
    bool allow_clear = true;
    bool use_weight_dropoff = true;
    bool use_sparsity_compensation_factor = false;
    float sparsity_compensation_factor = 1.0f;
    size_t integrator_threads = std::thread::hardware_concurrency();
    std::string integration_order_mode = "mixed";
    bool enable_anti_grazing = false;
    float start_voxel_subsampling_factor = 2.0f;
    int max_consecutive_ray_collisions = 2;
    int clear_checks_every_n_frames = 1;
    float max_integration_time_s = std::numeric_limits<float>::max();
  };

  TsdfIntegratorBase(const Config& config, Layer<TsdfVoxel>* layer) : config_(config), layer_(layer) { CHECK_NOTNULL(layer_); }
  virtual ~TsdfIntegratorBase() = default;

  virtual void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                                   const bool freespace_points = false) = 0;
  const Config& getConfig() const { return config_; }
  Layer<TsdfVoxel>* getLayer() { return layer_; }

 protected:
  Config config_;
  Layer<TsdfVoxel>* layer_;
};

}  // namespace voxblox
