// ORACLE-side restatement of voxblox/integrator/tsdf_integrator.{h,cc} (test infrastructure):
// the CPU TsdfIntegratorBase with the primitives the Kimera-Semantics integrators call, and
// MergedTsdfIntegrator::bundleRays.  Used ONLY to compile the real reference sources from
// /root/reference into oracle/_ref; never part of the product.  Voxblox is un-vendored and
// un-pinned upstream; this follows its published algorithm.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "voxblox/core/block_hash.h"
#include "voxblox/core/color.h"
#include "voxblox/core/common.h"
#include "voxblox/core/layer.h"
#include "voxblox/core/voxel.h"
#include "voxblox/integrator/integrator_utils.h"
#include "voxblox/utils/approx_hash_array.h"
#include "voxblox/utils/timing.h"

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

  TsdfIntegratorBase(const Config& config, Layer<TsdfVoxel>* layer) : config_(config) { setLayer(layer); }
  virtual ~TsdfIntegratorBase() = default;

  virtual void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                                   const bool freespace_points = false) = 0;
  const Config& getConfig() const { return config_; }

  void setLayer(Layer<TsdfVoxel>* layer) {
    CHECK_NOTNULL(layer);
    layer_ = layer;
    voxel_size_ = layer_->voxel_size();
    block_size_ = layer_->block_size();
    voxels_per_side_ = layer_->voxels_per_side();
    voxel_size_inv_ = 1.0 / voxel_size_;
    block_size_inv_ = 1.0 / block_size_;
    voxels_per_side_inv_ = 1.0 / voxels_per_side_;
  }

 protected:
  inline bool isPointValid(const Point& point_C, const bool freespace_point, bool* is_clearing) const {
    const FloatingPoint ray_distance = point_C.norm();
    if (ray_distance < config_.min_ray_length_m) {
      return false;
    } else if (ray_distance > config_.max_ray_length_m) {
      if (config_.allow_clear || freespace_point) {
        *is_clearing = true;
        return true;
      } else {
        return false;
      }
    } else {
      *is_clearing = freespace_point;
      return true;
    }
  }

  TsdfVoxel* allocateStorageAndGetVoxelPtr(const GlobalIndex& global_voxel_idx, Block<TsdfVoxel>::Ptr* last_block,
                                           BlockIndex* last_block_idx) {
    const BlockIndex block_idx = getBlockIndexFromGlobalVoxelIndex(global_voxel_idx, voxels_per_side_inv_);
    if ((block_idx != *last_block_idx) || (*last_block == nullptr)) {
      *last_block = layer_->getBlockPtrByIndex(block_idx);
      *last_block_idx = block_idx;
    }
    if (*last_block == nullptr) {
      std::lock_guard<std::mutex> lock(temp_block_mutex_);
      typename Layer<TsdfVoxel>::BlockHashMap::iterator it = temp_block_map_.find(block_idx);
      if (it != temp_block_map_.end()) {
        *last_block = it->second;
      } else {
        auto insert_status = temp_block_map_.emplace(
            block_idx, std::make_shared<Block<TsdfVoxel>>(voxels_per_side_, voxel_size_,
                                                          getOriginPointFromGridIndex(block_idx, block_size_)));
        *last_block = insert_status.first->second;
      }
    }
    (*last_block)->updated() = true;
    const VoxelIndex local_voxel_idx = getLocalFromGlobalVoxelIndex(global_voxel_idx, voxels_per_side_);
    return &((*last_block)->getVoxelByVoxelIndex(local_voxel_idx));
  }

  void updateLayerWithStoredBlocks() {
    for (const std::pair<const BlockIndex, Block<TsdfVoxel>::Ptr>& temp_block_pair : temp_block_map_)
      layer_->insertBlock(temp_block_pair);
    temp_block_map_.clear();
  }

  float computeDistance(const Point& origin, const Point& point_G, const Point& voxel_center) const {
    const Point v_voxel_origin = voxel_center - origin;
    const Point v_point_origin = point_G - origin;
    const FloatingPoint dist_G = v_point_origin.norm();
    const FloatingPoint dist_G_V = v_voxel_origin.dot(v_point_origin) / dist_G;
    const float sdf = static_cast<float>(dist_G - dist_G_V);
    return sdf;
  }

  void updateTsdfVoxel(const Point& origin, const Point& point_G, const GlobalIndex& global_voxel_idx, const Color& color,
                       const float weight, TsdfVoxel* tsdf_voxel) {
    const Point voxel_center = getCenterPointFromGridIndex(global_voxel_idx, voxel_size_);
    const float sdf = computeDistance(origin, point_G, voxel_center);
    float updated_weight = weight;
    const FloatingPoint dropoff_epsilon = voxel_size_;
    if (config_.use_weight_dropoff && sdf < -dropoff_epsilon) {
      updated_weight = weight * (config_.default_truncation_distance + sdf) /
                       (config_.default_truncation_distance - dropoff_epsilon);
      updated_weight = std::max(updated_weight, 0.0f);
    }
    if (config_.use_sparsity_compensation_factor) {
      if (std::abs(sdf) < config_.default_truncation_distance) updated_weight *= config_.sparsity_compensation_factor;
    }
    std::lock_guard<std::mutex> lock(mutexes_.get(global_voxel_idx));
    const float new_weight = tsdf_voxel->weight + updated_weight;
    if (new_weight < kFloatEpsilon) return;
    const float new_sdf = (sdf * updated_weight + tsdf_voxel->distance * tsdf_voxel->weight) / new_weight;
    if (std::abs(sdf) < config_.default_truncation_distance) {
      tsdf_voxel->color = Color::blendTwoColors(tsdf_voxel->color, tsdf_voxel->weight, color, updated_weight);
    }
    tsdf_voxel->distance = (new_sdf > 0.0) ? std::min(config_.default_truncation_distance, new_sdf)
                                           : std::max(-config_.default_truncation_distance, new_sdf);
    tsdf_voxel->weight = std::min(config_.max_weight, new_weight);
  }

  float getVoxelWeight(const Point& point_C) const {
    if (config_.use_const_weight) return 1.0f;
    const FloatingPoint dist_z = std::abs(point_C.z());
    if (dist_z > kEpsilon) return 1.0f / (dist_z * dist_z);
    return 0.0f;
  }

  Config config_;
  Layer<TsdfVoxel>* layer_;
  FloatingPoint voxel_size_;
  size_t voxels_per_side_;
  FloatingPoint block_size_;
  FloatingPoint voxel_size_inv_;
  FloatingPoint voxels_per_side_inv_;
  FloatingPoint block_size_inv_;
  std::mutex temp_block_mutex_;
  Layer<TsdfVoxel>::BlockHashMap temp_block_map_;
  ApproxHashArray<12, std::mutex, GlobalIndex, LongIndexHash> mutexes_;
};

class MergedTsdfIntegrator : public TsdfIntegratorBase {
 public:
  MergedTsdfIntegrator(const Config& config, Layer<TsdfVoxel>* layer) : TsdfIntegratorBase(config, layer) {}

 protected:
  void bundleRays(const Transformation& T_G_C, const Pointcloud& points_C, const bool freespace_points,
                  ThreadSafeIndex* index_getter, LongIndexHashMapType<AlignedVector<size_t>>::type* voxel_map,
                  LongIndexHashMapType<AlignedVector<size_t>>::type* clear_map) {
    size_t point_idx;
    while (index_getter->getNextIndex(&point_idx)) {
      const Point& point_C = points_C[point_idx];
      bool is_clearing;
      if (!isPointValid(point_C, freespace_points, &is_clearing)) continue;
      const Point point_G = T_G_C * point_C;
      GlobalIndex voxel_index = getGridIndexFromPoint<GlobalIndex>(point_G, voxel_size_inv_);
      if (is_clearing) (*clear_map)[voxel_index].push_back(point_idx);
      else (*voxel_map)[voxel_index].push_back(point_idx);
    }
  }
};

}  // namespace voxblox
