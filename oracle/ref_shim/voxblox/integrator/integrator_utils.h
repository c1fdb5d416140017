// ORACLE-side restatement of voxblox/integrator/integrator_utils.{h,cc}: ThreadSafeIndex and
// RayCaster (test infrastructure; Voxblox itself is not in /root/reference — SURVEY.md A.6, A.8).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "voxblox/core/block_hash.h"
#include "voxblox/core/common.h"

namespace voxblox {

class ThreadSafeIndex {
 public:
  explicit ThreadSafeIndex(size_t number_of_points) : atomic_idx_(0), number_of_points_(number_of_points) {}
  virtual ~ThreadSafeIndex() = default;
  bool getNextIndex(size_t* idx) {
    size_t sequential_idx = atomic_idx_.fetch_add(1);
    if (sequential_idx >= number_of_points_) return false;
    *idx = getNextIndexImpl(sequential_idx);
    return true;
  }
  void reset() { atomic_idx_.store(0); }

 protected:
  virtual size_t getNextIndexImpl(size_t sequential_idx) = 0;
  std::atomic<size_t> atomic_idx_;
  const size_t number_of_points_;
};

// "mixed" order.  Voxblox is NOT in /root/reference (un-vendored, un-pinned), so which permutation upstream's
// MixedThreadSafeIndex::getNextIndexImpl produces cannot be read here.  Both readings are implemented:
//   form 0 (default; upstream as published, voxblox/integrator/integrator_utils.cc):
//        number_of_groups_ = N / step_size_, step_size_ = 1024;
//        group_num = s % number_of_groups_; position_in_group = s / number_of_groups_;
//        idx = group_num * step_size_ + position_in_group        (consecutive positions are 1024 points apart)
//   form 1 (what rounds 1-4 of this repository assumed): 1024 groups of N / 1024 points,
//        idx = (s % 1024) * (N / 1024) + s / 1024                (consecutive positions are N / 1024 points apart)
// The product's adapter probes the ThreadSafeIndexFactory it is built against and selects the matching form
// (kimera_semantics_amd/host/hip_semantic_tsdf_integrator.cpp: probe_mixed_order); this switch is how the tests
// put either behaviour behind the real Kimera sources.
inline int& shim_mixed_order_form() {
  static int form = 0;
  return form;
}

class MixedThreadSafeIndex : public ThreadSafeIndex {
 public:
  explicit MixedThreadSafeIndex(size_t number_of_points)
      : ThreadSafeIndex(number_of_points), form_(shim_mixed_order_form()), number_of_groups_(number_of_points / step_size_) {}

 protected:
  size_t getNextIndexImpl(size_t sequential_idx) override {
    if (number_of_groups_ * step_size_ <= sequential_idx) return sequential_idx;
    if (form_ == 1) return (sequential_idx % step_size_) * number_of_groups_ + sequential_idx / step_size_;
    const size_t group_num = sequential_idx % number_of_groups_;
    const size_t position_in_group = sequential_idx / number_of_groups_;
    return group_num * step_size_ + position_in_group;
  }

 private:
  static constexpr size_t step_size_ = 1024;
  const int form_;
  const size_t number_of_groups_;
};

class SortedThreadSafeIndex : public ThreadSafeIndex {
 public:
  explicit SortedThreadSafeIndex(const Pointcloud& points_C) : ThreadSafeIndex(points_C.size()) {
    indices_and_squared_norms_.reserve(points_C.size());
    size_t idx = 0;
    for (const Point& point_C : points_C) indices_and_squared_norms_.emplace_back(idx++, point_C.squaredNorm());
    // upstream: std::sort (ties unspecified); pinned here to a stable order
    std::stable_sort(indices_and_squared_norms_.begin(), indices_and_squared_norms_.end(),
                     [](const std::pair<size_t, double>& a, const std::pair<size_t, double>& b) { return a.second < b.second; });
  }

 protected:
  size_t getNextIndexImpl(size_t sequential_idx) override { return indices_and_squared_norms_[sequential_idx].first; }

 private:
  std::vector<std::pair<size_t, double>> indices_and_squared_norms_;
};

class ThreadSafeIndexFactory {
 public:
  static ThreadSafeIndex* get(const std::string& mode, const Pointcloud& points_C) {
    if (mode == "mixed") return new MixedThreadSafeIndex(points_C.size());
    if (mode == "sorted") return new SortedThreadSafeIndex(points_C);
    LOG(FATAL) << "Unknown integration order mode: '" << mode << "'!";
    return nullptr;
  }
};

class RayCaster {
 public:
  RayCaster(const Point& origin, const Point& point_G, const bool is_clearing_ray, const bool voxel_carving_enabled,
            const FloatingPoint max_ray_length_m, const FloatingPoint voxel_size_inv,
            const FloatingPoint truncation_distance, const bool cast_from_origin = true) {
    const Ray unit_ray = (point_G - origin).normalized();
    Point ray_start, ray_end;
    if (is_clearing_ray) {
      FloatingPoint ray_length = (point_G - origin).norm();
      ray_length = std::min(std::max(ray_length - truncation_distance, static_cast<FloatingPoint>(0.0)), max_ray_length_m);
      ray_end = origin + unit_ray * ray_length;
      ray_start = voxel_carving_enabled ? origin : ray_end;
    } else {
      ray_end = point_G + unit_ray * truncation_distance;
      ray_start = voxel_carving_enabled ? origin : (point_G - unit_ray * truncation_distance);
    }
    const Point start_scaled = ray_start * voxel_size_inv;
    const Point end_scaled = ray_end * voxel_size_inv;
    if (cast_from_origin) setupRayCaster(start_scaled, end_scaled);
    else setupRayCaster(end_scaled, start_scaled);
  }
  RayCaster(const Point& start_scaled, const Point& end_scaled) { setupRayCaster(start_scaled, end_scaled); }

  bool nextRayIndex(GlobalIndex* ray_index) {
    if (current_step_++ > ray_length_in_steps_) return false;
    *ray_index = curr_index_;
    int t_min_idx;
    t_to_next_boundary_.minCoeff(&t_min_idx);
    curr_index_[t_min_idx] += ray_step_signs_[t_min_idx];
    t_to_next_boundary_[t_min_idx] += t_step_size_[t_min_idx];
    return true;
  }

 private:
  void setupRayCaster(const Point& start_scaled, const Point& end_scaled) {
    if (std::isnan(start_scaled.x()) || std::isnan(start_scaled.y()) || std::isnan(start_scaled.z()) ||
        std::isnan(end_scaled.x()) || std::isnan(end_scaled.y()) || std::isnan(end_scaled.z())) {
      ray_length_in_steps_ = 0;
      current_step_ = 0;
      return;
    }
    curr_index_ = getGridIndexFromPoint<GlobalIndex>(start_scaled);
    const GlobalIndex end_index = getGridIndexFromPoint<GlobalIndex>(end_scaled);
    const GlobalIndex diff_index = end_index - curr_index_;
    current_step_ = 0;
    ray_length_in_steps_ = std::abs(diff_index.x()) + std::abs(diff_index.y()) + std::abs(diff_index.z());
    const Ray ray_scaled = end_scaled - start_scaled;
    ray_step_signs_ = AnyIndex(signum(ray_scaled.x()), signum(ray_scaled.y()), signum(ray_scaled.z()));
    const AnyIndex corrected_step(std::max(0, ray_step_signs_.x()), std::max(0, ray_step_signs_.y()),
                                  std::max(0, ray_step_signs_.z()));
    const Point start_scaled_shifted = start_scaled - curr_index_.cast<FloatingPoint>();
    Ray distance_to_boundaries(corrected_step.cast<FloatingPoint>() - start_scaled_shifted);
    t_to_next_boundary_ = Ray((std::abs(ray_scaled.x()) < 0.0) ? 2.0 : distance_to_boundaries.x() / ray_scaled.x(),
                              (std::abs(ray_scaled.y()) < 0.0) ? 2.0 : distance_to_boundaries.y() / ray_scaled.y(),
                              (std::abs(ray_scaled.z()) < 0.0) ? 2.0 : distance_to_boundaries.z() / ray_scaled.z());
    t_step_size_ = Ray((std::abs(ray_scaled.x()) < 0.0) ? 2.0 : ray_step_signs_.x() / ray_scaled.x(),
                       (std::abs(ray_scaled.y()) < 0.0) ? 2.0 : ray_step_signs_.y() / ray_scaled.y(),
                       (std::abs(ray_scaled.z()) < 0.0) ? 2.0 : ray_step_signs_.z() / ray_scaled.z());
  }

  Ray t_to_next_boundary_;
  GlobalIndex curr_index_;
  AnyIndex ray_step_signs_;
  Ray t_step_size_;
  uint ray_length_in_steps_ = 0;
  uint current_step_ = 0;
};

}  // namespace voxblox
