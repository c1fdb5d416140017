// Type-level stand-in for voxblox/core/common.h (ethz-asl/voxblox is not installed here).
// Only what the Kimera-Semantics integrator boundary needs: scalar/index/point typedefs, the
// index helpers, and a rigid transform with the minkindr operations the hot path calls
// (getPosition(), operator*(Point)).  With real Voxblox on the include path this file is unused.
#pragma once
#include <cmath>
#include <cstdint>
#include <deque>
#include <list>
#include <memory>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <glog/logging.h>

namespace voxblox {

typedef float FloatingPoint;
typedef int IndexElement;
typedef int64_t LongIndexElement;

typedef Eigen::Matrix<FloatingPoint, 3, 1> Point;
typedef Eigen::Matrix<FloatingPoint, 3, 1> Ray;
typedef Eigen::Matrix<IndexElement, 3, 1> AnyIndex;
typedef AnyIndex VoxelIndex;
typedef AnyIndex BlockIndex;
typedef AnyIndex SignedIndex;
typedef Eigen::Matrix<LongIndexElement, 3, 1> LongIndex;
typedef LongIndex GlobalIndex;

template <typename Type>
using AlignedVector = std::vector<Type, Eigen::aligned_allocator<Type>>;
template <typename Type>
using AlignedDeque = std::deque<Type, Eigen::aligned_allocator<Type>>;

typedef AlignedVector<Point> Pointcloud;
typedef AlignedVector<GlobalIndex> LongIndexVector;
typedef AlignedVector<AnyIndex> IndexVector;
typedef IndexVector BlockIndexList;

constexpr FloatingPoint kEpsilon = 1e-6;
constexpr FloatingPoint kCoordinateEpsilon = 1e-6;
constexpr float kFloatEpsilon = 1e-6;

// kindr::minimal::QuatTransformationTemplate<float> surface used by the integrators:
// q (w, x, y, z) + translation; transform(p) = q.rotate(p) + t with Eigen's _transformVector.
class Transformation {
 public:
  Transformation() : w_(1.f), v_(0.f, 0.f, 0.f), t_(0.f, 0.f, 0.f) {}
  Transformation(float w, float x, float y, float z, const Point& t) : w_(w), v_(x, y, z), t_(t) {}
  const Point& getPosition() const { return t_; }
  float qw() const { return w_; }
  const Point& qvec() const { return v_; }
  Point operator*(const Point& p) const {
    auto cross = [](const Point& a, const Point& b) {
      return Point(a.y() * b.z() - a.z() * b.y(), a.z() * b.x() - a.x() * b.z(), a.x() * b.y() - a.y() * b.x());
    };
    Point uv = cross(v_, p);
    uv += uv;
    return (p + w_ * uv + cross(v_, uv)) + t_;
  }

 private:
  float w_;
  Point v_;
  Point t_;
};

template <typename IndexType>
inline IndexType getGridIndexFromPoint(const Point& point, const FloatingPoint grid_size_inv) {
  return IndexType(std::floor(point.x() * grid_size_inv + kCoordinateEpsilon),
                   std::floor(point.y() * grid_size_inv + kCoordinateEpsilon),
                   std::floor(point.z() * grid_size_inv + kCoordinateEpsilon));
}
template <typename IndexType>
inline IndexType getGridIndexFromPoint(const Point& scaled_point) {
  return IndexType(std::floor(scaled_point.x() + kCoordinateEpsilon), std::floor(scaled_point.y() + kCoordinateEpsilon),
                   std::floor(scaled_point.z() + kCoordinateEpsilon));
}
template <typename IndexType>
inline Point getCenterPointFromGridIndex(const IndexType& idx, FloatingPoint grid_size) {
  return Point((static_cast<FloatingPoint>(idx.x()) + 0.5) * grid_size,
               (static_cast<FloatingPoint>(idx.y()) + 0.5) * grid_size,
               (static_cast<FloatingPoint>(idx.z()) + 0.5) * grid_size);
}
template <typename IndexType>
inline Point getOriginPointFromGridIndex(const IndexType& idx, FloatingPoint grid_size) {
  return Point(static_cast<FloatingPoint>(idx.x()) * grid_size, static_cast<FloatingPoint>(idx.y()) * grid_size,
               static_cast<FloatingPoint>(idx.z()) * grid_size);
}
inline BlockIndex getBlockIndexFromGlobalVoxelIndex(const GlobalIndex& global_voxel_idx, FloatingPoint voxels_per_side_inv) {
  return BlockIndex(std::floor(static_cast<FloatingPoint>(global_voxel_idx.x()) * voxels_per_side_inv),
                    std::floor(static_cast<FloatingPoint>(global_voxel_idx.y()) * voxels_per_side_inv),
                    std::floor(static_cast<FloatingPoint>(global_voxel_idx.z()) * voxels_per_side_inv));
}
inline VoxelIndex getLocalFromGlobalVoxelIndex(const GlobalIndex& global_voxel_idx, const int voxels_per_side) {
  CHECK((voxels_per_side & (voxels_per_side - 1)) == 0) << "voxels_per_side must be a power of 2";
  const LongIndexElement m = voxels_per_side - 1;
  return VoxelIndex(static_cast<IndexElement>(global_voxel_idx.x() & m), static_cast<IndexElement>(global_voxel_idx.y() & m),
                    static_cast<IndexElement>(global_voxel_idx.z() & m));
}
inline GlobalIndex getGlobalVoxelIndexFromBlockAndVoxelIndex(const BlockIndex& block_index, const VoxelIndex& voxel_index,
                                                             int voxels_per_side) {
  return GlobalIndex(static_cast<LongIndexElement>(block_index.x()) * voxels_per_side + voxel_index.x(),
                     static_cast<LongIndexElement>(block_index.y()) * voxels_per_side + voxel_index.y(),
                     static_cast<LongIndexElement>(block_index.z()) * voxels_per_side + voxel_index.z());
}

template <typename T>
inline int signum(T val) {
  return (T(0) < val) - (val < T(0));
}

}  // namespace voxblox
