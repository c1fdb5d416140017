// Stand-in for voxblox/core/block.h: a cube of voxels_per_side^3 voxels, x fastest.
#pragma once
#include <memory>
#include <vector>

#include "voxblox/core/common.h"

namespace voxblox {

template <typename VoxelType>
class Block {
 public:
  typedef std::shared_ptr<Block<VoxelType>> Ptr;
  typedef std::shared_ptr<const Block<VoxelType>> ConstPtr;

  Block(size_t voxels_per_side, FloatingPoint voxel_size, const Point& origin)
      : voxels_per_side_(voxels_per_side),
        voxel_size_(voxel_size),
        origin_(origin),
        updated_(false),
        voxels_(voxels_per_side * voxels_per_side * voxels_per_side) {
    num_voxels_ = voxels_.size();
    voxel_size_inv_ = 1.0 / voxel_size_;
    block_size_ = voxels_per_side_ * voxel_size_;
    block_size_inv_ = 1.0 / block_size_;
  }
  inline size_t computeLinearIndexFromVoxelIndex(const VoxelIndex& index) const {
    return static_cast<size_t>(index.x() + voxels_per_side_ * (index.y() + index.z() * voxels_per_side_));
  }
  inline VoxelType& getVoxelByLinearIndex(size_t index) { return voxels_[index]; }
  inline const VoxelType& getVoxelByLinearIndex(size_t index) const { return voxels_[index]; }
  inline VoxelType& getVoxelByVoxelIndex(const VoxelIndex& index) { return voxels_[computeLinearIndexFromVoxelIndex(index)]; }
  inline const VoxelType& getVoxelByVoxelIndex(const VoxelIndex& index) const {
    return voxels_[computeLinearIndexFromVoxelIndex(index)];
  }
  size_t voxels_per_side() const { return voxels_per_side_; }
  FloatingPoint voxel_size() const { return voxel_size_; }
  FloatingPoint block_size() const { return block_size_; }
  size_t num_voxels() const { return num_voxels_; }
  const Point& origin() const { return origin_; }
  bool& updated() { return updated_; }
  bool updated() const { return updated_; }
  bool has_data() const { return has_data_; }
  bool& has_data() { return has_data_; }
  VoxelType* mutable_voxels() { return voxels_.data(); }

 private:
  size_t voxels_per_side_;
  FloatingPoint voxel_size_;
  Point origin_;
  bool updated_;
  bool has_data_ = false;
  std::vector<VoxelType> voxels_;
  size_t num_voxels_;
  FloatingPoint voxel_size_inv_;
  FloatingPoint block_size_;
  FloatingPoint block_size_inv_;
};

}  // namespace voxblox
