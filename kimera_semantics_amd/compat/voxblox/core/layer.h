// Stand-in for voxblox/core/layer.h: a sparse hash map of blocks.
#pragma once
#include <memory>
#include <utility>

#include "voxblox/core/block.h"
#include "voxblox/core/block_hash.h"
#include "voxblox/core/common.h"

namespace voxblox {

template <typename VoxelType>
class Layer {
 public:
  typedef std::shared_ptr<Layer> Ptr;
  typedef Block<VoxelType> BlockType;
  typedef typename AnyIndexHashMapType<typename BlockType::Ptr>::type BlockHashMap;
  typedef typename std::pair<BlockIndex, typename BlockType::Ptr> BlockMapPair;

  explicit Layer(FloatingPoint voxel_size, size_t voxels_per_side)
      : voxel_size_(voxel_size), voxels_per_side_(voxels_per_side) {
    CHECK_GT(voxel_size_, 0.0f);
    voxel_size_inv_ = 1.0 / voxel_size_;
    block_size_ = voxel_size_ * voxels_per_side_;
    CHECK_GT(block_size_, 0.0f);
    block_size_inv_ = 1.0 / block_size_;
    CHECK_GT(voxels_per_side_, 0u);
    voxels_per_side_inv_ = 1.0f / static_cast<FloatingPoint>(voxels_per_side_);
  }
  virtual ~Layer() {}

  inline typename BlockType::Ptr getBlockPtrByIndex(const BlockIndex& index) {
    typename BlockHashMap::iterator it = block_map_.find(index);
    return it != block_map_.end() ? it->second : typename BlockType::Ptr();
  }
  inline typename BlockType::ConstPtr getBlockPtrByIndex(const BlockIndex& index) const {
    typename BlockHashMap::const_iterator it = block_map_.find(index);
    return it != block_map_.end() ? it->second : typename BlockType::ConstPtr();
  }
  inline typename BlockType::Ptr allocateBlockPtrByIndex(const BlockIndex& index) {
    typename BlockHashMap::iterator it = block_map_.find(index);
    if (it != block_map_.end()) return it->second;
    return allocateNewBlock(index);
  }
  inline typename BlockType::Ptr allocateNewBlock(const BlockIndex& index) {
    auto insert_status = block_map_.emplace(
        index, std::make_shared<BlockType>(voxels_per_side_, voxel_size_, getOriginPointFromGridIndex(index, block_size_)));
    return insert_status.first->second;
  }
  inline void insertBlock(const std::pair<const BlockIndex, typename Block<VoxelType>::Ptr>& block_pair) {
    auto insert_status = block_map_.insert(block_pair);
    DCHECK(insert_status.second) << "Block already exists";
    (void)insert_status;
  }
  bool hasBlock(const BlockIndex& index) const { return block_map_.count(index) > 0; }
  void removeAllBlocks() { block_map_.clear(); }
  void getAllAllocatedBlocks(BlockIndexList* blocks) const {
    blocks->clear();
    for (const auto& kv : block_map_) blocks->push_back(kv.first);
  }
  void getAllUpdatedBlocks(BlockIndexList* blocks) const {
    blocks->clear();
    for (const auto& kv : block_map_)
      if (kv.second->updated()) blocks->push_back(kv.first);
  }
  size_t getNumberOfAllocatedBlocks() const { return block_map_.size(); }
  FloatingPoint block_size() const { return block_size_; }
  FloatingPoint block_size_inv() const { return block_size_inv_; }
  FloatingPoint voxel_size() const { return voxel_size_; }
  FloatingPoint voxel_size_inv() const { return voxel_size_inv_; }
  size_t voxels_per_side() const { return voxels_per_side_; }
  FloatingPoint voxels_per_side_inv() const { return voxels_per_side_inv_; }

 private:
  FloatingPoint voxel_size_;
  size_t voxels_per_side_;
  FloatingPoint block_size_;
  FloatingPoint voxel_size_inv_;
  FloatingPoint block_size_inv_;
  FloatingPoint voxels_per_side_inv_;
  BlockHashMap block_map_;
};

}  // namespace voxblox
