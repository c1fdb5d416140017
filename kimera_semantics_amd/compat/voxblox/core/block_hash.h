// Stand-in for voxblox/core/block_hash.h: the index hashers and the hash-map typedef templates.
#pragma once
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <utility>

#include "voxblox/core/common.h"

namespace voxblox {

struct AnyIndexHash {
  static constexpr size_t sl = 17191;
  static constexpr size_t sl2 = sl * sl;
  std::size_t operator()(const AnyIndex& index) const {
    return static_cast<unsigned int>(index.x() + index.y() * sl + index.z() * sl2);
  }
};
template <typename ValueType>
struct AnyIndexHashMapType {
  typedef std::unordered_map<AnyIndex, ValueType, AnyIndexHash, std::equal_to<AnyIndex>,
                             Eigen::aligned_allocator<std::pair<const AnyIndex, ValueType>>>
      type;
};
typedef std::unordered_set<AnyIndex, AnyIndexHash, std::equal_to<AnyIndex>, Eigen::aligned_allocator<AnyIndex>> IndexSet;

struct LongIndexHash {
  static constexpr size_t sl = 17191;
  static constexpr size_t sl2 = sl * sl;
  std::size_t operator()(const LongIndex& index) const {
    return static_cast<unsigned int>(index.x() + index.y() * sl + index.z() * sl2);
  }
};
template <typename ValueType>
struct LongIndexHashMapType {
  typedef std::unordered_map<LongIndex, ValueType, LongIndexHash, std::equal_to<LongIndex>,
                             Eigen::aligned_allocator<std::pair<const LongIndex, ValueType>>>
      type;
};
typedef std::unordered_set<LongIndex, LongIndexHash, std::equal_to<LongIndex>, Eigen::aligned_allocator<LongIndex>> LongIndexSet;

}  // namespace voxblox
