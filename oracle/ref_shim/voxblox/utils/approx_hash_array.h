// ORACLE-side restatement of voxblox/utils/approx_hash_array.h (test infrastructure).
// Voxblox is not in /root/reference; behaviour restated from its published algorithm
// (SURVEY.md A.3/A.4).
#pragma once
#include <atomic>
#include <limits>
#include <vector>

#include "voxblox/core/common.h"

namespace voxblox {

template <size_t unmasked_bits, typename StoredElement, typename IndexType, typename IndexTypeHasher>
class ApproxHashArray {
 public:
  ApproxHashArray() : pseudo_map_(size_t(1) << unmasked_bits) {}
  StoredElement& get(const size_t& hash) { return pseudo_map_[hash & bit_mask_]; }
  StoredElement& get(const IndexType& index, size_t* hash) {
    *hash = hasher_(index);
    return get(*hash);
  }
  StoredElement& get(const IndexType& index) { return get(hasher_(index)); }

 private:
  static constexpr size_t bit_mask_ = (size_t(1) << unmasked_bits) - 1;
  std::vector<StoredElement> pseudo_map_;
  IndexTypeHasher hasher_;
};

template <size_t unmasked_bits, size_t full_reset_threshold, typename IndexType, typename IndexTypeHasher>
class ApproxHashSet {
 public:
  ApproxHashSet() : offset_(0), pseudo_set_(size_t(1) << unmasked_bits) {
    for (std::atomic<size_t>& value : pseudo_set_) value.store(0, std::memory_order_relaxed);
    pseudo_set_[offset_].store(std::numeric_limits<size_t>::max());
  }
  inline bool isHashCurrentlyPresent(const size_t& hash) {
    const size_t array_index = (hash + offset_) & bit_mask_;
    return pseudo_set_[array_index].load(std::memory_order_relaxed) == hash;
  }
  inline bool isHashCurrentlyPresent(const IndexType& index) { return isHashCurrentlyPresent(hasher_(index)); }
  // true if the element was not there and has now been stored
  inline bool replaceHash(const size_t& hash) {
    const size_t array_index = (hash + offset_) & bit_mask_;
    if (pseudo_set_[array_index].load(std::memory_order_relaxed) == hash) {
      return false;
    } else {
      pseudo_set_[array_index].store(hash, std::memory_order_relaxed);
      return true;
    }
  }
  inline bool replaceHash(const IndexType& index) { return replaceHash(hasher_(index)); }
  inline void resetApproxSet() {
    if (++offset_ >= full_reset_threshold) {
      for (std::atomic<size_t>& value : pseudo_set_) value.store(0, std::memory_order_relaxed);
      offset_ = 0;
      pseudo_set_[offset_].store(std::numeric_limits<size_t>::max());
    }
  }

 private:
  static constexpr size_t bit_mask_ = (size_t(1) << unmasked_bits) - 1;
  size_t offset_;
  std::vector<std::atomic<size_t>> pseudo_set_;
  IndexTypeHasher hasher_;
};

}  // namespace voxblox
