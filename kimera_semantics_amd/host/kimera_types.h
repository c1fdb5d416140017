// kimera_types.h — the public Kimera-Semantics types the integrator boundary exchanges,
// restated for stand-alone builds (no catkin workspace).  With -DKS_USE_REAL_KIMERA the real
// headers are used instead and this file contributes nothing.
//
// Mirrors (layout and semantics, not text):
//   kimera_semantics/include/kimera_semantics/common.h:17-38        (label / probability typedefs)
//   kimera_semantics/include/kimera_semantics/color.h:19-56         (HashableColor, SemanticLabel2Color)
//   kimera_semantics/include/kimera_semantics/semantic_voxel.h:14-27 (SemanticVoxel, 92 bytes)
//   kimera_semantics/include/kimera_semantics/semantic_integrator_base.h:54-87,192-225
//        (ColorMode, SemanticConfig, the public data members other code may read)
#pragma once

#ifdef KS_USE_REAL_KIMERA
#include <kimera_semantics/color.h>
#include <kimera_semantics/common.h>
#include <kimera_semantics/semantic_integrator_base.h>
#include <kimera_semantics/semantic_voxel.h>
#else

#include <cmath>
#include <cstdint>
#include <fstream>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <voxblox/core/block_hash.h>
#include <voxblox/core/color.h>
#include <voxblox/core/common.h>
#include <voxblox/core/layer.h>

namespace kimera {
namespace vxb = voxblox;

typedef uint8_t SemanticLabel;
typedef vxb::AlignedVector<SemanticLabel> SemanticLabels;
static constexpr uint8_t kUnknownSemanticLabelId = 0u;
static constexpr size_t kTotalNumberOfLabels = 21;
typedef vxb::FloatingPoint SemanticProbability;
typedef Eigen::Matrix<SemanticProbability, kTotalNumberOfLabels, 1> SemanticProbabilities;
typedef Eigen::Matrix<SemanticProbability, kTotalNumberOfLabels, kTotalNumberOfLabels> SemanticLikelihoodFunction;

struct HashableColor : public vxb::Color {
  HashableColor() : vxb::Color() {}
  HashableColor(const vxb::Color& c) : vxb::Color(c) {}  // NOLINT: implicit like the reference
  HashableColor(uint8_t r, uint8_t g, uint8_t b) : vxb::Color(r, g, b, 255) {}
  HashableColor(uint8_t r, uint8_t g, uint8_t b, uint8_t a) : vxb::Color(r, g, b, a) {}
  bool operator==(const HashableColor& o) const { return r == o.r && g == o.g && b == o.b && a == o.a; }
};
typedef vxb::AlignedVector<HashableColor> HashableColors;
struct ColorHasher {  // hashes r,g,b only; equality also compares alpha
  size_t operator()(const HashableColor& k) const {
    return ((std::hash<uint8_t>()(k.r) ^ (std::hash<uint8_t>()(k.g) << 1)) >> 1) ^ (std::hash<uint8_t>()(k.b) << 1);
  }
};
typedef std::unordered_map<HashableColor, SemanticLabel, ColorHasher> ColorToSemanticLabelMap;
typedef std::unordered_map<SemanticLabel, HashableColor> SemanticLabelToColorMap;

// CSV "name,red,green,blue,alpha,id" -> the two maps.  Keeps the loader's observable
// behaviour: every line (header included) is a row, atoi semantics, later rows overwrite,
// id 0 <-> White is forced at the end.
class SemanticLabel2Color {
 public:
  explicit SemanticLabel2Color(const std::string& filename) {
    std::ifstream file(filename.c_str());
    CHECK(file.good()) << "Couldn't open file: " << filename;
    std::string line;
    size_t row_number = 1;
    while (std::getline(file, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.empty()) continue;
      std::vector<std::string> cells;
      size_t start = 0;
      for (;;) {
        const size_t comma = line.find(',', start);
        cells.push_back(line.substr(start, comma == std::string::npos ? std::string::npos : comma - start));
        if (comma == std::string::npos) break;
        start = comma + 1;
      }
      CHECK_EQ(cells.size(), 6u) << "Row " << row_number << " is invalid.";
      const uint8_t r = std::atoi(cells[1].c_str()), g = std::atoi(cells[2].c_str()), b = std::atoi(cells[3].c_str()),
                    a = std::atoi(cells[4].c_str()), id = std::atoi(cells[5].c_str());
      semantic_label_to_color_map_[id] = HashableColor(r, g, b, a);
      color_to_semantic_label_[HashableColor(r, g, b, a)] = id;
      ++row_number;
    }
    semantic_label_to_color_map_[kUnknownSemanticLabelId] = HashableColor(vxb::Color::White());
    color_to_semantic_label_[HashableColor(vxb::Color::White())] = kUnknownSemanticLabelId;
  }
  SemanticLabel getSemanticLabelFromColor(const HashableColor& color) const {
    const auto it = color_to_semantic_label_.find(color);
    if (it != color_to_semantic_label_.end()) return it->second;
    LOG(ERROR) << "Caught an unknown color: RGBA " << +color.r << ' ' << +color.g << ' ' << +color.b << ' ' << +color.a;
    return kUnknownSemanticLabelId;
  }
  HashableColor getColorFromSemanticLabel(const SemanticLabel& semantic_label) const {
    const auto it = semantic_label_to_color_map_.find(semantic_label);
    if (it != semantic_label_to_color_map_.end()) return it->second;
    LOG(ERROR) << "Caught an unknown semantic label: " << +semantic_label;
    return HashableColor();
  }
  ColorToSemanticLabelMap color_to_semantic_label_;
  SemanticLabelToColorMap semantic_label_to_color_map_;
};

struct SemanticVoxel {  // 92 bytes: label @0, priors @4, colour @88
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  SemanticLabel semantic_label = 0u;
  SemanticProbabilities semantic_priors = SemanticProbabilities::Constant(-0.60205999132);
  HashableColor color = HashableColor(vxb::Color::Gray());
};

enum class ColorMode : int { kColor = 0, kSemantic = 1, kSemanticProbability = 2 };

// Data-only counterpart of kimera::SemanticIntegratorBase: the configuration struct and the
// public members the reference exposes.  The CPU update methods are intentionally absent —
// on this path they run on the GPU.
class SemanticIntegratorBase {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  struct SemanticConfig {
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    SemanticProbability semantic_measurement_probability_ = 0.9f;
    ColorMode color_mode = ColorMode::kSemantic;
    std::shared_ptr<SemanticLabel2Color> semantic_label_to_color_ = nullptr;
    SemanticLabels dynamic_labels_ = SemanticLabels();
  };
  SemanticIntegratorBase(const SemanticConfig& semantic_config, vxb::Layer<SemanticVoxel>* semantic_layer)
      : semantic_config_(semantic_config), semantic_layer_(semantic_layer) {
    CHECK_NOTNULL(semantic_layer_);
    // the cached map configuration and the measurement likelihood, as the reference's constructor leaves them
    // (semantic_integrator_base.cpp:78-128; the GPU keeps its own copies: ks_create evaluates the same two logarithms)
    semantic_voxel_size_ = semantic_layer_->voxel_size();
    semantic_block_size_ = semantic_layer_->block_size();
    semantic_voxels_per_side_ = semantic_layer_->voxels_per_side();
    semantic_voxel_size_inv_ = 1.0 / semantic_voxel_size_;
    semantic_block_size_inv_ = 1.0 / semantic_block_size_;
    semantic_voxels_per_side_inv_ = 1.0 / semantic_voxels_per_side_;
    const SemanticProbability p = semantic_config_.semantic_measurement_probability_, q = 1.0f - p;
    CHECK(p > 0.0f && p < 1.0f && q > 0.0f && q < 1.0f);
    log_match_probability_ = std::log(p);
    log_non_match_probability_ = std::log(q);
    CHECK(log_match_probability_ > log_non_match_probability_);
    for (size_t j = 0; j < kTotalNumberOfLabels; ++j)
      for (size_t i = 0; i < kTotalNumberOfLabels; ++i)
        semantic_log_likelihood_(i, j) = j == 0 ? 0.0f   // the unknown label's column carries no evidence
                                                : (i == j ? log_match_probability_ : log_non_match_probability_);
  }
  const SemanticConfig semantic_config_;
  vxb::Layer<SemanticVoxel>* semantic_layer_;
  // (the reference's remaining public members, semantic_integrator_base.h:192-225; nothing on this path writes the
  // temporary block map: blocks are allocated on the GPU and appear in the layer at the next sync)
  mutable std::mutex temp_semantic_block_mutex_;
  vxb::Layer<SemanticVoxel>::BlockHashMap temp_semantic_block_map_;
  SemanticProbability log_match_probability_;
  SemanticProbability log_non_match_probability_;
  SemanticLikelihoodFunction semantic_log_likelihood_;
  vxb::FloatingPoint semantic_voxel_size_;
  size_t semantic_voxels_per_side_;
  vxb::FloatingPoint semantic_block_size_;
  vxb::FloatingPoint semantic_voxel_size_inv_;
  vxb::FloatingPoint semantic_voxels_per_side_inv_;
  vxb::FloatingPoint semantic_block_size_inv_;
};

}  // namespace kimera
#endif  // KS_USE_REAL_KIMERA
