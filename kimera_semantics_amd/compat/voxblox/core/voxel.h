// Stand-in for voxblox/core/voxel.h: the TSDF voxel (12 bytes).
#pragma once
#include "voxblox/core/color.h"
#include "voxblox/core/common.h"

namespace voxblox {
struct TsdfVoxel {
  float distance = 0.0f;
  float weight = 0.0f;
  Color color;
};
}  // namespace voxblox
