// Stand-in for <voxblox_ros/tsdf_server.h>: the members and virtuals of voxblox::TsdfServer that
// kimera::SemanticTsdfServer (kimera_semantics_ros/src/semantic_tsdf_server.cpp) and integration/server.patch touch.
// Signatures as in ethz-asl/voxblox voxblox_ros/include/voxblox_ros/tsdf_server.h (not vendored under /root/reference).
#pragma once
#include <memory>
#include <string>
#include <ros/ros.h>
#include <voxblox/core/layer.h>
#include <voxblox/integrator/tsdf_integrator.h>
namespace voxblox {
struct MeshIntegratorConfig {};
class TsdfMap {
 public:
  struct Config {
    float tsdf_voxel_size = 0.2f;
    size_t tsdf_voxels_per_side = 16u;
  };
  explicit TsdfMap(const Config& c) : layer_(new Layer<TsdfVoxel>(c.tsdf_voxel_size, c.tsdf_voxels_per_side)) {}
  Layer<TsdfVoxel>* getTsdfLayerPtr() { return layer_.get(); }
 private:
  std::unique_ptr<Layer<TsdfVoxel>> layer_;
};
class TsdfServer {
 public:
  TsdfServer(const ros::NodeHandle&, const ros::NodeHandle&, const TsdfMap::Config& config, const TsdfIntegratorBase::Config&,
             const MeshIntegratorConfig&)
      : tsdf_map_(new TsdfMap(config)) {}
  virtual ~TsdfServer() = default;
  virtual void updateMesh() {}
  virtual bool generateMesh() { return true; }
  virtual void publishPointclouds() {}
  virtual bool saveMap(const std::string&) { return true; }
  virtual bool loadMap(const std::string&) { return true; }
  virtual void clear() {}
 protected:
  std::shared_ptr<TsdfMap> tsdf_map_;
  std::unique_ptr<TsdfIntegratorBase> tsdf_integrator_;
};
}  // namespace voxblox
